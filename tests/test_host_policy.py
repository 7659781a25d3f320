"""One host layer's worth of logic: the launch decisions the two host layers (Python / ctypes, C++ torch binding) must make
identically are functions of the C ABI (include/lsq_hip.h: lsq_hip_policy_ticket, lsq_hip_policy_saves_mask), and both layers
call them.  Enumerate the policy inputs and hold both layers -- and the rules themselves -- to the same answers."""
import itertools

import pytest
import torch


def test_policy_functions_state_the_rules():
    from torchlsq import extension as E
    lib = E.library()
    MB = 1 << 20
    for nbytes in (0, 1, 8 * MB - 1, 8 * MB, 8 * MB + 1, 822 * MB):
        assert lib.lsq_hip_policy_ticket(0, 0, nbytes) == 0 and lib.lsq_hip_policy_ticket(1, 0, nbytes) == 1
        assert lib.lsq_hip_policy_ticket(1, 1, nbytes) == 1                          # "always" means always (the entry point ignores it)
        assert lib.lsq_hip_policy_ticket(2, 0, nbytes) == int(nbytes <= 8 * MB)      # auto: host-bound per-tensor sizes only
        assert lib.lsq_hip_policy_ticket(2, 1, nbytes) == 0
    for ev, init, rg, mb in itertools.product((0, 1), repeat=4):
        assert lib.lsq_hip_policy_saves_mask(ev, init, rg, mb) == int(ev and not init and rg and mb)


def test_both_host_layers_decide_alike():
    from torchlsq import extension as E
    from torchlsq import _hip_host as H
    if E.native_lsq() is None and not hasattr(torch.ops, "torchlsq_native"):
        pytest.skip("the C++ binding is not built")
    probe = torch.ops.torchlsq_native._policy_probe
    shapes = [(4, 64, 56, 56), (128, 512, 56, 56), (512, 512, 3, 3), (128, 1024, 14, 14), (256, 2048, 7, 7), (1,), (2, 3),
              (64, 197, 768), (8192, 4096), (1024, 1024), (2048, 1024), (2049, 1024), (4, 1024, 512), (3, 5, 7), (1 << 21,),
              ((1 << 21) + 1,), (1 << 22,), ((1 << 22) + 1,), (16, 3, 224, 224), (32, 256, 56, 56)]
    saved = H._SINGLE_LAUNCH_BWD[0]
    try:
        for mode in (0, 1, 2):
            E.set_single_launch_backward({0: False, 1: True, 2: "auto"}[mode])
            for shape, esz, pc in itertools.product(shapes, (2, 4, 8), (False, True)):
                n = 1
                for d in shape:
                    n *= d
                for ev, init, rg, mb in itertools.product((False, True), repeat=4):
                    native = list(probe(int(pc), n * esz, ev, init, rg, mb))
                    assert native[2] == mode
                    assert bool(native[0]) == H._wants_ticket(n * esz, pc), (mode, shape, esz, pc)
                    assert bool(native[1]) == H.saves_mask(ev, init, rg, mb)
    finally:
        E.set_single_launch_backward({0: False, 1: True, 2: "auto"}[saved])
