"""GPU: the single-launch backward (lsq_bwd_extras.ticket, include/lsq_hip.h).

With a ticket the workgroup that finishes last folds the per-workgroup partial sums and stores d_scale / d_shift;
without one a finalize launch does.  Both routes add the partials in the same fixed order, so they must agree BIT FOR
BIT; the ticket must come back all zero after every launch (its counters wrap), also from inside a HIP graph and when
several streams run backward passes at the same time (one ticket per stream).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    import torchlsq  # noqa: F401
    from torchlsq import extension
    extension._assert_has_ops()
    return extension


def _inputs(n, dtype, dev, seed=5):
    from torchlsq import synth
    x = synth.normal_like(n, seed, 1.5, 1.0, dtype=dtype, device=dev)
    g = synth.normal_like(n, seed + 1, 0.0, 1e-3, dtype=dtype, device=dev)
    return x, g


def _bits(t):
    t = t.detach().contiguous()
    return t.view(torch.int16 if t.element_size() == 2 else (torch.int32 if t.element_size() == 4 else torch.int64)).cpu().numpy().tobytes()


@pytest.mark.parametrize("n", [1, 63, 1024, 4096 + 3, 802816, 25690112 + 5])
@pytest.mark.parametrize("mode", ["train", "sym", "eval", "init"])
def test_ticket_route_equals_finalize_route_bit_for_bit(E, n, mode):
    dev = torch.device("cuda:0")
    for dtype in (torch.float32, torch.bfloat16):
        x, g = _inputs(n, dtype, dev)
        scale, shift = torch.tensor([0.03], device=dev), torch.tensor([0.1], device=dev)
        args = (0 if mode != "sym" else -64, 127 if mode != "sym" else 63, 0 if mode != "sym" else -128,
                255 if mode != "sym" else 127, True, 1.0, mode == "sym", mode == "eval", mode == "init")
        a = E.hip_backward_per_tensor(g, x, scale, shift, *args, want_wide=True, use_ticket=True)
        b = E.hip_backward_per_tensor(g, x, scale, shift, *args, want_wide=True, use_ticket=False)
        a3 = E.hip_backward_per_tensor(g, x, scale, shift, *args, use_ticket=True)
        b3 = E.hip_backward_per_tensor(g, x, scale, shift, *args, use_ticket=False)
        torch.cuda.synchronize()
        assert _bits(a[0]) == _bits(b[0]) and _bits(a[1]) == _bits(b[1]), (n, mode, dtype)
        for u, v in zip(a3, b3):
            assert _bits(u) == _bits(v), (n, mode, dtype)
    slab = E._TICKET_SLABS[0][0]
    assert int(slab.abs().sum()) == 0, "a ticket did not return to zero"


def test_unaligned_buffers_take_the_ticket_too(E):
    dev = torch.device("cuda:0")
    xb, gb = _inputs(100003, torch.float32, dev)
    x, g = xb[1:], gb[1:]          # 4-byte aligned only: the element-wise kernel
    scale, shift = torch.tensor([0.03], device=dev), torch.tensor([0.0], device=dev)
    a = E.hip_backward_per_tensor(g, x, scale, shift, 0, 127, 0, 255, True, 1.0, False, False, False, use_ticket=True)
    b = E.hip_backward_per_tensor(g, x, scale, shift, 0, 127, 0, 255, True, 1.0, False, False, False, use_ticket=False)
    for u, v in zip(a, b):
        assert _bits(u) == _bits(v)


def test_ticket_survives_many_launches_graphs_and_streams(E):
    dev = torch.device("cuda:0")
    x, g = _inputs(802816, torch.float32, dev)
    scale, shift = torch.tensor([0.03], device=dev), torch.tensor([0.0], device=dev)
    args = (0, 127, 0, 255, True, 1.0, False, False, False)
    want = E.hip_backward_per_tensor(g, x, scale, shift, *args, use_ticket=False)
    torch.cuda.synchronize()
    # back-to-back launches on one stream reuse one ticket
    outs = [E.hip_backward_per_tensor(g, x, scale, shift, *args, use_ticket=True) for _ in range(200)]
    torch.cuda.synchronize()
    for o in outs:
        assert _bits(o[1]) == _bits(want[1]) and _bits(o[2]) == _bits(want[2])
    # several streams at once: one ticket each
    streams = [torch.cuda.Stream() for _ in range(4)]
    res = []
    for rep in range(25):
        for st in streams:
            with torch.cuda.stream(st):
                res.append(E.hip_backward_per_tensor(g, x, scale, shift, *args, use_ticket=True))
    torch.cuda.synchronize()
    for o in res:
        assert _bits(o[1]) == _bits(want[1]) and _bits(o[2]) == _bits(want[2]) and _bits(o[0]) == _bits(want[0])
    # inside a HIP graph: captured launches take NO ticket (the graph may be replayed next to eager work on the capture
    # stream; two concurrent launches must never share an arrival counter) -- same results through the finalize launch
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            captured = [E.hip_backward_per_tensor(g, x, scale, shift, *args, use_ticket=True) for _ in range(10)]
    for _ in range(5):
        gr.replay()
    torch.cuda.synchronize()
    for o in captured:
        assert _bits(o[1]) == _bits(want[1]) and _bits(o[2]) == _bits(want[2])
    assert int(E._TICKET_SLABS[0][0].abs().sum()) == 0
    keys = set(E._TICKETS)             # (device, raw stream): the default stream, the 4 side streams; none for the capture stream
    assert all((0, s_.cuda_stream) in keys for s_ in streams) and (0, torch.cuda.default_stream().cuda_stream) in keys
    assert (0, st.cuda_stream) not in keys


def test_native_binding_uses_tickets_and_matches(E):
    if E.native_lsq() is None:
        pytest.skip("C++ binding not built")
    dev = torch.device("cuda:0")
    x, g = _inputs(802816, torch.float32, dev)
    scale, shift = torch.tensor([0.03], device=dev), torch.tensor([0.0], device=dev)
    args = (0, 127, 0, 255, True, 1.0, False, False, False)
    want = E.hip_backward_per_tensor(g, x, scale, shift, *args, use_ticket=False)
    nat = torch.ops.torchlsq_native
    E.set_single_launch_backward(True)
    for _ in range(20):
        got = nat.lsq_backward_per_tensor(g, x, scale, shift, *args)
        for u, v in zip(got, want):
            assert _bits(u) == _bits(v)
    dx, wide = nat.lsq_backward_per_tensor_wide(g, x, scale, shift, *args, 4 * x.numel())
    dx2, wide2 = E.hip_backward_per_tensor(g, x, scale, shift, *args, numel_for_scaler=4 * x.numel(), want_wide=True,
                                           use_ticket=False)
    E.set_single_launch_backward("auto")
    assert _bits(wide) == _bits(wide2) and _bits(dx) == _bits(dx2)


def test_default_policy_takes_a_ticket_for_host_bound_tensors_only(E):
    """"auto" (the default): per-tensor tensors of at most 8 MB go through the one-launch route (host-bound in eager mode:
    profiles/r03_ticket_sizes.txt), bigger ones and every per-channel tensor through kernel + finalize launch; same bits"""
    dev = torch.device("cuda:0")
    E.set_single_launch_backward("auto")
    assert E._wants_ticket(8 << 20) and not E._wants_ticket((8 << 20) + 4)
    args = (0, 127, 0, 255, True, 1.0, False, False, False)
    scale, shift = torch.tensor([0.03], device=dev), torch.tensor([0.0], device=dev)
    for n in (802816, (8 << 20) // 4, (8 << 20) // 4 + 1024):
        x, g = _inputs(n, torch.float32, dev)
        want = E.hip_backward_per_tensor(g, x, scale, shift, *args, use_ticket=False)
        got = E.hip_backward_per_tensor(g, x, scale, shift, *args)                      # policy
        outs = [got]
        if E.native_lsq() is not None:
            outs.append(torch.ops.torchlsq_native.lsq_backward_per_tensor(g, x, scale, shift, *args))
        for o in outs:
            for u, v in zip(o, want):
                assert _bits(u) == _bits(v), n
    E.set_single_launch_backward(False)
    assert not E._wants_ticket(1024)
    E.set_single_launch_backward(True)
    assert E._wants_ticket(1 << 30)
    E.set_single_launch_backward("auto")
    assert int(E._TICKET_SLABS[0][0].abs().sum()) == 0
