"""GPU: init_mode hands values THROUGH (lsq_kernel.h:9 y = x; :112 / :140 dX = grad) -- and with 16-bit storage "through"
means the storage bits, not float(x) put through the rounding conversion again: v_cvt_pk_bf16_f32 writes the canonical NaN
(0x7fc0) for every NaN, which tools/soak_parity.py showed as y != x on NaN inputs.  Every one of the 65536 bit patterns of
the storage type goes in once, as x and as the gradient, through the per-tensor kernels (packets and the element-wise kernel
of unaligned views) and three per-channel families; fp16 leaves out its signalling NaNs (v_cvt_f32_f16 quiets them on the way
IN, as any arithmetic would)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [
    ("per-tensor", (65536,), 0, False),
    ("per-tensor, unaligned view", (65536,), 0, False),
    ("row groups", (256, 256), 1, True),
    ("windows", (16, 16, 16, 16), 1, True),
    ("segment", (16, 4096), 0, True),
]


def _patterns(dtype):
    bits = np.arange(65536, dtype=np.uint16)
    if dtype == torch.float16:
        snan = ((bits & 0x7C00) == 0x7C00) & ((bits & 0x0200) == 0) & ((bits & 0x03FF) != 0)
        bits = np.where(snan, bits | 0x0200, bits).astype(np.uint16)
    return torch.from_numpy(bits.view(np.int16).copy())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("eval_mode", [False, True])
def test_init_mode_hands_the_storage_bits_through(dtype, case, eval_mode):
    import torchlsq  # noqa: F401
    from torchlsq import extension
    from torchlsq.functional import lsq
    extension._assert_has_ops()
    name, shape, axis, per_channel = case
    dev = torch.device("cuda:0")
    pat = _patterns(dtype).to(dev)

    def place(perm_seed):
        p = pat[torch.randperm(65536, generator=torch.Generator().manual_seed(perm_seed)).to(dev)]
        if "unaligned" in name:
            flat = torch.empty(65537, dtype=torch.int16, device=dev)[1:]
            flat.copy_(p)
            p = flat
        return p.view(dtype).view(shape)

    x = place(1).requires_grad_(True)
    g = place(2)
    C = shape[axis] if per_channel else 1
    scale = torch.full((C,), 0.05, device=dev).requires_grad_(True)
    shift = torch.full((C,), 0.1, device=dev).requires_grad_(True)
    y = lsq(x, scale, shift, 0, 127, 0, 255, axis, True, 1.0, True, per_channel, eval_mode, True)
    y.backward(g)
    torch.cuda.synchronize()
    assert torch.equal(y.detach().view(torch.int16), x.detach().view(torch.int16)), name + ": y is not x bit for bit"
    assert torch.equal(x.grad.view(torch.int16), g.view(torch.int16)), name + ": dX is not the gradient bit for bit"
