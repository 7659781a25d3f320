"""GPU: the LDS-DMA ring loops of the window-mode per-channel kernels against their register loops.

Both are reachable on any shape through the launch-variant code (bits 12-13: 1 = registers, 2 = ring), so the ring is
exercised here on shapes the default policy would not give it: fewer rows than ring stages, ragged last row tiles, dead
lanes in the last window, row-group windows of every width, unaligned-size fallbacks.  y and dx must be bit-identical;
d_scale / d_shift bit-identical for fp32 storage (fp64 sums, rounded once), equal to fp64 rounding for fp64 storage, and
within the parity bar for 16-bit storage (fp32 pre-sums on the ring).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REG, RING = 1 << 12, 2 << 12


@pytest.fixture(scope="module")
def E():
    """torchlsq.extension on the TOOLS build of the library (tools/_tune/liblsq_hip_tools.so: the same kernels plus the
    launch-variant `_ex` entry points and the lsq_hip_debug_* policy knobs, which liblsq_hip.so does not export)."""
    import torchlsq  # noqa: F401
    from torchlsq import extension
    extension._assert_has_ops()
    import lsq_tools
    lsq_tools.activate()
    yield extension
    lsq_tools.deactivate()


def _bits(t):
    t = t.detach().contiguous()
    return t.view({1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[t.element_size()]).cpu().numpy().tobytes()


SHAPES = [
    # (shape, axis)                      what it hits
    ((3, 16, 7, 7), 1),                  # folded rows (L < window), two channels per lane
    ((5, 64, 49), 1),                    # CPL 2, 5 rows < ring depth
    ((37, 2048, 49), 1),                 # BASELINE config 5 geometry, odd row count
    ((9, 256, 56, 56), 1),               # one channel per lane, many windows
    ((2, 3, 1000, 1001), 1),             # huge inner, last window partly dead
    ((1030, 4096), 1),                   # last axis, wide rows: 64-lane row groups, ragged last tile (1030 % 4)
    ((1001, 768), 1),                    # last axis, 96/192 lanes per row
    ((333, 7, 256), 2),                  # channels-last, 32/64 lanes per row, R = 8 / 4
    ((50, 8), 1),                        # tiny rows: many row groups
    ((4100, 1024), 1),                   # exactly 256 lanes per row (fp32)
    ((123, 40), 1),                      # L not a multiple of the packet: element-wise kernels (no ring; must still work)
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float64, torch.float16])
@pytest.mark.parametrize("mode", ["train", "sym", "eval", "init"])
def test_ring_equals_register_loops(E, dtype, mode):
    from torchlsq import synth
    dev = torch.device("cuda:0")
    pdt = torch.float64 if dtype == torch.float64 else torch.float32
    for k, (shape, axis) in enumerate(SHAPES):
        n = int(np.prod(shape))
        x = synth.normal_like(n, 500 + k, 0.4, 1.0, dtype=dtype, device=dev).view(shape)
        g = synth.normal_like(n, 600 + k, 0.0, 1e-2, dtype=dtype, device=dev).view(shape)
        C = shape[axis]
        s = synth.uniform_like(C, 700 + k, 0.02, 0.2, device=dev, dtype=pdt)
        b = synth.normal_like(C, 800 + k, 0.0, 0.1, device=dev, dtype=pdt)
        q = (-8, 7, -128, 127, True, 1.0, mode == "sym", mode == "eval", mode == "init")
        for bpc in (1, 4, 16):
            base = 4 | (3 << 8) | (bpc << 16)
            y_reg = E.hip_forward_per_channel(x, s, b, axis, *q, variant=base | REG)
            y_ring = E.hip_forward_per_channel(x, s, b, axis, *q, variant=base | RING)
            r_reg = E.hip_backward_per_channel(g, x, s, b, axis, *q, variant=base | REG)
            r_ring = E.hip_backward_per_channel(g, x, s, b, axis, *q, variant=base | RING)
            torch.cuda.synchronize()
            what = (shape, axis, str(dtype), mode, bpc)
            assert _bits(y_reg) == _bits(y_ring), ("y", what)
            assert _bits(r_reg[0]) == _bits(r_ring[0]), ("dx", what)
            if dtype == torch.float32:      # fp64 sums rounded once to fp32: the same bits whatever the grid
                assert _bits(r_reg[1]) == _bits(r_ring[1]) and _bits(r_reg[2]) == _bits(r_ring[2]), ("ds/db", what)
            elif dtype == torch.float64:    # fp64 outputs: the two loops may be launched on different grids (their
                for u, v in ((r_reg[1], r_ring[1]), (r_reg[2], r_ring[2])):     # residency differs), i.e. add the same
                    scale_ = float(u.abs().max()) + 1e-300                       # partial sums in a different order
                    assert float((u - v).abs().max()) <= 1e-12 * scale_, ("ds/db", what)
            else:
                # 16-bit storage: the ring pre-adds up to 4 rows in fp32 and scales the sums: compare on the scale of
                # the sum of |terms| (an upper bound of it: every |term| <= |g| * max(|err/s|, |q - zp|) * scaler)
                for u, v in ((r_reg[1], r_ring[1]), (r_reg[2], r_ring[2])):
                    tol = 1e-6 * float(g.float().abs().sum()) * 300.0 / max(1, C) + 1e-30
                    assert float((u.double() - v.double()).abs().max()) <= tol, ("ds/db", what)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float64])
def test_row_group_windows_equal_256_lane_windows(E, dtype):
    """last-axis shapes: the row-group decomposition (constants in registers, fixed-order combine) against the 256-lane
    windows with their LDS channel table (variant bit 11), forward and backward"""
    from torchlsq import synth
    dev = torch.device("cuda:0")
    pdt = torch.float64 if dtype == torch.float64 else torch.float32
    legacy = 1 << 11
    for k, (shape, axis) in enumerate([((1030, 4096), 1), ((1001, 768), 1), ((333, 7, 256), 2), ((50, 8), 1), ((4100, 1024), 1),
                                       ((7, 512), 1), ((3000, 128), 1), ((129, 6144), 1)]):
        n = int(np.prod(shape))
        x = synth.normal_like(n, 900 + k, 0.4, 1.0, dtype=dtype, device=dev).view(shape)
        g = synth.normal_like(n, 950 + k, 0.0, 1e-2, dtype=dtype, device=dev).view(shape).abs()     # no cancellation: plain rtol
        C = shape[axis]
        s = synth.uniform_like(C, 970 + k, 0.02, 0.2, device=dev, dtype=pdt)
        b = synth.normal_like(C, 990 + k, 0.0, 0.1, device=dev, dtype=pdt)
        q = (-8, 7, -128, 127, True, 1.0, False, False, False)
        for bpc in (2, 8):
            base = 4 | (3 << 8) | (bpc << 16) | REG
            y_new = E.hip_forward_per_channel(x, s, b, axis, *q, variant=base)
            y_old = E.hip_forward_per_channel(x, s, b, axis, *q, variant=base | legacy)
            r_new = E.hip_backward_per_channel(g, x, s, b, axis, *q, variant=base)
            r_old = E.hip_backward_per_channel(g, x, s, b, axis, *q, variant=base | legacy)
            torch.cuda.synchronize()
            assert _bits(y_new) == _bits(y_old) and _bits(r_new[0]) == _bits(r_old[0]), (shape, str(dtype), bpc)
            for u, v in ((r_new[1], r_old[1]), (r_new[2], r_old[2])):
                assert torch.allclose(u.double(), v.double(), rtol=2e-6 if dtype == torch.bfloat16 else 1e-6, atol=1e-30), (shape, str(dtype), bpc)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float64, torch.float16])
@pytest.mark.parametrize("mode", ["train", "sym", "init"])
def test_768_and_1024_lane_row_group_workgroups(E, dtype, mode):
    """last-axis shapes: one 768/1024-lane workgroup per CU (the policy's choice for 8 M .. 48 M elements, forced here on
    small and ragged shapes: fewer rows than row groups, rows per group < ring depth, every window width) against the
    3-4-wave workgroups.  dx bit-identical; d_scale / d_shift: the same terms summed in another order."""
    from torchlsq import synth
    dev = torch.device("cuda:0")
    lib = E.library()
    pdt = torch.float64 if dtype == torch.float64 else torch.float32
    sym, init = mode == "sym", mode == "init"
    try:
        for k, (shape, axis) in enumerate([((1030, 4096), 1), ((1001, 768), 1), ((333, 7, 256), 2), ((50, 8), 1), ((4100, 1024), 1),
                                           ((7, 512), 1), ((3000, 128), 1), ((3, 2048), 1), ((20000, 384), 1), ((1, 768), 1)]):
            n = int(np.prod(shape))
            x = synth.normal_like(n, 1900 + k, 0.4, 1.0, dtype=dtype, device=dev).view(shape)
            g = synth.normal_like(n, 1950 + k, 0.0, 1e-2, dtype=dtype, device=dev).view(shape)
            C = shape[axis]
            s = synth.uniform_like(C, 1970 + k, 0.02, 0.2, device=dev, dtype=pdt)
            b = synth.normal_like(C, 1990 + k, 0.0, 0.1, device=dev, dtype=pdt)
            q = (-8, 7, -128, 127, False, 1.0, sym, False, init)       # no gradient scaler: |term| <= 8 |g|
            outs = {}
            for knob in (2, 1):
                lib.lsq_hip_debug_set_ww_big(knob)
                E._WS_BYTES_PC.clear()
                outs[knob] = E.hip_backward_per_channel(g, x, s, b, axis, *q)
                torch.cuda.synchronize()
            assert _bits(outs[1][0]) == _bits(outs[2][0]), (shape, str(dtype), mode)
            # 16-bit storage pre-sums rows in fp32 and the row sets differ with the workgroup size: the bar is the parity
            # bar, 1e-6 of the sum of the |terms| (bounded here by 8 sum|g| per channel); wider storage sums in fp64
            drive = (2.0 * (x.double().abs() + 1.6)) if init else g.double().abs()     # init mode: 2 (y - x) drives the terms
            bound = 8.0 * drive.movedim(axis, -1).reshape(-1, C).sum(0)
            tol = (1e-6 if dtype in (torch.bfloat16, torch.float16) else 1e-12) * bound + 1e-30
            for u, v in ((outs[1][1], outs[2][1]), (outs[1][2], outs[2][2])):
                if dtype in (torch.float32, torch.float64):
                    assert torch.allclose(u.double(), v.double(), rtol=1e-6, atol=0) or bool(((u.double() - v.double()).abs() <= tol).all()), \
                        (shape, str(dtype), mode)
                else:
                    assert bool(((u.double() - v.double()).abs() <= tol).all()), (shape, str(dtype), mode)
    finally:
        lib.lsq_hip_debug_set_ww_big(0)
        E._WS_BYTES_PC.clear()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_row_group_workgroup_sizes_on_random_last_axis_shapes(E, dtype):
    """60 seeded random [rows, C] shapes (C a multiple of the packet, 8 .. 8192; 1 .. 6000 rows): 768/1024-lane workgroups and
    the streaming hint forced on against the 3-4-wave workgroups without it.  dx bit-identical, d_scale / d_shift within the
    parity bar."""
    from torchlsq import synth
    dev = torch.device("cuda:0")
    lib = E.library()
    rng = np.random.default_rng(20260902)
    vec = 8 if dtype == torch.bfloat16 else 4
    try:
        for k in range(60):
            C = int(rng.integers(1, 8192 // vec + 1)) * vec
            rows = int(rng.integers(1, 6001)) if C <= 2048 else int(rng.integers(1, 1200))
            n = rows * C
            x = synth.normal_like(n, 3000 + k, 0.4, 1.0, dtype=dtype, device=dev).view(rows, C)
            g = synth.normal_like(n, 3100 + k, 0.0, 1e-2, dtype=dtype, device=dev).view(rows, C)
            s = synth.uniform_like(C, 3200 + k, 0.02, 0.2, device=dev)
            b = synth.normal_like(C, 3300 + k, 0.0, 0.1, device=dev)
            q = (-8, 7, -128, 127, False, 1.0, False, False, False)
            outs = {}
            for big, nt in ((2, 2), (1, 1)):
                lib.lsq_hip_debug_set_ww_big(big)
                lib.lsq_hip_debug_set_ring_nt(nt)
                E._WS_BYTES_PC.clear()
                outs[big] = E.hip_backward_per_channel(g, x, s, b, 1, *q)
                torch.cuda.synchronize()
            assert _bits(outs[1][0]) == _bits(outs[2][0]), (rows, C, str(dtype))
            bound = 8.0 * g.double().abs().sum(0)
            tol = (1e-6 if dtype == torch.bfloat16 else 1e-12) * bound + 1e-30
            for u, v in ((outs[1][1], outs[2][1]), (outs[1][2], outs[2][2])):
                d = (u.double() - v.double()).abs()
                assert bool((d <= tol).all()) or (dtype == torch.float32 and torch.allclose(u.double(), v.double(), rtol=1e-6, atol=0)), \
                    (rows, C, str(dtype))
    finally:
        lib.lsq_hip_debug_set_ww_big(0)
        lib.lsq_hip_debug_set_ring_nt(0)
        E._WS_BYTES_PC.clear()


def test_default_policy_takes_the_ring_on_large_shapes(E):
    """the launch note of the window-mode backward reports the grid; with the ring a [256,2048,7,7] bf16 backward is
    sized for 4 resident workgroups per CU (LDS-bound) and fills one round"""
    import lsq_tools
    from torchlsq import synth
    dev = torch.device("cuda:0")
    n = 256 * 2048 * 49
    x = synth.normal_like(n, 1, 0.0, 1.0, dtype=torch.bfloat16, device=dev).view(256, 2048, 7, 7)
    g = synth.normal_like(n, 2, 0.0, 1e-3, dtype=torch.bfloat16, device=dev).view(256, 2048, 7, 7)
    s, b = synth.uniform_like(2048, 3, 0.05, 0.35, device=dev), synth.normal_like(2048, 4, 0.0, 0.1, device=dev)
    E.hip_backward_per_channel(g, x, s, b, 1, -8, 7, -128, 127, True, 1.0, False, False, False)
    torch.cuda.synchronize()
    note = lsq_tools.last_launch()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert note["kind"] == "windows" and note["ring_depth"] == 4 and note["ring_nt"] == 1, note
    assert note["grid_x"] == 49 and note["resident_per_cu"] >= 3           # 49 windows of 2048 positions; residency known
    total = note["grid_x"] * note["grid_y"]
    rounds = -(-total // (note["resident_per_cu"] * cus))
    assert total / (rounds * note["resident_per_cu"] * cus) >= 0.9, (note, "the last round of workgroups is not full")
