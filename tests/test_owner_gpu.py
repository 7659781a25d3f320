"""GPU: OWNER windows of the per-channel backward (lsq_pc_geom.hpp plan_own; DESIGN_HISTORY.md section 4) held to the CPU oracle on the
production library's default policy: the shapes here are the ones the policy sends there (NCHW-style activations of at most
13 M (fp32) / 20 M (16-bit) elements with at least ~200 owners), across storage types, channel-row lengths (1 or 2 channels per lane, 1..8 channels
per owner), row counts that do and do not divide into the row slots (stand-in lanes / the generic loop), training modes, and the
un-rounded `wide` output the sharded path reads.  dx bit-exact, d_scale / d_shift within 1e-6 of sum|terms|."""
import numpy as np
import pytest
import torch

from helpers import assert_bits_equal, assert_reduction_close
from oracle import lsq_oracle as O

pytestmark = pytest.mark.gpu

SHAPES = [(24, 2048, 7, 7), (33, 2048, 7, 7), (16, 1024, 5, 5), (40, 4096, 3, 3), (12, 1536, 14, 14), (30, 2048, 50), (64, 1024, 4, 4),
          (17, 2048, 8), (128, 2048, 7)]


@pytest.fixture(scope="module")
def T():
    import torchlsq  # noqa: F401
    import lsq_tools
    lsq_tools.activate()          # the tools build runs the production policy (all knobs 0) and tells which family ran
    yield lsq_tools
    lsq_tools.deactivate()


def _inputs(shape, dtype, dev, seed=5):
    from torchlsq import synth
    n = int(np.prod(shape))
    C = shape[1]
    x = synth.normal_like(n, seed, 0.3, 1.0, dtype=dtype, device=dev).view(shape)
    g = synth.normal_like(n, seed + 1, 0.0, 1e-3, dtype=dtype, device=dev).view(shape)
    pdt = torch.float64 if dtype == torch.float64 else torch.float32
    s = synth.uniform_like(C, seed + 2, 0.02, 0.2, device=dev, dtype=pdt)
    b = synth.normal_like(C, seed + 3, 0.0, 0.1, device=dev, dtype=pdt)
    return x, g, s, b


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16, torch.float64])
@pytest.mark.parametrize("shape", SHAPES)
def test_owner_windows_against_the_oracle(T, shape, dtype):
    from torchlsq import extension as E
    dev = torch.device("cuda:0")
    x, g, s, b = _inputs(shape, dtype, dev)
    q = (-8, 7, -128, 127)
    outer, C, inner = O.axis_to_ocl(shape, 1)
    xs = x.double().cpu().numpy() if dtype == torch.float64 else x.float().cpu().numpy()
    gs = g.double().cpu().numpy() if dtype == torch.float64 else g.float().cpu().numpy()
    took = 0
    for sym, init in ((False, False), (True, False), (False, True)):
        dx, ds, db = E.hip_backward_per_channel(g, x, s, b, 1, *q, True, 1.0, sym, False, init)
        note = T.last_launch()
        took += note["kind"] == "owners"
        torch.cuda.synchronize()
        r = O.bwd_pc(gs, xs, s.cpu().numpy(), b.cpu().numpy(), outer, C, inner, *q, True, 1.0, sym, False, init)
        tag = "%s %s sym=%d init=%d (%s)" % (shape, dtype, sym, init, note["kind"])
        if dtype in (torch.float32, torch.float64):
            assert_bits_equal(dx.cpu().numpy(), r.dx, tag + " dx")
        else:
            assert torch.equal(dx.cpu().view(torch.int16), torch.from_numpy(np.ascontiguousarray(r.dx)).to(dtype).view(torch.int16)), tag + " dx"
        assert_reduction_close(ds.cpu().numpy(), r.ds_wide, r.abs_ds, tag + " ds")
        assert_reduction_close(db.cpu().numpy(), r.db_wide, r.abs_db, tag + " db")
    # the three training modes of one shape take the same family; the 7 x 7 shapes are owner-window shapes for every storage type
    assert took in (0, 3), (shape, dtype, took)
    if shape[1:] == (2048, 7, 7):
        assert took == 3, (shape, dtype)


@pytest.mark.parametrize("shape", [(32, 2048, 7, 7), (37, 2048, 7, 7)])       # 37 rows: a short last tile (the loop's ragged form)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_owner_windows_wide_output_and_global_count(T, dtype, shape):
    """the sharded path's `wide` output (un-rounded fp64 sums) and a foreign element count in the scaler, straight from the
    owner kernel's epilogue; and the packed route (unscaled terms, caller's buffer)"""
    from torchlsq import extension as E
    dev = torch.device("cuda:0")
    x, g, s, b = _inputs(shape, dtype, dev, seed=9)
    q = (-8, 7, -128, 127)
    n = x.numel()
    dx, wide = E.hip_backward_per_channel(g, x, s, b, 1, *q, True, 1.0, False, False, False, numel_for_scaler=8 * n, want_wide=True)
    assert T.last_launch()["kind"] == "owners"
    dx2, ds, db = E.hip_backward_per_channel(g, x, s, b, 1, *q, True, 1.0, False, False, False, numel_for_scaler=8 * n)
    torch.cuda.synchronize()
    assert torch.equal(dx, dx2) and wide.shape == (2, 2048)
    assert torch.equal(wide[0].to(torch.float32), ds) and torch.equal(wide[1].to(torch.float32), db)
    packed = torch.full((2 * 2048 + 1,), float(n), dtype=torch.float64, device=dev)
    E.hip_backward_per_channel(g, x, s, b, 1, *q, False, 1.0, False, False, False, want_wide=True, wide_out=packed)
    assert T.last_launch()["kind"] == "owners"
    ds3, db3 = E.hip_sharded_finish(packed, 2048, True, dtype, q[1], True, 1.0)
    dxl, dsl, dbl = E.hip_backward_per_channel(g, x, s, b, 1, *q, True, 1.0, False, False, False)
    torch.cuda.synchronize()
    assert float(packed[-1]) == float(n)
    # both routes against the oracle at the parity bar (mixed-sign sums: 1e-6 of sum|terms|, not of the small |sum|)
    outer, C, inner = O.axis_to_ocl(shape, 1)
    r = O.bwd_pc(g.float().cpu().numpy(), x.float().cpu().numpy(), s.cpu().numpy(), b.cpu().numpy(), outer, C, inner, *q, True, 1.0, False)
    for got_s, got_b, tag in ((ds3, db3, "packed route"), (dsl, dbl, "scaler per term")):
        assert_reduction_close(got_s.cpu().numpy(), r.ds_wide, r.abs_ds, tag + " ds")
        assert_reduction_close(got_b.cpu().numpy(), r.db_wide, r.abs_db, tag + " db")


def test_owner_windows_run_to_run_and_graph_replay(T):
    """two launches agree bit for bit in fp32 outputs (the LDS atomics' arrival order only moves fp64 roundings), and a HIP-graph
    replay of the one-launch op (no workspace, no finalize node) gives the same"""
    from torchlsq import extension as E
    dev = torch.device("cuda:0")
    x, g, s, b = _inputs((48, 2048, 7, 7), torch.float32, dev, seed=21)
    q = (0, 127, 0, 255, True, 1.0, False, False, False)
    a = E.hip_backward_per_channel(g, x, s, b, 1, *q)
    assert T.last_launch()["kind"] == "owners"
    c = E.hip_backward_per_channel(g, x, s, b, 1, *q)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        E.hip_backward_per_channel(g, x, s, b, 1, *q)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            d = E.hip_backward_per_channel(g, x, s, b, 1, *q)
        gr.replay()
    torch.cuda.synchronize()
    for u, v, w in zip(a, c, d):
        assert torch.equal(u, v) and torch.equal(u, w)


@pytest.mark.parametrize("seed", range(24))
def test_owner_windows_random_shapes_forced(T, seed):
    """seeded random NCHW-style shapes with owner windows FORCED wherever the plan allows (lsq_hip_debug_set_own(1): also above
    the policy's size bound), against the oracle: row counts that leave ragged last tiles (the generic row loop), runs of 1 to 8
    channels, one or two channels per lane, few rows (thin workgroups), all storage types"""
    from torchlsq import extension as E
    rng = np.random.RandomState(1000 + seed)
    dev = torch.device("cuda:0")
    dtype = [torch.float32, torch.bfloat16, torch.float16, torch.float64][seed % 4]
    C = int(rng.choice([768, 1024, 1536, 2048, 3072, 4096]))
    inner = int(rng.choice([4, 6, 8, 9, 12, 16, 25, 36, 49, 50, 64, 81, 100, 144, 196]))
    outer = int(rng.randint(16, 160))
    while outer * C * inner > (3 << 22):
        outer = max(16, outer // 2)
        if outer == 16:
            break
    shape = (outer, C, inner)
    x, g, s, b = _inputs(shape, dtype, dev, seed=100 + seed)
    q = (-8, 7, -128, 127) if seed % 2 else (0, 127, 0, 255)
    sym, init = bool(seed % 3 == 1), bool(seed % 5 == 2)
    T.set_knob("set_own", 1)
    try:
        dx, ds, db = E.hip_backward_per_channel(g, x, s, b, 1, *q, True, 1.0, sym, False, init)
        note = T.last_launch()
    finally:
        T.set_knob("set_own", 0)
    torch.cuda.synchronize()
    xs = x.double().cpu().numpy() if dtype == torch.float64 else x.float().cpu().numpy()
    gs = g.double().cpu().numpy() if dtype == torch.float64 else g.float().cpu().numpy()
    r = O.bwd_pc(gs, xs, s.cpu().numpy(), b.cpu().numpy(), outer, C, inner, *q, True, 1.0, sym, False, init)
    tag = "%s %s sym=%d init=%d (%s %dx%d of %d lanes)" % (shape, dtype, sym, init, note["kind"], note["grid_x"], note["grid_y"], note["block"])
    if dtype in (torch.float32, torch.float64):
        assert_bits_equal(dx.cpu().numpy(), r.dx, tag + " dx")
    else:
        assert torch.equal(dx.cpu().view(torch.int16), torch.from_numpy(np.ascontiguousarray(r.dx)).to(dtype).view(torch.int16)), tag + " dx"
    assert_reduction_close(ds.cpu().numpy(), r.ds_wide, r.abs_ds, tag + " ds")
    assert_reduction_close(db.cpu().numpy(), r.db_wide, r.abs_db, tag + " db")
