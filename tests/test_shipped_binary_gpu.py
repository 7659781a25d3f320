"""GPU: the SHIPPED library (torchlsq/liblsq_hip.so), not the tools build, on the kernel families the launch policy picks by
shape -- owner windows (NCHW activations of at most 13 M fp32 / 20 M 16-bit elements), the row groups' fat workgroup and
their LDS-DMA ring -- held to

  * digests of the REFERENCE's own per-channel ops on the same seeded inputs (tests/golden/config_digests.json, written by
    tests/golden/make_golden.py from /root/reference/torchlsq/csrc/ops/cpu/lsq_cpu.cpp:145-294): y / dx bit-exact,
    d_scale / d_shift within 1e-6 of sum|terms|;
  * the CPU oracle on more owner-band shapes and the training modes (affine / symmetric / init);
  * the tools build of the same sources (tools/_tune/liblsq_hip_tools.so, what the branch-pinning suites run on): bit-identical
    outputs on seeded shapes, so what those suites prove transfers to the product.

Which family a shape runs is asked of the shipped library itself (lsq_hip_plan_backward_per_channel: the launch policy run as
a plan) -- no launch note, no debug entry point."""
import numpy as np
import pytest
import torch

from helpers import assert_bits_equal, assert_reduction_close, sha
from oracle import lsq_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    import torchlsq  # noqa: F401
    from torchlsq import extension
    extension._assert_has_ops()
    import lsq_tools
    lsq_tools.deactivate()          # (a no-op unless an earlier module left the tools build active)
    lib = extension.library()
    assert lib._name.endswith("torchlsq/liblsq_hip.so"), lib._name          # the product, as shipped
    assert not hasattr(lib, "lsq_hip_debug_last_launch")
    return extension


def _sha_t(t):
    t = t.detach().contiguous()
    if t.dtype == torch.bfloat16:
        t = t.view(torch.int16)
    return sha(t.cpu().numpy())


# digest -> (storage type, what the shipped library's plan must say about the backward)
DIGESTS = {
    "own33_fp32": (torch.float32, dict(kind="owners")),            # 33 rows: a short last row tile (the ragged loop form)
    "own33_bf16": (torch.bfloat16, dict(kind="owners")),
    "own64_fp32": (torch.float32, dict(kind="owners")),            # symmetric range
    "own64_bf16": (torch.bfloat16, dict(kind="owners")),
    "own16_fp32": (torch.float32, dict(kind="owners")),            # 14 x 14 maps: one channel row = 49 packets
    "own32_bf16": (torch.bfloat16, dict(kind="owners")),           # 28 x 28 maps, two channels per owner
    "vit_fp32": (torch.float32, dict(kind="row-groups", ring_depth=0)),             # below 2^24 elements: register loops
    "vit_bf16": (torch.bfloat16, dict(kind="row-groups", block=768)),               # the fat workgroup, one per CU
    "rgring_fp32": (torch.float32, dict(kind="row-groups", ring_depth=4, block=256)),   # the row groups' LDS-DMA ring
}


@pytest.mark.parametrize("name", sorted(DIGESTS))
def test_policy_shapes_match_the_reference_digests(E, config_digests, name):
    from torchlsq import synth
    from torchlsq.functional import lsq
    dtype, want_plan = DIGESTS[name]
    d = config_digests[name]
    dev = torch.device("cuda:0")
    x, g, scale, shift = synth.make_inputs(d["config"], device=dev, dtype=dtype, abs_grad=d["abs_grad"])
    kw = synth.op_kwargs(d["config"])
    plan = E.hip_plan_backward_per_channel(x, kw["axis"], sym=not kw["is_affine"])
    for k, v in want_plan.items():
        assert plan[k] == v, (name, plan)
    x.requires_grad_(True); scale.requires_grad_(True); shift.requires_grad_(True)
    y = lsq(x, scale, shift, **kw)
    y.backward(g)
    torch.cuda.synchronize()
    assert _sha_t(x.float()) == d["inputs_sha256"]["x"], "GPU input generation is not bit-identical to the CPU generator"
    assert _sha_t(g.float()) == d["inputs_sha256"]["g"]
    if dtype == torch.bfloat16:
        assert _sha_t(y) == d["y_bf16_sha256"], name + ": y differs from the reference (fp32 csrc, rounded to bf16)"
        assert _sha_t(x.grad) == d["dx_bf16_sha256"], name + ": dx differs from the reference"
    else:
        assert _sha_t(y) == d["y_sha256"], name + ": y differs from the reference"
        assert _sha_t(x.grad) == d["dx_sha256"], name + ": dx differs from the reference"
    ds = scale.grad.cpu().numpy()
    db = shift.grad.cpu().numpy() if shift.grad is not None else np.zeros(len(d["db"]))
    assert_reduction_close(ds, d["ds"], d["oracle_abs_ds"], name + " ds")
    assert_reduction_close(db, d["db"], d["oracle_abs_db"], name + " db")


def _inputs(shape, axis, dtype, dev, seed):
    from torchlsq import synth
    n = int(np.prod(shape))
    C = shape[axis]
    x = synth.normal_like(n, seed, 0.3, 1.0, dtype=dtype, device=dev).view(shape)
    g = synth.normal_like(n, seed + 1, 0.0, 1e-3, dtype=dtype, device=dev).view(shape)
    pdt = torch.float64 if dtype == torch.float64 else torch.float32
    s = synth.uniform_like(C, seed + 2, 0.02, 0.2, device=dev, dtype=pdt)
    b = synth.normal_like(C, seed + 3, 0.0, 0.1, device=dev, dtype=pdt)
    return x, g, s, b


def _against_oracle(E, shape, axis, dtype, q, sym, init, seed, want_kind=None):
    dev = torch.device("cuda:0")
    x, g, s, b = _inputs(shape, axis, dtype, dev, seed)
    plan = E.hip_plan_backward_per_channel(x, axis, sym=sym, init_mode=init)
    if want_kind is not None:
        assert plan["kind"] == want_kind, (shape, dtype, plan)
    y = E.hip_forward_per_channel(x, s, b, axis, *q, True, 1.0, sym, False, init)
    dx, ds, db = E.hip_backward_per_channel(g, x, s, b, axis, *q, True, 1.0, sym, False, init)
    torch.cuda.synchronize()
    outer, C, inner = O.axis_to_ocl(shape, axis)
    wide = dtype == torch.float64
    xs = x.double().cpu().numpy() if wide else x.float().cpu().numpy()
    gs = g.double().cpu().numpy() if wide else g.float().cpu().numpy()
    oy = O.fwd_pc(xs, s.cpu().numpy(), b.cpu().numpy(), outer, C, inner, *q, init)
    r = O.bwd_pc(gs, xs, s.cpu().numpy(), b.cpu().numpy(), outer, C, inner, *q, True, 1.0, sym, False, init)
    tag = "%s %s sym=%d init=%d (%s, %d x %d of %d lanes)" % (shape, dtype, sym, init, plan["kind"], plan["grid_x"], plan["grid_y"], plan["block"])
    if dtype in (torch.float32, torch.float64):
        assert_bits_equal(y.cpu().numpy(), oy, tag + " y")
        assert_bits_equal(dx.cpu().numpy(), r.dx, tag + " dx")
    else:
        assert torch.equal(y.cpu().view(torch.int16), torch.from_numpy(np.ascontiguousarray(oy)).to(dtype).view(torch.int16)), tag + " y"
        assert torch.equal(dx.cpu().view(torch.int16), torch.from_numpy(np.ascontiguousarray(r.dx)).to(dtype).view(torch.int16)), tag + " dx"
    assert_reduction_close(ds.cpu().numpy(), r.ds_wide, r.abs_ds, tag + " ds")
    assert_reduction_close(db.cpu().numpy(), r.db_wide, r.abs_db, tag + " db")
    return plan


OWNER_SHAPES = [(64, 2048, 7, 7), (33, 2048, 7, 7), (16, 1024, 14, 14), (32, 512, 28, 28), (40, 4096, 3, 3), (58, 2048, 7, 7),
                (64, 2048, 5, 5), (24, 1024, 8)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", OWNER_SHAPES)
def test_owner_band_on_the_shipped_library(E, shape, dtype):
    """the production owner kernel -- ragged tail loop, stand-in lanes, priority turns -- against the oracle in the three
    training modes, on shapes the SHIPPED policy sends there (asserted through the plan query)"""
    for k, (sym, init) in enumerate(((False, False), (True, False), (False, True))):
        _against_oracle(E, shape, 1, dtype, (-8, 7, -128, 127), sym, init, seed=31 + k, want_kind="owners")


@pytest.mark.parametrize("shape,axis,dtype,want", [
    ((12608, 768), 1, torch.bfloat16, dict(kind="row-groups", block=768)),          # fat workgroup (16-bit, 2^23 .. 5 * 2^24 elements)
    ((4096, 1024), 1, torch.bfloat16, dict(kind="row-groups", block=256, ring_depth=4)),   # 16-bit row groups: the ring at any size
    ((16400, 1024), 1, torch.float32, dict(kind="row-groups", ring_depth=4)),       # fp32 ring from 2^24 elements
    ((3000, 768), 1, torch.float32, dict(kind="row-groups", ring_depth=0)),         # fp32 register loops below
    ((320, 256, 14, 14), 1, torch.float32, dict(kind="windows")),                   # above the owner band: 256-lane windows
    ((512, 512, 3, 3), 0, torch.float32, dict(kind="segment")),                     # BASELINE config 3's family
])
def test_other_families_on_the_shipped_library(E, shape, axis, dtype, want):
    plan = _against_oracle(E, shape, axis, dtype, (0, 127, 0, 255), False, False, seed=77)
    for k, v in want.items():
        assert plan[k] == v, (shape, dtype, plan)


def test_the_plan_is_what_the_tools_build_launches(E):
    """lsq_hip_plan_backward_per_channel of the shipped library == the launch note of the tools build's real launch, on the
    shapes above: the plan query is the policy, not a description of it"""
    import lsq_tools
    dev = torch.device("cuda:0")
    cases = [(s, 1, dt) for s in OWNER_SHAPES[:4] for dt in (torch.float32, torch.bfloat16)] + \
            [((12608, 768), 1, torch.bfloat16), ((16400, 1024), 1, torch.float32), ((320, 256, 14, 14), 1, torch.float32),
             ((512, 512, 3, 3), 0, torch.float32), ((3000, 768), 1, torch.float32)]
    plans = []
    for shape, axis, dtype in cases:
        x, g, s, b = _inputs(shape, axis, dtype, dev, 5)
        plans.append(E.hip_plan_backward_per_channel(x, axis))
    lsq_tools.activate()
    try:
        for (shape, axis, dtype), plan in zip(cases, plans):
            x, g, s, b = _inputs(shape, axis, dtype, dev, 5)
            E.hip_backward_per_channel(g, x, s, b, axis, 0, 127, 0, 255, True, 1.0, False, False, False)
            note = lsq_tools.last_launch()
            for k in ("kind", "grid_x", "grid_y", "block", "ring_depth", "ring_nt"):
                assert plan[k] == note[k], (shape, dtype, k, plan, note)
    finally:
        lsq_tools.deactivate()


def test_tools_build_and_product_give_the_same_bits(E):
    """20 seeded shapes across the kernel families, forward and backward, on the product and on the tools build (same sources,
    -DLSQ_TOOLS, all knobs 0): every output bit-identical -- fp64 d_scale / d_shift to an fp64 rounding.  (Owner windows add their
    waves' sums in a fixed order; the 256-lane windows' LDS atomics can only move an fp64 rounding, 1e-9 odds of reaching an
    fp32 bit.)"""
    import lsq_tools
    rng = np.random.RandomState(20251003)
    dev = torch.device("cuda:0")
    cases = []
    for i in range(20):
        dtype = [torch.float32, torch.bfloat16, torch.float16, torch.float64][i % 4]
        kind = i % 5
        if kind == 0:      # owner band
            shape, axis = (int(rng.randint(16, 96)), int(rng.choice([1024, 2048])), 7, 7), 1
        elif kind == 1:    # token layout
            shape, axis = (int(rng.randint(500, 6000)), int(rng.choice([384, 768, 1024])),), 1
        elif kind == 2:    # NCHW above the owner band / odd inner
            shape, axis = (int(rng.randint(4, 24)), int(rng.choice([96, 256])), int(rng.choice([13, 28])), int(rng.choice([13, 28]))), 1
        elif kind == 3:    # weights
            shape, axis = (int(rng.choice([256, 512])), int(rng.choice([128, 512])), 3, 3), 0
        else:              # ragged everything
            shape, axis = (int(rng.randint(3, 40)), int(rng.randint(3, 70)), int(rng.randint(1, 50))), int(rng.randint(0, 3))
        cases.append((shape, axis, dtype, bool(i % 3 == 1), bool(i % 7 == 3)))

    def run_all():
        outs = []
        for k, (shape, axis, dtype, sym, init) in enumerate(cases):
            x, g, s, b = _inputs(shape, axis, dtype, dev, 900 + k)
            q = (-8, 7, -128, 127) if k % 2 else (0, 127, 0, 255)
            y = E.hip_forward_per_channel(x, s, b, axis, *q, True, 1.0, sym, False, init)
            dx, ds, db = E.hip_backward_per_channel(g, x, s, b, axis, *q, True, 1.0, sym, False, init)
            xp, gp = x.reshape(-1), g.reshape(-1)
            yt = E.hip_forward_per_tensor(xp, s[:1], b[:1], *q, True, 1.0, sym, False, init)
            dxt, dst, dbt = E.hip_backward_per_tensor(gp, xp, s[:1], b[:1], *q, True, 1.0, sym, False, init)
            torch.cuda.synchronize()
            outs.append([t.cpu() for t in (y, dx, ds, db, yt, dxt, dst, dbt)])
        return outs

    before = E.host_binding()
    E.set_host_binding("ctypes")        # the host layer the tools build runs under: the library is the only difference
    try:
        product = run_all()
        lsq_tools.activate()
        try:
            tools = run_all()
        finally:
            lsq_tools.deactivate()
    finally:
        E.set_host_binding(before)
    names = ("y", "dx", "ds", "db", "y (per tensor)", "dx (per tensor)", "ds (per tensor)", "db (per tensor)")
    for case, a, b in zip(cases, product, tools):
        for n, u, v in zip(names, a, b):
            assert u.dtype == v.dtype and u.shape == v.shape, (case, n)
            if u.dtype == torch.float64 and n in ("ds", "db"):
                # fp64 sums of the 256-lane windows: the four waves' LDS atomics land in arrival order, two LAUNCHES of one
                # binary agree to an fp64 rounding only (DESIGN_HISTORY.md section 4) -- so do two binaries
                assert torch.allclose(u, v, rtol=1e-13, atol=0, equal_nan=True), (case, n)
            else:
                assert torch.equal(u.view(torch.uint8), v.view(torch.uint8)), (case, n)
