"""GPU: every size threshold of the per-channel launch policy (lsq_per_channel.hip: forward_per_channel, launch_bwd_pc,
backward_per_channel; lsq_pc_geom.hpp) pinned from both sides: one shape just below and one just above it, fp32 and bf16
storage, each held to the CPU oracle (y / dx bit-exact, d_scale / d_shift within 1e-6 of sum|terms|) -- and the launch note of
the tools build (tools/lsq_tools.py) says which kernel family / loop form / workgroup size actually ran, so a threshold that
moves, or a branch that stops being taken, fails here.  tools/exp_policy_cliffs.py sweeps the same thresholds for time."""
import numpy as np
import pytest
import torch

from helpers import assert_bits_equal, assert_reduction_close
from oracle import lsq_oracle as O

pytestmark = pytest.mark.gpu

MB = 1 << 20


@pytest.fixture(scope="module")
def T():
    import torchlsq  # noqa: F401
    import lsq_tools
    from torchlsq import extension
    extension._assert_has_ops()
    lsq_tools.activate()
    yield lsq_tools
    lsq_tools.deactivate()


def _case(T, shape, axis, dtype, qrange=(0, 127, 0, 255)):
    """run forward + backward on the default policy; return (forward note, backward note); parity against the oracle"""
    from torchlsq import extension as E, synth
    dev = torch.device("cuda:0")
    n = int(np.prod(shape))
    C = shape[axis]
    x = synth.normal_like(n, 61, 0.5, 1.0, dtype=dtype, device=dev).view(shape)
    g = synth.normal_like(n, 62, 0.0, 1e-3, dtype=dtype, device=dev).view(shape)
    s = synth.uniform_like(C, 63, 0.01, 0.05, device=dev)
    b = synth.normal_like(C, 64, 0.0, 0.1, device=dev)
    q = qrange + (True, 1.0, False, False, False)
    y = E.hip_forward_per_channel(x, s, b, axis, *q)
    fnote = T.last_launch()
    dx, ds, db = E.hip_backward_per_channel(g, x, s, b, axis, *q)
    bnote = T.last_launch()
    torch.cuda.synchronize()
    xs, gs_ = x.float().cpu().numpy(), g.float().cpu().numpy()
    outer, C_, inner = O.axis_to_ocl(shape, axis)
    oy = O.fwd_pc(xs, s.cpu().numpy(), b.cpu().numpy(), outer, C_, inner, *qrange)
    r = O.bwd_pc(gs_, xs, s.cpu().numpy(), b.cpu().numpy(), outer, C_, inner, *qrange, True, 1.0, False)
    tag = "%s %s" % (shape, dtype)
    if dtype == torch.float32:
        assert_bits_equal(y.cpu().numpy(), oy, tag + " y")
        assert_bits_equal(dx.cpu().numpy(), r.dx, tag + " dx")
    else:
        assert torch.equal(y.cpu().view(torch.int16), torch.from_numpy(np.ascontiguousarray(oy)).to(dtype).view(torch.int16)), tag + " y"
        assert torch.equal(dx.cpu().view(torch.int16), torch.from_numpy(np.ascontiguousarray(r.dx)).to(dtype).view(torch.int16)), tag + " dx"
    assert_reduction_close(ds.cpu().numpy(), r.ds_wide, r.abs_ds, tag + " ds")
    assert_reduction_close(db.cpu().numpy(), r.db_wide, r.abs_db, tag + " db")
    del x, g, y, dx
    torch.cuda.empty_cache()
    return fnote, bnote


def _rows(elements, C):
    """(rows just below, rows just above) an element-count threshold for [rows, C]"""
    lo = (elements - 1) // C
    return lo, lo + 1 if (lo + 1) * C >= elements else lo + 2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rows_per_workgroup_floor_gives_way_to_one_workgroup_per_cu(T, dtype):
    """the floor on the rows a workgroup walks (partial-sum traffic under ~5 %) binds only once every CU has a workgroup:
    smaller tensors spread their row tiles over the CUs instead -- a smooth rule, no size switch: as soon as there are row
    tiles for every CU the grid covers the chip (within the rounding to whole tiles per workgroup), on both sides of the 2^21
    elements where round 2 switched, and a small tensor is not cut into more pieces than it has rows"""
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    for rows in (128, 600, 1900, 2730, 2731, 3600, 8000):
        _, b = _case(T, (rows, 768), 1, dtype)
        wgs = b["grid_x"] * b["grid_y"]
        assert b["kind"] == "row-groups", (rows, b)
        assert wgs <= b["grid_x"] * rows, (rows, b)
        assert wgs >= min(b["grid_x"] * (rows // 4), cus) * 0.75, (rows, b)     # (a row tile: one to four rows)
    (_, lo), (_, hi) = _case(T, (2730, 768), 1, dtype), _case(T, (2731, 768), 1, dtype)
    assert abs(lo["grid_x"] * lo["grid_y"] - hi["grid_x"] * hi["grid_y"]) <= lo["grid_x"], (lo, hi)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_streaming_hint_above_32_mb(T, dtype):
    esz = 4 if dtype == torch.float32 else 2
    lo, hi = _rows(32 * MB // esz + 1, 2048 * 7)        # NCHW-style windows: [rows, 2048, 7] quantized on axis 1
    (_, b_lo), (_, b_hi) = _case(T, (lo, 2048, 7), 1, dtype, (-8, 7, -128, 127)), _case(T, (hi, 2048, 7), 1, dtype, (-8, 7, -128, 127))
    # ([rows,2048,7]: runs of 224 / 112 bytes -- owner windows only up to 5 * 2^20 elements, test_owner_windows_band)
    assert b_lo["kind"] in ("windows", "owners") and b_hi["kind"] == "windows" and b_lo["ring_depth"] == b_hi["ring_depth"] == 4
    assert (b_lo["ring_nt"], b_hi["ring_nt"]) == (0, 1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_owner_windows_band(T, dtype):
    """NCHW activations with short channel rows, up to 20 M elements in 16-bit storage / 13 M in fp32: OWNER windows -- a fat
    workgroup owns k whole channels (the LARGEST packet-aligned group that still gives every CU an owner) for ALL rows, stores
    d_scale / d_shift itself, one launch, no workspace (profiles/r04_owner_windows_ab2.txt: -2 .. -50 % there); above the
    bound, with too few owners for the chip, with runs that are under 512 bytes and not whole cache lines above 5 * 2^20 elements,
    or with channel rows longer than
    256 lanes: the 256-lane windows + finalize launch as before"""
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    q = (-8, 7, -128, 127)
    # 7 x 7 = 49 elements per channel row; 2048 channels in groups of 8 = one owner per CU on a 256-CU part
    lo, hi = (208, 216) if dtype == torch.bfloat16 else (135, 136)       # 20.9 / 21.7 M and 13.5 / 13.6 M elements
    assert lo * 2048 * 49 <= ((20 << 20) if dtype == torch.bfloat16 else (13 << 20)) < hi * 2048 * 49
    (_, a), (_, b) = _case(T, (lo, 2048, 7, 7), 1, dtype, q), _case(T, (hi, 2048, 7, 7), 1, dtype, q)
    assert a["kind"] == "owners" and b["kind"] == "windows", (a, b)
    assert a["grid_y"] == 1 and a["block"] <= 512 and a["ring_depth"] == 4, a
    assert a["grid_x"] >= cus and (cus != 256 or a["grid_x"] == 256), a            # the fattest group that leaves no CU without an owner
    # few rows: thinner workgroups (fewer row slots), still one launch
    (_, c) = _case(T, (16, 2048, 7, 7), 1, dtype, q)
    assert c["kind"] == "owners" and c["block"] < a["block"], (a, c)
    # too few owners to cover the chip (512 channels / k): windows
    k = 8 if dtype == torch.bfloat16 else 4
    (_, d) = _case(T, (64, 512, 7, 7), 1, dtype, q)
    assert d["kind"] == "windows" and (512 // k) * 4 < 3 * cus, d
    # long channel rows (56 x 56): windows
    (_, e) = _case(T, (8, 256, 56, 56), 1, dtype)
    assert e["kind"] == "windows", e
    # 58 rows = 2 x 29: no divisor near the ten (five) row slots a workgroup holds -- all the slots and a short last tile
    (_, f) = _case(T, (58, 2048, 7, 7), 1, dtype, q)
    assert f["kind"] == "owners" and f["block"] > 256, f
    # short runs are fine (eight channels of 8 positions: 256 / 128 bytes, whole cache lines) ...
    (_, g) = _case(T, (512, 2048, 8), 1, dtype, q)
    assert g["kind"] == "owners", g
    if dtype == torch.float32:
        # ... unless they are under 512 bytes and not whole 128-byte lines (eight channels of 7 positions = 224 bytes): those
        # only up to 5 * 2^20 elements
        (_, h), (_, i) = _case(T, (360, 2048, 7), 1, dtype, q), _case(T, (368, 2048, 7), 1, dtype, q)
        assert h["kind"] == "owners" and i["kind"] == "windows", (h, i)


def test_big_row_group_workgroups_band(T):
    """16-bit storage: one 768-lane workgroup per CU for last-axis tensors whose rows fit one window --
    2^23 .. 5 * 2^24 elements: rows of at least 64 lanes, narrower ones up to 5 * 2^22 elements or where they do not tile a
    256-lane workgroup (round 5; round 4 had [rows,64] 11-16 % behind with it, before the combine went over all lanes);
    below 2^23 (round 5, profiles/r05_ww_big_small_tensors.txt): rows of at most 96 lanes, from 3 M elements on, and from
    0.8 M on where the row does not tile a 256-lane workgroup (48, 80, 96 lanes);
    fp32 never (register loops or the usual ring are level or ahead at every size on the round-4 box)"""
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    is_fat = lambda note: note["block"] > 256                                               # noqa: E731
    is_big = lambda note: is_fat(note) and note["grid_x"] * note["grid_y"] == cus           # noqa: E731  (one per CU)
    dtype = torch.bfloat16
    lo, hi = _rows(1 << 23, 1024)                  # rows of 128 lanes: the band begins at 2^23
    (_, a), (_, b) = _case(T, (lo, 1024), 1, dtype), _case(T, (hi, 1024), 1, dtype)
    assert not is_fat(a) and is_big(b), (a, b)
    assert b["block"] <= 768
    lo, hi = _rows(5 << 24, 768)                   # the upper end of the band
    (_, a), (_, b) = _case(T, (lo, 768), 1, dtype), _case(T, (hi, 768), 1, dtype)
    assert is_big(a) and not is_fat(b), (a, b)
    # narrow rows (32 lanes) inside the band: fat up to 5 * 2^22 elements, the usual workgroups (on the ring) above; rows that
    # do not tile a 256-lane workgroup (48 lanes): fat through the whole band
    lo, hi = _rows(5 << 22, 256)
    (_, c0), (_, c) = _case(T, (lo, 256), 1, dtype), _case(T, (hi, 256), 1, dtype)
    assert is_big(c0), c0
    assert not is_fat(c) and c["kind"] == "row-groups" and c["ring_depth"] == 4, c
    (_, c1) = _case(T, ((3 << 24) // 384, 384), 1, dtype)
    assert is_big(c1), c1
    (_, d) = _case(T, ((1 << 24) // 768, 768), 1, torch.float32)
    assert not is_fat(d) and d["kind"] == "row-groups", d
    # below 2^23 elements
    for shape, fat in (((3152, 768), True), ((10000, 768), True), ((512, 768), False),         # 96 lanes: from 0.8 M
                       ((8192, 384), True), ((2048, 384), True), ((1024, 384), False),        # 48 lanes: from 0.8 M
                       ((8192, 512), True), ((2048, 512), False),                             # 64 lanes tile the workgroup: from 3 M
                       ((12608, 256), True), ((4096, 256), False),                            # 32 lanes: from 3 M
                       ((4096, 1024), False), ((2048, 2048), False)):                         # 128+ lanes: never down here
        for dt in (torch.bfloat16, torch.float16):
            (_, n) = _case(T, shape, 1, dt)
            assert is_fat(n) == fat and n["kind"] == "row-groups" and n["ring_depth"] == 4, (shape, dt, n)
        (_, n) = _case(T, shape, 1, torch.float32)
        assert not is_fat(n), (shape, n)
    # rows narrower than anything the round-5 sweep measured keep the usual four-wave workgroups: fewer than 16 lanes below
    # 2^23 elements ([rows, 8 .. 120] 16-bit), fewer than 8 lanes inside the band (round-5 advisor: a 768-lane workgroup over
    # 1-15-lane rows means hundreds of row groups and a ~100 KB combine buffer nobody timed)
    for shape in ((65536, 64), (40000, 96), (100000, 40), (1 << 18, 16), (200000, 56)):
        (_, n) = _case(T, shape, 1, dtype)
        assert not is_fat(n), (shape, n)


def test_row_group_ring_by_storage_type(T):
    """16-bit row groups: the LDS-DMA ring whatever the size (one tile per workgroup included: -3 .. -25 % on small tensors);
    fp32: register loops below 2^24 elements (level or ahead at every width: 1 M elements -8 .. -12 %), the ring from there
    up to 160 MB (profiles/r04_rowgroup_ring_small.txt, r04_rowgroup_mid.txt)"""
    (_, a) = _case(T, (1568, 512), 1, torch.bfloat16)             # one row tile per workgroup
    assert a["kind"] == "row-groups" and a["ring_depth"] == 4, a
    lo, hi = _rows(1 << 24, 1024)
    (_, b), (_, c) = _case(T, (lo, 1024), 1, torch.float32), _case(T, (hi, 1024), 1, torch.float32)
    assert b["kind"] == c["kind"] == "row-groups" and (b["ring_depth"], c["ring_depth"]) == (0, 4), (b, c)
    (_, d) = _case(T, (1024, 1024), 1, torch.float32)
    assert d["ring_depth"] == 0, d


def test_fp32_row_groups_leave_the_ring_above_160_mb(T):
    lo, hi = _rows(160 * MB // 4 + 1, 768)
    (_, a), (_, b) = _case(T, (lo, 768), 1, torch.float32), _case(T, (hi, 768), 1, torch.float32)
    assert a["kind"] == b["kind"] == "row-groups"
    assert a["ring_depth"] == 4 and b["ring_depth"] == 0, (a, b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_row_groups_give_way_to_windows_at_512_mb(T, dtype):
    esz = 4 if dtype == torch.float32 else 2
    lo, hi = _rows(512 * MB // esz, 2048)
    (_, a), (_, b) = _case(T, (lo, 2048), 1, dtype), _case(T, (hi, 2048), 1, dtype)
    assert (a["kind"], b["kind"]) == ("row-groups", "windows"), (a, b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_ring_needs_as_many_row_tiles_as_stages(T, dtype):
    """a workgroup that would walk fewer row tiles than the ring is deep runs the register loops"""
    # ((32, 512, 7, 7): too few channels for owner windows, test_owner_windows_band)
    (_, a), (_, b) = _case(T, (32, 512, 7, 7), 1, dtype, (-8, 7, -128, 127)), _case(T, (256, 2048, 7, 7), 1, dtype, (-8, 7, -128, 127))
    assert a["kind"] == b["kind"] == "windows"
    assert a["ring_depth"] == 0 and b["ring_depth"] == 4, (a, b)
    assert 32 / a["grid_y"] < 4 <= 256 / b["grid_y"]


def test_segment_mode_few_rows_long_channels(T):
    """few outer indices + long packet-aligned channel rows -> one channel per workgroup (conv / linear weights on axis 0);
    8 outer indices or a very short row -> windows (short rows: test_short_channel_rows_take_the_segment_walk)"""
    (f1, b1) = _case(T, (512, 512, 3, 3), 0, torch.float32, (-128, 127, -128, 127))
    (f2, b2) = _case(T, (7, 64, 4096), 1, torch.float32)
    (f3, b3) = _case(T, (8, 64, 4096), 1, torch.float32)
    (f4, b4) = _case(T, (512, 64), 0, torch.float32, (-128, 127, -128, 127))           # 64 elements per channel: a sixteenth of the span
    assert f1["kind"] == b1["kind"] == f2["kind"] == b2["kind"] == "segment"
    assert f3["kind"] == b3["kind"] == f4["kind"] == b4["kind"] == "windows"
    assert b1["grid_x"] == 512 and b1["grid_y"] == 1          # one workgroup per channel: d_scale finished in the kernel


def test_forward_of_16_bit_last_axis_takes_the_coarser_grid_when_it_has_the_rows(T):
    """16-bit last-axis forwards with a 32 KiB channel table (2048 channels per window) run 4 workgroups per CU (half the table
    builds of the usual 16 per CU) once a workgroup of that grid walks at least 8 rows -- [8192,4096]; with fewer rows the usual
    grid stays -- [4096,4096], where the rows-per-table floor makes the two coincide anyway"""
    (a, _), (b, _) = _case(T, (4096, 4096), 1, torch.bfloat16), _case(T, (8192, 4096), 1, torch.bfloat16)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    wg = lambda n: n["grid_x"] * n["grid_y"]
    rows = lambda n, r: r / n["grid_y"]
    assert rows(a, 4096) < 14 and rows(b, 8192) >= 14 and wg(b) <= 4 * cus + 2, (a, b)
    assert a["ring_depth"] == b["ring_depth"] == 0          # register loops either way


def test_short_channel_rows_take_the_segment_walk(T):
    """weights on axis 0 whose channel rows are shorter than a workgroup's span (256 lanes x 16 bytes): one workgroup per
    channel -- no partials, no finalize launch -- for rows of at least 1/8 of the span on up to 8 workgroups per CU, 1/2 of it
    up to 16 per CU, 3/4 of it on any channel count; window kernels otherwise (profiles/r03_seg_min_ab.txt, r04_seg_weights.txt)"""
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    f32, bf16 = torch.float32, torch.bfloat16
    q = (-128, 127, -128, 127)
    many = 16 * cus + 8
    for shape, dtype, want in (((768, 768), f32, "segment"), ((64, 64, 3, 3), f32, "segment"), ((768, 768), bf16, "segment"),
                               ((512, 128), f32, "segment"), ((512, 124), f32, "windows"),          # 1/8 of 1024 elements
                               ((512, 256), bf16, "segment"), ((512, 248), bf16, "windows"),        # 1/8 of 2048
                               ((8 * cus, 256), f32, "segment"), ((8 * cus + 8, 256), f32, "windows"),    # 8 workgroups per CU: 1/8 of the span
                               ((16 * cus, 512), f32, "segment"), ((16 * cus, 504), f32, "windows"),      # 16 per CU: half the span
                               ((many, 512), f32, "windows"),
                               ((many, 768), f32, "segment"), ((many, 764), f32, "windows"),        # 3/4 of the span: any count
                               ((many, 1536), bf16, "segment"), ((many, 1528), bf16, "windows"),
                               ((8, 512, 768), f32, "windows")):                                    # eight "rows" of channels: not a weight
        axis = 1 if len(shape) == 3 and shape[0] == 8 else 0
        f, b = _case(T, shape, axis, dtype, q)
        assert (f["kind"], b["kind"]) == (want, want), (shape, dtype, f, b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float64, torch.float16])
def test_last_axis_forward_without_the_channel_table(T, dtype):
    """a lane whose packet components are different channels (quantized axis last, or rows of fewer elements than a packet)
    reads its own scale / shift instead of building the window's LDS table: same bits as the table path (tools knob 2), also
    with scale / shift that are only 4-byte aligned, a ragged last window and the int8 levels"""
    from torchlsq import extension as E, synth
    lib = T.activate()
    dev = torch.device("cuda:0")
    q = (0, 127, 0, 255, True, 1.0, False, False, False)
    for shape, axis in (((1024, 4096), 1), ((197, 768), 1), ((64, 7, 7, 256), 3), ((33, 40), 1), ((4096, 8, 3), 1), ((512, 2056), 1)):
        n = int(np.prod(shape))
        C = shape[axis]
        x = synth.normal_like(n, 71, 0.5, 1.0, dtype=dtype, device=dev).view(shape)
        pdt = torch.float64 if dtype == torch.float64 else torch.float32
        sbuf = synth.uniform_like(C + 1, 72, 0.01, 0.05, device=dev, dtype=pdt)
        bbuf = synth.normal_like(C + 1, 73, 0.0, 0.1, device=dev, dtype=pdt)
        for s, b in ((sbuf[:C], bbuf[:C]), (sbuf[1:], bbuf[1:])):       # 16-byte aligned / only element-aligned parameters
            outs = []
            for knob in (2, 0, 1, 3):
                lib.lsq_hip_debug_set_fwd_direct(knob)
                y = E.hip_forward_per_channel(x, s, b, axis, *q)
                yq, lv = E.hip_forward_per_channel(x, s, b, axis, *q, levels_bias=0)
                outs.append((y, yq, lv))
            lib.lsq_hip_debug_set_fwd_direct(0)
            torch.cuda.synchronize()
            for o in outs[1:]:
                for u, v in zip(o, outs[0]):
                    assert torch.equal(u.view(torch.uint8) if u.dtype != torch.int8 else u, v.view(torch.uint8) if v.dtype != torch.int8 else v), (shape, dtype)
