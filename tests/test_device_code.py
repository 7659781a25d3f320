"""CPU: design rules checked on the gfx950 machine code that actually ships in liblsq_hip.so.

The device code objects are pulled out of the library's .hip_fatbin section and disassembled with the ROCm LLVM
tools (no GPU needed):
  * no v_fma_mix* -- hipcc likes to select "fp16 -> fp32, multiply, -> fp16" as a mixed-precision FMA with a +0
    addend, which loses the sign of a zero product (IO::to_elem in csrc/lsq_math.hpp prevents it);
  * no scratch (register spills / runtime-indexed private arrays) in any kernel;
  * FMA contraction is off: v_fma_f32 / v_fmac_f32 only inside the IEEE division expansion (and the explicit fp64
    fma of the statistics kernels);
  * the streaming kernels move 16-byte packets.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "lsqfakequantize-pytorch_amd", "torchlsq", "liblsq_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    """{kernel symbol: (disassembly text, metadata dict)} for every gfx950 kernel in the library."""
    for tool in ("clang-offload-bundler", "llvm-objdump", "llvm-readelf"):
        if not os.path.isfile(os.path.join(LLVM, tool)):
            pytest.skip("ROCm LLVM tool %s not found" % tool)
    tmp = tmp_path_factory.mktemp("devcode")
    fat = str(tmp / "fat.bin")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", LIB, fat], check=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    assert starts, "no offload bundle in liblsq_hip.so"
    out = {}
    for i, s in enumerate(starts):
        part = str(tmp / ("bundle%d.bin" % i))
        with open(part, "wb") as f:
            f.write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = str(tmp / ("dev%d.co" % i))
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
        meta = {}
        for m in re.finditer(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+)", notes, re.S):
            meta[m.group(1)] = int(m.group(2))
        for m in re.finditer(r"^[0-9a-f]+ <(\w+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", asm, re.S | re.M):
            if m.group(1) in meta:
                out[m.group(1)] = (m.group(2), meta[m.group(1)])
    assert len(out) > 100, "expected the whole kernel table, found %d kernels" % len(out)
    return out


def _ops(body):
    return re.findall(r"^\s+([a-z_0-9]+)\s", body, re.M)


def test_no_mixed_precision_fma_and_no_scratch(kernels):
    for name, (body, scratch) in kernels.items():
        ops = _ops(body)
        assert not [o for o in ops if o.startswith("v_fma_mix") or o.startswith("v_mad_mix")], name
        assert scratch == 0 and not [o for o in ops if o.startswith("scratch_")], "%s uses %d bytes of scratch" % (name, scratch)


def test_fp_contraction_is_off(kernels):
    """x*inv_s + zp must round twice, like the reference's x86-64 build: the only fp32/fp64 FMAs allowed in the
    fake-quantize kernels are those of the correctly rounded division expansion (v_div_fmas / v_div_fixup around them)."""
    checked = 0
    for name, (body, _) in kernels.items():
        if not re.search(r"(fwd|bwd)_(pt|pc|seg|mask)_", name):      # the fake-quantize kernels proper
            continue
        ops = _ops(body)
        n_fma = sum(1 for o in ops if re.fullmatch(r"v_(fma|fmac|mad|mac)_f(32|64)(_e32|_e64)?", o))
        n_div = sum(1 for o in ops if o.startswith("v_div_fmas_f"))
        assert n_fma <= 5 * n_div, "%s: %d FMAs for %d divisions (5 per division expected)" % (name, n_fma, n_div)
        checked += 1
    assert checked > 100


def test_streaming_kernels_move_16_byte_packets(kernels):
    picked = [n for n in kernels if re.search(r"(fwd_pt_kernel|bwd_pt_kernel|fwd_pc_kernel|bwd_pc_kernel|fwd_seg_kernel|bwd_seg_kernel)"
                                              r"INS_(6io_f32|7io_bf16|6io_f16|6io_f64)ELi(4|8|2)", n)]
    assert len(picked) > 20
    rings = 0
    for name in picked:
        ops = _ops(kernels[name][0])
        # 16-byte packets in: ordinary loads, or the LDS-DMA ring's global -> LDS copies read back with ds_read_b128
        if "global_load_lds_dwordx4" in ops:
            rings += 1
            assert "ds_read_b128" in ops and "global_store_dwordx4" in ops, name
        else:
            assert "global_load_dwordx4" in ops and "global_store_dwordx4" in ops, name
    assert rings > 10       # the window-mode per-channel kernels ship with their LDS-DMA variants
