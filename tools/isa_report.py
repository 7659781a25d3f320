#!/usr/bin/env python3
"""Compile the kernel translation units to gfx950 assembly and report, per kernel: VGPR/SGPR
counts, scratch bytes, LDS bytes, and a few instruction counts that encode the design rules
(no fp contraction outside the IEEE division sequence, 16-byte global accesses, no scratch).

usage: python tools/isa_report.py [substring-filter]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lsqfakequantize-pytorch_amd", "csrc")
FLAGS = ["-std=c++17", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
         "--cuda-device-only", "-S"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), text=True,
                         capture_output=True).stdout.splitlines()
    return dict(zip(names, out))


def main():
    filt = sys.argv[1] if len(sys.argv) > 1 else ""
    extra = [a for a in sys.argv[2:]]
    tmp = tempfile.mkdtemp(prefix="lsq_isa_")
    rows = []
    for src in ("lsq_per_tensor.hip", "lsq_per_channel.hip", "lsq_observe.hip"):
        asm = os.path.join(tmp, src + ".s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + [os.path.join(CSRC, src), "-o", asm],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read()
        # kernel bodies
        bodies = {}
        for m in re.finditer(r"^(_ZN3lsq\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
            bodies[m.group(1)] = m.group(2)
        meta = {}
        for m in re.finditer(r"\.group_segment_fixed_size: (\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size: (\d+).*?"
                             r"\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", text, re.S):
            meta[m.group(2)] = (int(m.group(1)), int(m.group(3)), int(m.group(4)), int(m.group(5)))
        for name, body in bodies.items():
            ins = re.findall(r"^\s+([a-z_0-9]+)", body, re.M)
            cnt = lambda pat: sum(1 for i in ins if re.fullmatch(pat, i))
            lds, scratch, sgpr, vgpr = meta.get(name, (-1, -1, -1, -1))
            rows.append((name, vgpr, sgpr, scratch, lds, cnt(r"v_(fma|fmac|mad)_f(32|64).*"), cnt(r"v_div_fmas_f(32|64)"),
                         cnt(r"global_load_dwordx4"), cnt(r"global_store_dwordx4"),
                         cnt(r"global_load_(dword|ushort|short_d16.*|dwordx2|ubyte)"), cnt(r"ds_add_f64|ds_add_rtn_f64"),
                         cnt(r"v_rndne_f(32|64).*"), len(ins)))
    dm = demangle([r[0] for r in rows])
    print("%-5s %-5s %-7s %-5s %-4s %-4s %-5s %-5s %-5s %-6s %-5s %-6s  kernel" %
          ("vgpr", "sgpr", "scratch", "lds", "fma", "div", "ld16", "st16", "ldsm", "ldsadd", "rndne", "instr"))
    for r in sorted(rows, key=lambda r: dm[r[0]]):
        short = re.sub(r"\(.*", "", dm[r[0]]).replace("lsq::", "").replace("void ", "")
        if filt in short:
            print("%-5d %-5d %-7d %-5d %-4d %-4d %-5d %-5d %-5d %-6d %-5d %-6d  %s" % (r[1:] + (short,)))


if __name__ == "__main__":
    main()
