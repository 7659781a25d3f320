#!/usr/bin/env python3
"""Build the REFERENCE's own CPU path into oracle/_ref/ (test infrastructure only).

This is *checker* tooling: nothing under `oracle/` is imported, linked or executed by the
product path (`lsqfakequantize-pytorch_amd/`).  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may use what this script produces.

Two artefacts, both compiled from the reference sources WHERE THEY LIE under
/root/reference (no source is copied into this repository; outputs go to oracle/_ref/ only,
which is git-ignored but travels to the GPU box with the gpurun snapshot):

1. oracle/_ref/libtorchlsq_ref_ops.so  -- the reference's real op library: its four non-CUDA
   translation units (exactly the globs of reference setup.py:71-75)
       torchlsq/csrc/torchlsq.cpp
       torchlsq/csrc/ops/lsq.cpp
       torchlsq/csrc/ops/autograd/lsq_autograd.cpp
       torchlsq/csrc/ops/cpu/lsq_cpu.cpp
   with the reference's own flags (setup.py:80,92-94,107): -std=c++17 -O3 -fopenmp
   -DAT_PARALLEL_OPENMP=1 -DTORCH18, linked against the torch that is installed in this image.
   Loading it with torch.ops.load_library registers the reference's `torchlsq::*` ops
   (schema lsq.cpp:137-146, CPU kernels lsq_cpu.cpp:298-311, autograd lsq_autograd.cpp:290-303).

   API-drift note (documented in DESIGN.md): against torch >= 2.x lsq_cpu.cpp does not compile
   unmodified, because `TensorIteratorConfig::add_input(TensorBase&&)` is now `= delete` and
   lsq_cpu.cpp:175-176,238-239 pass the temporary `torch::_unsafe_view(...)` to it.  The recipe
   therefore streams that ONE file through `sed`, renaming those four calls to ATen's
   `add_owned_input` (same semantics, takes ownership of the temporary), straight into g++'s
   stdin.  No arithmetic is touched, nothing is written anywhere but oracle/_ref/.

2. oracle/_ref/liblsq_ref_scalar.so -- `oracle/ref_scalar_driver.cpp`, a torch-free loop driver
   that #includes the reference's scalar-math header torchlsq/csrc/ops/kernels/lsq_kernel.h
   (+ global_scope.h) unmodified, straight from /root/reference.  It exposes the reference's
   per-element functions over plain arrays so the C restatement in oracle/lsq_oracle.c can be
   checked bit-for-bit without torch in the loop.

If /root/reference is absent (the GPU box), this script does nothing and the prebuilt files that
travelled with the snapshot are used as they are.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF = "/root/reference"
CSRC = os.path.join(REF, "torchlsq", "csrc")

OPS_SO = os.path.join(OUT, "libtorchlsq_ref_ops.so")
SCALAR_SO = os.path.join(OUT, "liblsq_ref_scalar.so")

# reference setup.py:80 (-std=c++17 -O3), :92-94 (OpenMP), :107 (TORCH18 for torch >= 1.8)
REF_FLAGS = ["-std=c++17", "-O3", "-fopenmp", "-DAT_PARALLEL_OPENMP=1", "-DTORCH18"]


def reference_present() -> bool:
    return os.path.isfile(os.path.join(CSRC, "ops", "cpu", "lsq_cpu.cpp"))


def _newer(target, sources):
    if not os.path.isfile(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def build_scalar(verbose=True):
    drv = os.path.join(HERE, "ref_scalar_driver.cpp")
    hdr = os.path.join(CSRC, "ops", "kernels", "lsq_kernel.h")
    if _newer(SCALAR_SO, [drv, hdr, __file__]):
        return SCALAR_SO
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off",
           "-DQUANTOPS_CPU", "-I", os.path.join(CSRC, "ops", "kernels"),
           drv, "-o", SCALAR_SO]
    if verbose:
        print("[oracle/_ref]", " ".join(cmd))
    subprocess.check_call(cmd)
    return SCALAR_SO


def build_ops(verbose=True):
    srcs = [os.path.join(CSRC, "torchlsq.cpp"),
            os.path.join(CSRC, "ops", "lsq.cpp"),
            os.path.join(CSRC, "ops", "autograd", "lsq_autograd.cpp"),
            os.path.join(CSRC, "ops", "cpu", "lsq_cpu.cpp")]
    if _newer(OPS_SO, srcs + [__file__]):
        return OPS_SO
    from torch.utils import cpp_extension as ce
    import torch
    inc = []
    for p in ce.include_paths() + [sysconfig.get_paths()["include"]]:
        inc += ["-isystem", p]
    common = REF_FLAGS + ["-fPIC", "-w", "-DTORCH_API_INCLUDE_EXTENSION_H",
                          "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
                          "-DTORCH_EXTENSION_NAME=_C"] + inc
    objs = []
    for s in srcs:
        o = os.path.join(OUT, os.path.basename(s).replace(".cpp", ".ref.o"))
        objs.append(o)
        if s.endswith("lsq_cpu.cpp"):
            # stream through sed -> g++ stdin; -I <dir of the file> keeps its relative
            # #include "../global_scope.h" resolving to the reference's own headers.
            sed = subprocess.Popen(
                ["sed", "-e", r"s/\.add_input(torch::_unsafe_view(/.add_owned_input(torch::_unsafe_view(/", s],
                stdout=subprocess.PIPE)
            cmd = ["g++"] + common + ["-I", os.path.dirname(s), "-x", "c++", "-c", "-", "-o", o]
            if verbose:
                print("[oracle/_ref] sed add_input->add_owned_input", s, "|", " ".join(cmd[:3]), "... -o", o)
            subprocess.check_call(cmd, stdin=sed.stdout)
            sed.stdout.close()
            if sed.wait() != 0:
                raise RuntimeError("sed failed")
        else:
            cmd = ["g++"] + common + ["-c", s, "-o", o]
            if verbose:
                print("[oracle/_ref]", " ".join(cmd[:3]), "...", s)
            subprocess.check_call(cmd)
    libdirs = ce.library_paths()
    link = ["g++", "-shared", "-fopenmp"] + objs + ["-o", OPS_SO]
    for d in libdirs:
        link += ["-L" + d, "-Wl,-rpath," + d]
    link += ["-lc10", "-ltorch", "-ltorch_cpu", "-ltorch_python"]
    if verbose:
        print("[oracle/_ref] link ->", OPS_SO)
    subprocess.check_call(link)
    for o in objs:
        os.remove(o)
    return OPS_SO


def build_all(verbose=True):
    """Build both artefacts when the reference is present; otherwise keep prebuilt files."""
    os.makedirs(OUT, exist_ok=True)
    if not reference_present():
        if verbose:
            print("[oracle/_ref] /root/reference absent: using prebuilt files:",
                  [f for f in (OPS_SO, SCALAR_SO) if os.path.isfile(f)])
        return
    build_scalar(verbose)
    build_ops(verbose)


if __name__ == "__main__":
    build_all()
    sys.exit(0)
