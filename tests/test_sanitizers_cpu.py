"""CPU: the kernels for host memory (csrc/cpu/lsq_cpu_twin.cpp, include/lsq_cpu.h) under the compiler's sanitizers --
AddressSanitizer + UndefinedBehaviorSanitizer over every entry point, three storage types, eight modes, awkward sizes and 1 / 3 /
8 OpenMP threads.  (ThreadSanitizer is not used: libgomp is not instrumented, so every variable an `omp parallel` region shares
with its master thread reports as a race.  GPU sanitizers are not available on this pool; the device code has its own
disassembly checks, tests/test_device_code.py.)  The reference has nothing of the kind."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TWIN = os.path.join(ROOT, "lsqfakequantize-pytorch_amd", "csrc", "cpu", "lsq_cpu_twin.cpp")
DRIVER = os.path.join(ROOT, "tests", "sanitize_driver.cpp")
INC = os.path.join(ROOT, "include")


def _build_and_run(tmp_path, flags, env_extra):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "drv")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fopenmp", "-ffp-contract=off", "-I", INC] + flags + [TWIN, DRIVER, "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if b.returncode != 0 and ("cannot find" in b.stderr or "unrecognized" in b.stderr):
        pytest.skip("this toolchain lacks the sanitizer runtime: " + b.stderr[-200:])
    assert b.returncode == 0, b.stderr[-3000:]
    env = dict(os.environ, OMP_NUM_THREADS="4", **env_extra)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-1000:], r.stderr[-4000:])


def test_cpu_kernels_under_asan_and_ubsan(tmp_path):
    _build_and_run(tmp_path, ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"],
                   {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})


def test_launch_geometry_invariants_under_ubsan(tmp_path):
    """lsq_pc_geom.hpp compiled HOST-ONLY with UBSan and driven over 300 000 random shapes (tests/geom_driver.cpp): window,
    row-group, owner and segment geometry never divide by zero or overflow, and keep the invariants the kernels rely on (whole
    waves, fewer than one wave of stand-in lanes, the ring's rows, the LDS budget per CU, splits <= row tiles, ...)"""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.isfile(hipcc):
        pytest.skip("no hipcc")
    exe = str(tmp_path / "geom")
    csrc = os.path.join(ROOT, "lsqfakequantize-pytorch_amd", "csrc")
    b = subprocess.run([hipcc, "-std=c++17", "-O1", "-g", "--cuda-host-only", "-fsanitize=undefined", "-fno-sanitize-recover=undefined",
                        "-I", csrc, os.path.join(ROOT, "tests", "geom_driver.cpp"), "-o", exe], capture_output=True, text=True, timeout=900)
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.startswith("ok"), (r.stdout[-1000:], r.stderr[-3000:])
