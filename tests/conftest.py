"""pytest configuration: markers, import paths, golden fixtures.

`-m "not gpu"` runs here (no GPU): oracle vs golden vectors, host logic, C-ABI surface.
`-m gpu` runs on the MI355X box: the parity tests proper, through torch.ops.torchlsq -> C ABI -> HIP.
Nothing in the gpu tests reads /root/reference.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "lsqfakequantize-pytorch_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (PKG_DIR, ROOT, os.path.join(ROOT, "tools")):     # tools/: lsq_tools.py (the tools build of the library)
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s")
    # make sure the native pieces exist (hipcc cross-compiles without a GPU; seconds when up to date)
    import __graft_entry__ as ge
    ge.build_hip(verbose=False)
    from oracle import lsq_oracle
    lsq_oracle.build()


@pytest.fixture(scope="session")
def small_cases():
    with open(os.path.join(GOLDEN, "small_cases.json")) as f:
        manifest = json.load(f)
    arrays = np.load(os.path.join(GOLDEN, "small_cases.npz"))
    return manifest, arrays


@pytest.fixture(scope="session")
def traces():
    """State-machine traces of the reference LSQFakeQuantizer (tests/golden/make_module_traces.py)."""
    with open(os.path.join(GOLDEN, "module_traces.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def config_digests():
    with open(os.path.join(GOLDEN, "config_digests.json")) as f:
        return json.load(f)["configs"]


def install_oracle_cpu_backend():
    """Historical name, kept for the test modules that call it: the product now serves CPU tensors itself
    (liblsq_cpu.so under the CPU dispatch key, include/lsq_cpu.h), so nothing is plugged in any more -- the host-logic
    and sharded tests run on the PRODUCT's CPU kernels, which tests/test_cpu_twin.py holds to the oracle."""
    import torchlsq  # noqa: F401
    from torchlsq import extension as E
    assert E._CPU_LIB is not None, "liblsq_cpu.so did not load: " + E.cpu_error_str


@pytest.fixture(scope="session")
def oracle_cpu_backend():
    install_oracle_cpu_backend()
    return True
