"""pytest configuration: markers, import paths, golden fixtures, and the oracle-as-CPU-backend plug.

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
for p in (PKG_DIR, ROOT):
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


_cpu_backend_lib = []


def install_oracle_cpu_backend():
    """Register the CPU ORACLE under the CPU dispatch key of torchlsq::* (tests only).

    The product registers nothing for CPU tensors.  With the oracle plugged in, the product's whole
    Python layer (functional.lsq -> front op -> autograd -> dispatcher, LSQFakeQuantizer, the sharded
    wrapper) runs on a machine without a GPU, the same way the reference's own CPU kernels sit under
    its CPU key (lsq_cpu.cpp:298-311)."""
    if _cpu_backend_lib:
        return
    import torch
    import torchlsq  # noqa: F401
    from torchlsq import extension as E
    from oracle import lsq_oracle as O

    def _np(t):
        return t.detach().contiguous().numpy()

    def fwd_pt(x, scale, shift, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode):
        E.check_forward_dtypes(x, scale, shift)
        y = O.fwd_pt(_np(x), scale[0].item(), shift[0].item(), qmin, qmax, tmin, tmax, init_mode)
        return torch.from_numpy(y).view(x.shape)

    def bwd_pt_impl(grad, x, scale, shift, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode, n4s=0):
        E.check_backward_dtypes(grad, x, scale, shift)
        return O.bwd_pt(_np(grad), _np(x), scale[0].item(), shift[0].item(), qmin, qmax, tmin, tmax, use_gs, gs, sym,
                        eval_mode, init_mode, numel_for_scaler=(n4s if n4s > 0 else None))

    def bwd_pt(grad, x, scale, shift, *a):
        if x.numel() == 0:
            return x.clone(), scale.clone(), shift.clone()
        r = bwd_pt_impl(grad, x, scale, shift, *a)
        return torch.from_numpy(r.dx).view(x.shape), torch.from_numpy(r.ds), torch.from_numpy(r.db)

    def bwd_pt_wide(grad, x, scale, shift, *a):
        r = bwd_pt_impl(grad, x, scale, shift, *a)
        return torch.from_numpy(r.dx).view(x.shape), torch.from_numpy(np.concatenate([r.ds_wide, r.db_wide]))

    def fwd_pc(x, scale, shift, axis, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode):
        E.check_forward_dtypes(x, scale, shift)
        E.check_channel_args(x, scale, shift, axis, False)
        outer, C, inner = O.axis_to_ocl(tuple(x.shape), axis)
        y = O.fwd_pc(_np(x), _np(scale), _np(shift), outer, C, inner, qmin, qmax, tmin, tmax, init_mode)
        return torch.from_numpy(y).view(x.shape)

    def bwd_pc_impl(grad, x, scale, shift, axis, qmin, qmax, tmin, tmax, use_gs, gs, sym, eval_mode, init_mode, n4s=0):
        E.check_backward_dtypes(grad, x, scale, shift)
        E.check_channel_args(x, scale, shift, axis, True)
        outer, C, inner = O.axis_to_ocl(tuple(x.shape), axis)
        return O.bwd_pc(_np(grad), _np(x), _np(scale), _np(shift), outer, C, inner, qmin, qmax, tmin, tmax, use_gs, gs,
                        sym, eval_mode, init_mode, numel_for_scaler=(n4s if n4s > 0 else None))

    def bwd_pc(grad, x, scale, shift, axis, *a):
        if x.numel() == 0:
            return x.clone(), scale.clone(), shift.clone()
        r = bwd_pc_impl(grad, x, scale, shift, axis, *a)
        return torch.from_numpy(r.dx).view(x.shape), torch.from_numpy(r.ds), torch.from_numpy(r.db)

    def bwd_pc_wide(grad, x, scale, shift, axis, *a):
        r = bwd_pc_impl(grad, x, scale, shift, axis, *a)
        return torch.from_numpy(r.dx).view(x.shape), torch.from_numpy(np.stack([r.ds_wide, r.db_wide]))

    def minmax_pt(x):
        return torch.aminmax(x.detach().to(E._param_dtype(x)))

    def minmax_pc(x, axis):
        dims = [d for d in range(x.dim()) if d != axis]
        y = x.detach().to(E._param_dtype(x))
        return torch.amin(y, dims), torch.amax(y, dims)

    def meanstd_pt(x):
        y = x.detach().to(E._param_dtype(x))
        return y.mean(), y.std()

    def meanstd_pc(x, axis):
        dims = [d for d in range(x.dim()) if d != axis]
        y = x.detach().to(E._param_dtype(x))
        return torch.mean(y, dims), torch.std(y, dims)

    lib = torch.library.Library("torchlsq", "IMPL", "CPU")
    lib.impl("lsq_minmax_per_tensor", minmax_pt)
    lib.impl("lsq_minmax_per_channel", minmax_pc)
    lib.impl("lsq_meanstd_per_tensor", meanstd_pt)
    lib.impl("lsq_meanstd_per_channel", meanstd_pc)
    lib.impl("lsq_forward_per_tensor", fwd_pt)
    lib.impl("lsq_backward_per_tensor", bwd_pt)
    lib.impl("lsq_backward_per_tensor_wide", bwd_pt_wide)
    lib.impl("lsq_forward_per_channel", fwd_pc)
    lib.impl("lsq_backward_per_channel", bwd_pc)
    lib.impl("lsq_backward_per_channel_wide", bwd_pc_wide)
    _cpu_backend_lib.append(lib)


@pytest.fixture(scope="session")
def oracle_cpu_backend():
    install_oracle_cpu_backend()
    return True
