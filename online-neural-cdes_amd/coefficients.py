"""GPU construction of control-path coefficients: mirrors of ``torchcde.linear_interpolation_coeffs`` and
``torchcde.natural_cubic_coeffs`` (/root/reference/modules/torchcde/torchcde/interpolation_linear.py:131-180,
interpolation_cubic.py:170-190) for fp32 CUDA tensors on the default integer time grid, backed by the HIP
kernels in csrc/ncde_prepare.hip.  (Host/numpy mirrors used to build test inputs live in data.py.)"""
import ctypes

import torch

from . import _lib


def _check(x):
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32):
        raise NotImplementedError("coefficient construction runs on fp32 CUDA tensors (no CPU fallback)")
    if x.dim() < 2:
        raise ValueError("X must have at least two dimensions, corresponding to time and channels.")
    if x.size(-2) < 2:
        raise ValueError("Must have a time dimension of size at least 2.")
    return x.contiguous().reshape(-1, x.size(-2), x.size(-1))


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def linear_interpolation_coeffs(x, t=None, rectilinear=None):
    """Knots of the (rectilinear) linear interpolation; NaNs are missing values."""
    if t is not None:
        raise NotImplementedError("only the default integer time grid is supported")
    x3 = _check(x)
    B, L, C = x3.shape
    if rectilinear is not None:
        assert isinstance(rectilinear, int) and 0 <= rectilinear < C, "Index of the time channel must be an integer in [0, {}]".format(C - 1)
        assert not torch.isnan(x3[..., rectilinear]).any(), "There exist nan values in the time column which is not allowed."
    T = 2 * L - 1 if rectilinear is not None else L
    out = torch.empty(B, T, C, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.lib().ncde_prepare_linear(x3.data_ptr(), B, L, C, -1 if rectilinear is None else rectilinear, out.data_ptr(), _stream())
    _lib.check(rc, "ncde_prepare_linear")
    return out.reshape(*x.shape[:-2], T, C)


def natural_cubic_coeffs(x, t=None):
    """a | b | 2c | 3d of the natural cubic spline through x; NaNs are missing values."""
    if t is not None:
        raise NotImplementedError("only the default integer time grid is supported")
    x3 = _check(x)
    B, L, C = x3.shape
    out = torch.empty(B, L - 1, 4 * C, dtype=torch.float32, device=x.device)
    need = _lib.check(_lib.lib().ncde_prepare_workspace_bytes(_lib.INTERP["cubic"], B, L, C), "ncde_prepare_workspace_bytes")
    ws = torch.empty(int(need), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.lib().ncde_prepare_cubic(x3.data_ptr(), B, L, C, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    _lib.check(rc, "ncde_prepare_cubic")
    return out.reshape(*x.shape[:-2], L - 1, 4 * C)


natural_cubic_spline_coeffs = natural_cubic_coeffs
