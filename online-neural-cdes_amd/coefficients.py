"""GPU construction of control-path coefficients: mirrors of ``torchcde.linear_interpolation_coeffs`` and
``torchcde.natural_cubic_coeffs`` (/root/reference/modules/torchcde/torchcde/interpolation_linear.py:131-180,
interpolation_cubic.py:170-190) for fp32 CUDA tensors, on the default integer time grid or a user grid ``t``, backed by the
HIP kernels in csrc/ncde_prepare.hip.  (Host/numpy mirrors used to build test inputs live in data.py.)"""
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


def _grid(t, length, device):
    """misc.validate_input_path's checks on a user time grid (torchcde/misc.py:70-100) -> contiguous fp32 tensor on `device`."""
    if t is None:
        return None
    t = torch.as_tensor(t)
    if not t.is_floating_point():
        raise ValueError("t must both be floating point.")
    if t.dim() != 1:
        raise ValueError("t must be one dimensional. It instead has shape {}.".format(tuple(t.shape)))
    if t.numel() != length:
        raise ValueError("The time dimension of X must equal the length of t. X has time dimension {} and t has shape {}.".format(length, tuple(t.shape)))
    if not bool((t[1:] > t[:-1]).all()):
        raise ValueError("t must be monotonically increasing.")
    return t.to(device=device, dtype=torch.float32).contiguous()


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def linear_interpolation_coeffs(x, t=None, rectilinear=None):
    """Knots of the (rectilinear) linear interpolation; NaNs are missing values."""
    x3 = _check(x)
    B, L, C = x3.shape
    tg = _grid(t, 2 * L - 1 if rectilinear is not None else L, x.device)      # the reference validates t against the PREPARED path
    if rectilinear is not None:
        assert isinstance(rectilinear, int) and 0 <= rectilinear < C, "Index of the time channel must be an integer in [0, {}]".format(C - 1)
        assert not torch.isnan(x3[..., rectilinear]).any(), "There exist nan values in the time column which is not allowed."
    T = 2 * L - 1 if rectilinear is not None else L
    out = torch.empty(B, T, C, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.lib().ncde_prepare_linear_grid(x3.data_ptr(), None if (tg is None or rectilinear is not None) else tg.data_ptr(), B, L, C,
                                                 -1 if rectilinear is None else rectilinear, out.data_ptr(), _stream())
    _lib.check(rc, "ncde_prepare_linear_grid")
    return out.reshape(*x.shape[:-2], T, C)


def natural_cubic_coeffs(x, t=None):
    """a | b | 2c | 3d of the natural cubic spline through x; NaNs are missing values."""
    x3 = _check(x)
    B, L, C = x3.shape
    tg = _grid(t, L, x.device)
    out = torch.empty(B, L - 1, 4 * C, dtype=torch.float32, device=x.device)
    need = _lib.check(_lib.lib().ncde_prepare_workspace_bytes(_lib.INTERP["cubic"], B, L, C), "ncde_prepare_workspace_bytes")
    ws = torch.empty(int(need), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.lib().ncde_prepare_cubic_grid(x3.data_ptr(), None if tg is None else tg.data_ptr(), B, L, C, out.data_ptr(), ws.data_ptr(),
                                                ws.numel(), _stream())
    _lib.check(rc, "ncde_prepare_cubic_grid")
    return out.reshape(*x.shape[:-2], L - 1, 4 * C)


natural_cubic_spline_coeffs = natural_cubic_coeffs
