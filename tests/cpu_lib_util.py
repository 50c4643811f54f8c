"""Loader for the C++ restatement behind the same C-ABI (oracle/_build/libncde_cpu.so, HOST pointers).  Test infrastructure:
the product package never loads this library.  The struct classes are the binding's own (one layout, two implementations)."""
import ctypes
import os

import numpy as np

from ncde_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU_LIB_PATH = os.path.join(ROOT, "oracle", "_build", "libncde_cpu.so")
_H = None


def cpu_lib():
    global _H
    if _H is None:
        h = ctypes.CDLL(CPU_LIB_PATH)
        P, G, vp, sz = ctypes.POINTER(_lib.NcdeProblem), ctypes.POINTER(_lib.NcdeGrads), ctypes.c_void_p, ctypes.c_size_t
        h.ncde_last_error_string.restype = ctypes.c_char_p
        h.ncde_stage_record_bytes.argtypes = [P]
        h.ncde_stage_record_bytes.restype = ctypes.c_int64
        h.ncde_forward.argtypes = [P, vp, vp, sz, vp]
        h.ncde_forward_record.argtypes = [P, vp, vp, vp, sz, vp]
        h.ncde_adjoint.argtypes = [P, vp, vp, G, vp, sz, vp]
        h.ncde_backward.argtypes = [P, vp, vp, G, vp, sz, vp]
        _H = h
    return _H


def ptr(a):
    return a.ctypes.data


class CpuCase:
    """One solve through libncde_cpu.so: numpy arrays in, numpy arrays out (same NcdeProblem as the HIP path)."""

    def __init__(self, coeffs, kind, z0, params, layers, method, sequence):
        self.keep = [np.ascontiguousarray(coeffs, np.float32), np.ascontiguousarray(z0, np.float32)]
        self.params = {k: np.ascontiguousarray(v, np.float32) for k, v in params.items()}
        c, z = self.keep
        p = _lib.NcdeProblem()
        p.abi_version = _lib.NCDE_ABI_VERSION
        p.batch, p.hidden = z.shape
        p.n_knots, p.channels = (c.shape[1], c.shape[2]) if kind == "linear" else (c.shape[1] + 1, c.shape[2] // 4)
        p.interp, p.method, p.output = _lib.INTERP[kind], _lib.METHOD[method], int(bool(sequence))
        p.n_layers = len(layers)
        for i, (w, b) in enumerate(layers):
            p.layer_out[i], p.layer_in[i] = self.params[w].shape
            p.layer_W[i], p.layer_b[i] = ptr(self.params[w]), ptr(self.params[b])
        p.Wo, p.bo = ptr(self.params["Wo"]), ptr(self.params["bo"])
        p.coeffs, p.coeffs_stride_b, p.coeffs_stride_t = ptr(c), c.strides[0] // 4, c.strides[1] // 4
        p.z0 = ptr(z)
        self.p, self.layers = p, layers
        self.n_out = p.n_knots if sequence else 2

    def forward(self, record=False):
        out = np.empty((self.p.batch, self.n_out, self.p.hidden), np.float32)
        lib = cpu_lib()
        if record:
            st = np.empty(lib.ncde_stage_record_bytes(ctypes.byref(self.p)) // 4, np.float32)
            rc = lib.ncde_forward_record(ctypes.byref(self.p), ptr(out), ptr(st), None, 0, None)
            assert rc == 0, lib.ncde_last_error_string()
            return out, st
        rc = lib.ncde_forward(ctypes.byref(self.p), ptr(out), None, 0, None)
        assert rc == 0, lib.ncde_last_error_string()
        return out

    def backward(self, src, grad_out, discrete=False):
        g = _lib.NcdeGrads()
        bufs = {k: np.zeros_like(v) for k, v in self.params.items()}
        dz0 = np.empty((self.p.batch, self.p.hidden), np.float32)
        g.grad_z0 = ptr(dz0)
        for i, (w, b) in enumerate(self.layers):
            g.grad_layer_W[i], g.grad_layer_b[i] = ptr(bufs[w]), ptr(bufs[b])
        g.grad_Wo, g.grad_bo = ptr(bufs["Wo"]), ptr(bufs["bo"])
        src = np.ascontiguousarray(src, np.float32)
        go = np.ascontiguousarray(grad_out, np.float32)
        lib = cpu_lib()
        fn = lib.ncde_backward if discrete else lib.ncde_adjoint
        rc = fn(ctypes.byref(self.p), ptr(src), ptr(go), ctypes.byref(g), None, 0, None)
        assert rc == 0, lib.ncde_last_error_string()
        return dz0, bufs
