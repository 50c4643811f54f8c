"""ctypes binding of libncde_hip.so (C-ABI declared in include/ncde_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or fails to load, ``lib()``
raises.  Build it with ``python __graft_entry__.py`` (or ``make -C online-neural-cdes_amd/csrc``).
"""
import ctypes
import os

NCDE_ABI_VERSION = 4
NCDE_MAX_LAYERS = 8

INTERP = {"linear": 0, "cubic": 1}
METHOD = {"euler": 0, "midpoint": 1, "rk4": 2}
OUT_INTERVAL, OUT_KNOTS, OUT_TIMES = 0, 1, 2
FIELD_KIND = {"original": 0, "minimal": 1, "gru": 2}
FIELD_INPUT = {"matmul": 0, "evaluate": 1, "derivative": 2}
FLAG_AUTO, FLAG_FORCE_GENERIC, FLAG_FORCE_FAST, FLAG_FP32_MFMA, FLAG_ADJOINT_V1, FLAG_ADJOINT_V2 = 0, 1, 2, 4, 8, 16
FLAG_ADJOINT_V4 = 32
FLAG_SPLIT_BF16 = 64        # specialised forward kernels: 3-way split-bf16 GEMMs instead of the default 2-way split-fp16 ones
FLAG_NO_COOP = 0x400          # batch-tiled forward / backward: per-workgroup kernels instead of the XCD-cooperative output phase (round 5)
FLAG_COOP_FAULT_INJECT = 0x800   # verification: one workgroup of the first cooperative launch withholds its arrival (see include/ncde_hip.h)
FLAG_ADJOINT_SPLIT_FP16 = 128   # development builds of the library only (ignored otherwise): see DESIGN.md 5.4c


def FLAG_TILED_WINDOW_STEPS(n):
    return (int(n) & 0xFF) << 16
FLAG_TILED_NS1, FLAG_TILED_NS2, FLAG_TILED_NS4, FLAG_FORCE_TILED = 0x1000, 0x2000, 0x4000, 0x8000

_c_float_p = ctypes.c_void_p  # device pointers are passed as integers


class NcdeProblem(ctypes.Structure):
    _fields_ = [
        ("abi_version", ctypes.c_int32),
        ("batch", ctypes.c_int32),
        ("n_knots", ctypes.c_int32),
        ("channels", ctypes.c_int32),
        ("hidden", ctypes.c_int32),
        ("interp", ctypes.c_int32),
        ("method", ctypes.c_int32),
        ("output", ctypes.c_int32),
        ("flags", ctypes.c_uint32),
        ("n_layers", ctypes.c_int32),
        ("layer_in", ctypes.c_int32 * NCDE_MAX_LAYERS),
        ("layer_out", ctypes.c_int32 * NCDE_MAX_LAYERS),
        ("layer_W", _c_float_p * NCDE_MAX_LAYERS),
        ("layer_b", _c_float_p * NCDE_MAX_LAYERS),
        ("Wo", _c_float_p),
        ("bo", _c_float_p),
        ("coeffs", _c_float_p),
        ("coeffs_stride_b", ctypes.c_int64),
        ("coeffs_stride_t", ctypes.c_int64),
        ("z0", _c_float_p),
        # ABI version 2: vector-field variants
        ("field_kind", ctypes.c_int32),
        ("field_input", ctypes.c_int32),
        ("Wg", _c_float_p),
        ("bg", _c_float_p),
        ("Wr", _c_float_p),
        ("br", _c_float_p),
        # ABI version 3: general time axis
        ("time_plan", ctypes.c_void_p),
        ("n_t_out", ctypes.c_int32),
        ("n_steps_fwd", ctypes.c_int32),
        ("n_steps_adj", ctypes.c_int32),
        ("reserved_", ctypes.c_int32),
    ]


class NcdeTimeSpec(ctypes.Structure):
    _fields_ = [
        ("n_t", ctypes.c_int32),
        ("time_is_f64", ctypes.c_int32),
        ("t", ctypes.POINTER(ctypes.c_double)),
        ("step_size", ctypes.c_double),
        ("knots", ctypes.POINTER(ctypes.c_double)),
    ]


class NcdeTimePlanInfo(ctypes.Structure):
    _fields_ = [
        ("n_t_out", ctypes.c_int32),
        ("n_steps_fwd", ctypes.c_int32),
        ("n_steps_adj", ctypes.c_int32),
        ("stages", ctypes.c_int32),
        ("bytes", ctypes.c_int64),
    ]


class NcdeAdaptiveOptions(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in ("rtol", "atol", "min_step", "max_step", "first_step", "safety", "ifactor", "dfactor")] + \
               [("max_num_steps", ctypes.c_int32), ("trace_capacity", ctypes.c_int32), ("trace", ctypes.POINTER(ctypes.c_double)),
                ("replay_count", ctypes.c_int32), ("reserved_", ctypes.c_int32), ("replay", ctypes.POINTER(ctypes.c_double))]


class NcdeAdaptiveStats(ctypes.Structure):
    _fields_ = [("nfe", ctypes.c_int32), ("n_accepted", ctypes.c_int32), ("n_rejected", ctypes.c_int32), ("reserved_", ctypes.c_int32)]


class NcdeGrads(ctypes.Structure):
    _fields_ = [
        ("grad_z0", _c_float_p),
        ("grad_layer_W", _c_float_p * NCDE_MAX_LAYERS),
        ("grad_layer_b", _c_float_p * NCDE_MAX_LAYERS),
        ("grad_Wo", _c_float_p),
        ("grad_bo", _c_float_p),
        ("grad_Wg", _c_float_p),
        ("grad_bg", _c_float_p),
        ("grad_Wr", _c_float_p),
        ("grad_br", _c_float_p),
    ]


EXPORTS = (
    "ncde_version", "ncde_last_error_string", "ncde_num_outputs", "ncde_workspace_bytes",
    "ncde_kernel_name", "ncde_forward", "ncde_adjoint", "ncde_time_kernel",
    "ncde_prepare_workspace_bytes", "ncde_prepare_linear", "ncde_prepare_cubic", "ncde_prepare_linear_grid", "ncde_prepare_cubic_grid",
    "ncde_stage_record_bytes", "ncde_forward_record", "ncde_backward",
    "ncde_time_plan_build", "ncde_dopri5_workspace_bytes", "ncde_dopri5_forward", "ncde_dopri5_adjoint",
    "ncde_dopri5_record_bytes", "ncde_dopri5_forward_record", "ncde_dopri5_backward", "ncde_dopri5_kernel_name",
    "ncde_coop_status_offset",
)

_LIB = None
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libncde_hip.so")


class NcdeError(RuntimeError):
    pass


def source_fingerprint():
    """sha256 over the kernel sources (csrc/*.hip, csrc/*.h, include/ncde_hip.h) the library is built from: ties a
    committed rocprofv3 counter summary (tools/pmc_summary.py records it) to the kernels bench.py is timing."""
    import glob
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "csrc", "*.hip")) + glob.glob(os.path.join(here, "csrc", "*.h")))
    files.append(os.path.join(os.path.dirname(here), "include", "ncde_hip.h"))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP extension is not built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise NcdeError(
            "libncde_hip.so is not built (%s). Run `python __graft_entry__.py` or "
            "`make -C online-neural-cdes_amd/csrc`; there is no CPU fallback." % LIB_PATH)
    h = ctypes.CDLL(LIB_PATH)
    P = ctypes.POINTER(NcdeProblem)
    G = ctypes.POINTER(NcdeGrads)
    vp, sz = ctypes.c_void_p, ctypes.c_size_t
    h.ncde_version.restype = ctypes.c_int
    h.ncde_last_error_string.restype = ctypes.c_char_p
    h.ncde_num_outputs.argtypes = [P]
    h.ncde_num_outputs.restype = ctypes.c_int
    h.ncde_workspace_bytes.argtypes = [P, ctypes.c_int]
    h.ncde_workspace_bytes.restype = ctypes.c_int64
    h.ncde_coop_status_offset.argtypes = [P, ctypes.c_int]
    h.ncde_coop_status_offset.restype = ctypes.c_int64
    h.ncde_kernel_name.argtypes = [P, ctypes.c_int]
    h.ncde_kernel_name.restype = ctypes.c_char_p
    h.ncde_forward.argtypes = [P, vp, vp, sz, vp]
    h.ncde_forward.restype = ctypes.c_int
    h.ncde_adjoint.argtypes = [P, vp, vp, G, vp, sz, vp]
    h.ncde_adjoint.restype = ctypes.c_int
    h.ncde_time_kernel.argtypes = [P, ctypes.c_int, vp, vp, G, vp, sz, vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
    h.ncde_time_kernel.restype = ctypes.c_int
    h.ncde_stage_record_bytes.argtypes = [P]
    h.ncde_stage_record_bytes.restype = ctypes.c_int64
    h.ncde_forward_record.argtypes = [P, vp, vp, vp, sz, vp]
    h.ncde_forward_record.restype = ctypes.c_int
    h.ncde_backward.argtypes = [P, vp, vp, G, vp, sz, vp]
    h.ncde_backward.restype = ctypes.c_int
    i32 = ctypes.c_int
    h.ncde_prepare_workspace_bytes.argtypes = [i32, i32, i32, i32]
    h.ncde_prepare_workspace_bytes.restype = ctypes.c_int64
    h.ncde_prepare_linear.argtypes = [vp, i32, i32, i32, i32, vp, vp]
    h.ncde_prepare_linear.restype = ctypes.c_int
    h.ncde_prepare_cubic.argtypes = [vp, i32, i32, i32, vp, vp, sz, vp]
    h.ncde_prepare_cubic.restype = ctypes.c_int
    h.ncde_prepare_linear_grid.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
    h.ncde_prepare_linear_grid.restype = ctypes.c_int
    h.ncde_prepare_cubic_grid.argtypes = [vp, vp, i32, i32, i32, vp, vp, sz, vp]
    h.ncde_prepare_cubic_grid.restype = ctypes.c_int
    h.ncde_time_plan_build.argtypes = [P, ctypes.POINTER(NcdeTimeSpec), vp, sz, ctypes.POINTER(NcdeTimePlanInfo)]
    h.ncde_time_plan_build.restype = ctypes.c_int
    TS, AO, AS = ctypes.POINTER(NcdeTimeSpec), ctypes.POINTER(NcdeAdaptiveOptions), ctypes.POINTER(NcdeAdaptiveStats)
    h.ncde_dopri5_workspace_bytes.argtypes = [P, TS, ctypes.c_int]
    h.ncde_dopri5_workspace_bytes.restype = ctypes.c_int64
    h.ncde_dopri5_forward.argtypes = [P, TS, AO, vp, vp, sz, vp, AS]
    h.ncde_dopri5_forward.restype = ctypes.c_int
    h.ncde_dopri5_adjoint.argtypes = [P, TS, AO, vp, vp, G, vp, sz, vp, AS]
    h.ncde_dopri5_adjoint.restype = ctypes.c_int
    h.ncde_dopri5_kernel_name.argtypes = [P, ctypes.c_int]
    h.ncde_dopri5_kernel_name.restype = ctypes.c_char_p
    h.ncde_dopri5_record_bytes.argtypes = [P, TS, AO]
    h.ncde_dopri5_record_bytes.restype = ctypes.c_int64
    h.ncde_dopri5_forward_record.argtypes = [P, TS, AO, vp, vp, sz, vp, sz, vp, AS]
    h.ncde_dopri5_forward_record.restype = ctypes.c_int
    h.ncde_dopri5_backward.argtypes = [P, TS, AO, vp, sz, vp, G, vp, sz, vp]
    h.ncde_dopri5_backward.restype = ctypes.c_int
    if h.ncde_version() != NCDE_ABI_VERSION:
        raise NcdeError("libncde_hip.so ABI %d != binding %d" % (h.ncde_version(), NCDE_ABI_VERSION))
    _LIB = h
    return h


def check(rc, what):
    """Map a negative NcdeStatus to the exception class the reference raises for the same mistake."""
    if rc >= 0:
        return rc
    msg = "%s: %s" % (what, (lib().ncde_last_error_string() or b"").decode())
    if rc == -1:
        raise ValueError(msg)            # malformed input (cf. solver.py:189-190, misc.py:224-226)
    if rc == -2:
        raise NotImplementedError(msg)
    raise NcdeError(msg)
