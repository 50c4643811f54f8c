"""``cdeint``: drop-in for ``torchcde.cdeint`` (/root/reference/modules/torchcde/torchcde/solver.py:140-238)
backed by the fused HIP kernels behind the C-ABI of include/ncde_hip.h.

What the reference does per call -- wrap (X, func) in a vector field, hand it to torchdiffeq's Python
time loop (solvers.py:94-119) and, for the backward pass, to OdeintAdjointMethod (adjoint.py:37-145)
-- happens here in ONE kernel launch per direction.  There is no CPU fallback.  What the fused kernels do
not cover but the reference accepts (an arbitrary ``func``, decreasing output times, gradients of the control
path or of ``t``, non-fp32 tensors, shapes no fused kernel exists for -- hidden widths beyond 256 in training)
runs on the package's own UNFUSED torch-op solver on the GPU (unfused.py) behind a one-time UserWarning naming
the reason; what neither path covers raises NotImplementedError.
"""
import ctypes
import warnings

import numpy as np
import torch

from . import _lib
from .interpolation import LinearInterpolation, NaturalCubicSpline

_FIXED_METHODS = ("euler", "midpoint", "rk4")
_ALL_METHODS = ("dopri8", "dopri5", "bosh3", "fehlberg2", "adaptive_heun", "euler", "midpoint", "rk4",
                "explicit_adams", "implicit_adams", "fixed_adams", "scipy_solver")


class FieldSpec:
    """What the fused kernels need to know about ``func``: the Linear+ReLU stack (entries may repeat
    the same Parameters = a shared layer) and the final Linear (+tanh, viewed [H, C])."""

    def __init__(self, layers, Wo, bo, kind="original", mode="matmul", Wg=None, bg=None, Wr=None, br=None):
        self.layers = list(layers)
        self.Wo, self.bo = Wo, bo
        # variants (src/ncde/vector_fields/gating.py; vector_field_type of solver.py:112-137)
        self.kind, self.mode = kind, mode
        self.Wg, self.bg, self.Wr, self.br = Wg, bg, Wr, br

    def extra_params(self):
        out = []
        if self.kind == "gru":
            out += [self.Wr, self.br]
        if self.kind in ("minimal", "gru"):
            out += [self.Wg, self.bg]
        return out

    def unique_params(self):
        seen, out = set(), []
        for p in [q for wb in self.layers for q in wb] + self.extra_params() + [self.Wo, self.bo]:
            if id(p) not in seen:
                seen.add(id(p))
                out.append(p)
        return out


def _field_spec(func):
    if hasattr(func, "fused_spec"):
        return func.fused_spec()
    raise NotImplementedError(      # (not reached from cdeint: a func without fused_spec() is routed to the unfused solver)
        "a fused kernel call needs `func` to expose fused_spec() (e.g. ncde_amd.OriginalVectorField or ncde_amd.MLPField)")


def _unfused_reason(X, func, z0, t, adjoint, adjoint_params, method=None):
    """Why this call cannot run on the fused kernels (None = it can).  Such calls run on the unfused torch-op solver
    (unfused.py) -- on the GPU: CPU tensors stay refused, there is no CPU fallback."""
    if not torch.is_tensor(z0):
        return None      # tuple-valued z0: refused further down, on either path
    if not hasattr(func, "fused_spec"):
        return "func does not expose fused_spec()"
    if method == "dopri5":      # the adaptive kernels evaluate the original field with the matmul input only
        spec = func.fused_spec()
        if spec.kind != "original" or spec.mode != "matmul":
            return "method='dopri5' with a gated vector field or the evaluate / derivative input"
    # decreasing output times?  X.interval / X.grid_points are tagged and known to increase: no device sync for them
    # (ADVICE round 4); any other tensor is copied to the host ONCE (_host_times, shared with _time_mode / _time_plan).
    if not _is_tagged_time(X, t):
        tv = _host_times(t)
        if tv.dim() == 1 and tv.numel() >= 2 and bool(tv[0] > tv[1]):
            return "decreasing output times"
    if torch.is_tensor(t) and t.requires_grad:
        return "the output times require gradients"
    ap = set(id(q) for q in adjoint_params) if adjoint_params is not None else set()
    for buffer in X.buffers():
        if buffer.requires_grad and (not adjoint or id(buffer) in ap):
            return "the control path requires gradients"
    if z0.is_cuda and (z0.dtype != torch.float32 or X.fused_coeffs.dtype != torch.float32):
        return "tensors are not fp32"
    return None


def _is_tagged_time(X, t):
    """True for X's own interval / grid_points tensors (tagged by interpolation.py; a weakref, so a recycled id() of a dead
    control can never match): their values are known without looking at the device."""
    kind = getattr(t, "_ncde_kind", None)
    owner = getattr(t, "_ncde_owner", None)
    return kind is not None and owner is not None and owner() is X and t.numel() == (X.n_knots if kind == "knots" else 2)


def _host_times(t):
    """Host copy (fp64) of the output times: at most ONE device-to-host copy per tensor version, kept on the tensor object."""
    if not torch.is_tensor(t):
        return torch.as_tensor(t).detach().double()
    hit = getattr(t, "_ncde_host", None)
    if hit is not None and hit[0] == t._version:
        return hit[1]
    tv = t.detach().cpu().double()
    try:
        t._ncde_host = (t._version, tv)
    except Exception:      # (a tensor subclass that refuses attributes)
        pass
    return tv


def _time_mode(X, t):
    """-> _lib.OUT_INTERVAL / OUT_KNOTS.  Tagged tensors from X.interval / X.grid_points avoid a device sync."""
    if X._default_grid and _is_tagged_time(X, t):     # a control on a user knot grid always takes the time plan
        return _lib.OUT_KNOTS if t._ncde_kind == "knots" else _lib.OUT_INTERVAL
    tv = _host_times(t)
    assert tv.dim() == 1, "t must be one dimensional"
    assert (tv[1:] > tv[:-1]).all(), "t must be strictly increasing or decreasing"  # misc.py:336-343
    n = X.n_knots
    if X._default_grid:
        if tv.numel() == n and torch.equal(tv, torch.arange(n, dtype=torch.double)):
            return _lib.OUT_KNOTS
        if tv.numel() == 2 and tv[0] == 0 and tv[1] == n - 1:
            return _lib.OUT_INTERVAL
    return None     # any other increasing t: the general time axis (time plan)


_PLAN_CACHE = {}    # (method, step, t bytes, t dtype, knot bytes, n_knots, device) -> (device plan, info)


def _time_plan(X, t, method, step, device):
    """Build (once per distinct time axis) the table the plan-driven kernels walk: ncde_time_plan_build evaluates the
    reference's grid / stage-time / knot-index arithmetic on the host (solvers.py:78-87, 103-117; one device sync for t)."""
    tt = torch.as_tensor(t).detach()
    f64 = tt.dtype == torch.float64
    tv = np.ascontiguousarray(_host_times(t).numpy())
    kn = None if X._default_grid else np.ascontiguousarray(X._t.detach().cpu().double().numpy())
    key = (method, float(step), tv.tobytes(), f64, None if kn is None else kn.tobytes(), X.n_knots, str(device))
    hit = _PLAN_CACHE.get(key)
    if hit is not None:
        return hit
    p = _lib.NcdeProblem()
    p.abi_version, p.n_knots, p.method = _lib.NCDE_ABI_VERSION, X.n_knots, _lib.METHOD[method]
    dp = ctypes.POINTER(ctypes.c_double)
    ts = _lib.NcdeTimeSpec(n_t=len(tv), time_is_f64=int(f64), t=tv.ctypes.data_as(dp), step_size=float(step),
                           knots=None if kn is None else kn.ctypes.data_as(dp))
    info = _lib.NcdeTimePlanInfo()
    lib = _lib.lib()
    rc = lib.ncde_time_plan_build(ctypes.byref(p), ctypes.byref(ts), None, 0, ctypes.byref(info))
    if rc == -1:
        raise AssertionError(lib.ncde_last_error_string().decode())       # misc.py:336-343 asserts on a non-monotone t
    _lib.check(rc, "ncde_time_plan_build")
    buf = np.zeros(info.bytes // 4, dtype=np.int32)
    _lib.check(lib.ncde_time_plan_build(ctypes.byref(p), ctypes.byref(ts), buf.ctypes.data, buf.nbytes, ctypes.byref(info)),
               "ncde_time_plan_build")
    plan = torch.from_numpy(buf).to(device)
    hit = (plan, (info.n_t_out, info.n_steps_fwd, info.n_steps_adj))
    if len(_PLAN_CACHE) > 64:
        _PLAN_CACHE.clear()
    _PLAN_CACHE[key] = hit
    return hit


def build_problem(coeffs, interp, z0, spec, method, output, flags=0, plan=None):
    """Fill an NcdeProblem from torch tensors (all must stay alive while the call is in flight)."""
    p = _lib.NcdeProblem()
    p.abi_version = _lib.NCDE_ABI_VERSION
    B, H = z0.shape
    p.batch, p.hidden = B, H
    if interp == "linear":
        p.n_knots, p.channels = coeffs.shape[1], coeffs.shape[2]
    else:
        p.n_knots, p.channels = coeffs.shape[1] + 1, coeffs.shape[2] // 4
    p.interp = _lib.INTERP[interp]
    p.method = _lib.METHOD[method]
    p.output = output
    p.flags = flags
    if len(spec.layers) > _lib.NCDE_MAX_LAYERS:
        raise NotImplementedError("at most %d hidden layers" % _lib.NCDE_MAX_LAYERS)
    p.n_layers = len(spec.layers)
    for i, (w, b) in enumerate(spec.layers):
        p.layer_out[i], p.layer_in[i] = w.shape
        p.layer_W[i], p.layer_b[i] = w.data_ptr(), b.data_ptr()
    p.Wo, p.bo = spec.Wo.data_ptr(), spec.bo.data_ptr()
    d_last = spec.layers[-1][0].shape[0] if spec.layers else H
    rows = H * p.channels if spec.mode == "matmul" else H
    if tuple(spec.Wo.shape) != (rows, d_last):
        raise ValueError("final layer must be [%d, %d], got %s" % (rows, d_last, tuple(spec.Wo.shape)))
    p.field_kind, p.field_input = _lib.FIELD_KIND[spec.kind], _lib.FIELD_INPUT[spec.mode]
    if spec.kind != "original":
        if tuple(spec.Wg.shape) != (rows, d_last):
            raise ValueError("sigmoid head must be [%d, %d], got %s" % (rows, d_last, tuple(spec.Wg.shape)))
        p.Wg, p.bg = spec.Wg.data_ptr(), spec.bg.data_ptr()
    if spec.kind == "gru":
        d0 = H if spec.mode == "matmul" else H + p.channels
        if tuple(spec.Wr.shape) != (d0, d0):
            raise ValueError("reset net must be [%d, %d], got %s" % (d0, d0, tuple(spec.Wr.shape)))
        p.Wr, p.br = spec.Wr.data_ptr(), spec.br.data_ptr()
    p.coeffs = coeffs.data_ptr()
    p.coeffs_stride_b, p.coeffs_stride_t = coeffs.stride(0), coeffs.stride(1)
    p.z0 = z0.data_ptr()
    if plan is not None:          # general time axis: (device table, (n_t_out, n_steps_fwd, n_steps_adj))
        p.output = _lib.OUT_TIMES
        p.time_plan = plan[0].data_ptr()
        p.n_t_out, p.n_steps_fwd, p.n_steps_adj = plan[1]
    return p


def _stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_ARENA = {}      # (device index, stream, host thread) -> one byte tensor, grown on demand and reused by every solve issued there
_ARENA_MAX = 32


def clear_workspace_arena():
    """Drop every cached workspace (they are ordinary torch allocations: torch.cuda.empty_cache() can then return the memory)."""
    _ARENA.clear()


def _workspace_bytes(need, device):
    """Scratch for one C-ABI call: an arena per (device, stream, host thread) instead of an allocation per call.  The kernels are
    stream-ordered and a workspace is dead when its call's kernels are, so consecutive calls issued by ONE thread on ONE stream may
    share it.  Two host threads on the same stream could interleave their launches (ctypes drops the GIL) -- A's main kernel, B's
    main kernel, A's reduction reading B's partials -- so the thread is part of the key; entries of dead streams / threads are
    pruned once the table grows."""
    import threading
    need = max(int(need), 256)
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream,
           threading.get_ident())
    buf = _ARENA.get(key)
    if buf is None or buf.numel() < need:
        if len(_ARENA) >= _ARENA_MAX:
            _ARENA.clear()
        buf = torch.empty(need + need // 8, dtype=torch.uint8, device=dev)
        _ARENA[key] = buf
    return buf[:need]


def _workspace(p, pass_, device):
    return _workspace_bytes(_lib.check(_lib.lib().ncde_workspace_bytes(ctypes.byref(p), pass_), "ncde_workspace_bytes"), device)


# ---- the cooperative kernels' status word (include/ncde_hip.h: ncde_coop_status_offset) -------------------------------------------
# A cooperative launch that gives up (bounded spin: its workgroups never all became resident) is re-executed INSIDE the library by the
# per-workgroup kernels, so the tensors this module returns are correct either way.  What is left to the host is to notice -- the call
# took seconds longer -- and to stop asking for the cooperative path: after every call that could launch one, the status word is copied
# (stream-ordered, asynchronously) into a pinned host slot; the slots of finished calls are looked at -- without synchronising
# anything -- at the start of the next fused call, and a set word warns once and adds FLAG_NO_COOP to every later call on that device.
_COOP_SLOTS = {}         # device index -> (pinned int32 ring, next slot)
_COOP_PENDING = []       # (device index, pinned slot view, event)
_COOP_DISABLED = {}      # device index -> number of calls whose cooperative launches gave up


def _coop_track(p, pass_, ws, dev):
    off = _lib.lib().ncde_coop_status_offset(ctypes.byref(p), pass_)
    if off < 0:
        return
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    ring, nxt = _COOP_SLOTS.get(idx, (None, 0))
    if ring is None:
        ring = torch.zeros(256, dtype=torch.int32).pin_memory()
    if len(_COOP_PENDING) >= 192:      # (a caller that never comes back through cdeint: look now, blocking on the oldest)
        coop_status(wait=True)
    slot = ring[nxt:nxt + 1]
    slot.copy_(ws[int(off):int(off) + 4].view(torch.int32), non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    _COOP_PENDING.append((idx, slot, ev))
    _COOP_SLOTS[idx] = (ring, (nxt + 1) % 256)


def coop_status(wait=False):
    """Look at the status words of the fused calls issued so far (``wait=True``: of all of them, synchronising on their events;
    default: of those the GPU has finished).  Returns {device index: number of calls in which a cooperative launch gave up and the
    per-workgroup kernels re-executed the pass}; such a device no longer gets cooperative launches from this process."""
    keep = []
    for idx, slot, ev in _COOP_PENDING:
        if wait:
            ev.synchronize()
        if not ev.query():
            keep.append((idx, slot, ev))
            continue
        if int(slot[0]) != 0:
            if idx not in _COOP_DISABLED:
                warnings.warn("ncde_amd: a cooperative (XCD-wide) kernel launch on cuda:%d timed out -- its workgroups never all became "
                              "resident (another process's kernels on the GPU, a CU mask?).  The pass was re-executed on the per-workgroup "
                              "kernels, so the results are correct; this process now passes FLAG_NO_COOP on that device." % idx, RuntimeWarning)
            _COOP_DISABLED[idx] = _COOP_DISABLED.get(idx, 0) + 1
    _COOP_PENDING[:] = keep
    return dict(_COOP_DISABLED)


def _coop_flags(flags, device):
    """`flags` + FLAG_NO_COOP once a cooperative launch has given up on `device` (looked up without synchronising)."""
    if _COOP_PENDING:
        coop_status()
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return flags | _lib.FLAG_NO_COOP if idx in _COOP_DISABLED else flags


def _check_tensor(x, name):
    if not (x.is_cuda and x.dtype == torch.float32):
        raise NotImplementedError("cdeint fused path needs fp32 tensors on the GPU; %s is %s on %s" % (name, x.dtype, x.device))


class _FusedCdeint(torch.autograd.Function):
    """adjoint=True : forward = ncde_forward, backward = ncde_adjoint (continuous adjoint, adjoint.py:37-145).
    adjoint=False: forward = ncde_forward_record (solution + every stage input), backward = ncde_backward, the exact
    transpose of the discretised solve -- the gradients autograd produces in the reference when ``odeint`` is
    taped (solver.py:224), without a tape."""

    @staticmethod
    def forward(ctx, z0, coeffs, cfg, *params):
        spec = cfg["spec"]
        z0c = z0.detach().contiguous()
        p = build_problem(coeffs, cfg["interp"], z0c, spec, cfg["method"], cfg["output"], cfg["flags"], cfg["plan"])
        if cfg["plan"] is not None:
            n_out = cfg["plan"][1][0]
        else:
            n_out = coeffs.shape[1] + (1 if cfg["interp"] == "cubic" else 0) if cfg["output"] == _lib.OUT_KNOTS else 2
        out = torch.empty(z0.shape[0], n_out, z0.shape[1], dtype=torch.float32, device=z0.device)
        record = (not cfg["adjoint"]) and cfg.get("needs_grad", True) and any(ctx.needs_input_grad)
        stages = None
        with torch.cuda.device(z0.device):   # the C-ABI launches on the calling thread's current device / stream
            ws = _workspace(p, 0, z0.device)
            if record:
                nbytes = _lib.check(_lib.lib().ncde_stage_record_bytes(ctypes.byref(p)), "ncde_stage_record_bytes")
                stages = torch.empty(max(int(nbytes) // 4, 1), dtype=torch.float32, device=z0.device)
                rc = _lib.lib().ncde_forward_record(ctypes.byref(p), out.data_ptr(), stages.data_ptr(), ws.data_ptr(),
                                                    ws.numel(), _stream_ptr())
            else:
                rc = _lib.lib().ncde_forward(ctypes.byref(p), out.data_ptr(), ws.data_ptr(), ws.numel(), _stream_ptr())
            if rc >= 0:
                _coop_track(p, 0, ws, z0.device)
        _lib.check(rc, "ncde_forward_record" if record else "ncde_forward")
        ctx.cfg = cfg
        ctx.coeffs = coeffs
        ctx.recorded = record
        if record:
            ctx.save_for_backward(out, stages, *params)
        else:
            ctx.save_for_backward(out, *params)
        ctx.z0_shape = z0.shape
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable     # the kernels are not themselves differentiable: no create_graph
    def backward(ctx, grad_out):
        cfg = ctx.cfg
        if ctx.recorded:
            out, stages, *params = ctx.saved_tensors
        else:
            out, *params = ctx.saved_tensors
        spec, coeffs = cfg["spec"], ctx.coeffs
        dev = out.device
        grad_out = grad_out.contiguous().float()
        z0 = out[:, 0]     # only its shape matters here: the backward kernels never read z0 (row 0 of `out` is a valid pointer)
        p = build_problem(coeffs, cfg["interp"], z0, spec, cfg["method"], cfg["output"], cfg["flags"], cfg["plan"])
        uniq = spec.unique_params()
        gbuf = {id(q): torch.empty_like(q, memory_format=torch.contiguous_format) for q in uniq}
        g = _lib.NcdeGrads()
        grad_z0 = torch.empty(ctx.z0_shape, dtype=torch.float32, device=dev)
        g.grad_z0 = grad_z0.data_ptr()
        for i, (w, b) in enumerate(spec.layers):
            g.grad_layer_W[i], g.grad_layer_b[i] = gbuf[id(w)].data_ptr(), gbuf[id(b)].data_ptr()
        g.grad_Wo, g.grad_bo = gbuf[id(spec.Wo)].data_ptr(), gbuf[id(spec.bo)].data_ptr()
        if spec.kind != "original":
            g.grad_Wg, g.grad_bg = gbuf[id(spec.Wg)].data_ptr(), gbuf[id(spec.bg)].data_ptr()
        if spec.kind == "gru":
            g.grad_Wr, g.grad_br = gbuf[id(spec.Wr)].data_ptr(), gbuf[id(spec.br)].data_ptr()
        with torch.cuda.device(dev):
            if ctx.recorded:
                ws = _workspace(p, 2, dev)
                rc = _lib.lib().ncde_backward(ctypes.byref(p), stages.data_ptr(), grad_out.data_ptr(), ctypes.byref(g),
                                              ws.data_ptr(), ws.numel(), _stream_ptr())
            else:
                ws = _workspace(p, 1, dev)
                rc = _lib.lib().ncde_adjoint(ctypes.byref(p), out.data_ptr(), grad_out.data_ptr(), ctypes.byref(g),
                                             ws.data_ptr(), ws.numel(), _stream_ptr())
            if rc >= 0:
                _coop_track(p, 2 if ctx.recorded else 1, ws, dev)
        _lib.check(rc, "ncde_backward" if ctx.recorded else "ncde_adjoint")
        if not ctx.recorded and cfg["func"] is not None and hasattr(cfg["func"], "nfe"):
            cfg["func"].nfe += cfg["nfe_adjoint"]   # the adjoint sweep re-evaluates f (base.py:90); autograd does not
        grads = []
        keep = cfg["adjoint_param_ids"]     # adjoint_params of odeint_adjoint (adjoint.py:176-183): others get no gradient
        for q, needs in zip(params, ctx.needs_input_grad[3:]):
            grads.append(gbuf[id(q)] if needs and (keep is None or id(q) in keep) else None)
        return (grad_z0 if ctx.needs_input_grad[0] else None, None, None, *grads)


class _AdaptiveSpec:
    """Host description of one dopri5 call: output times / knots as host doubles + NcdeAdaptiveOptions."""

    def __init__(self, X, t, rtol, atol, options):
        tt = torch.as_tensor(t).detach()
        self.tv = np.ascontiguousarray(tt.cpu().double().numpy())
        assert self.tv.ndim == 1, "t must be one dimensional"
        assert (self.tv[1:] > self.tv[:-1]).all() or (self.tv[1:] < self.tv[:-1]).all(), "t must be strictly increasing or decreasing"
        if self.tv[1] < self.tv[0]:
            raise NotImplementedError("decreasing output times are outside the fused path")
        self.kn = None if X._default_grid else np.ascontiguousarray(X._t.detach().cpu().double().numpy())
        dp = ctypes.POINTER(ctypes.c_double)
        self.ts = _lib.NcdeTimeSpec(n_t=len(self.tv), time_is_f64=1, t=self.tv.ctypes.data_as(dp), step_size=1.0,
                                    knots=None if self.kn is None else self.kn.ctypes.data_as(dp))
        o = _lib.NcdeAdaptiveOptions()
        o.rtol, o.atol = float(rtol), float(atol)
        for k in ("min_step", "max_step", "first_step", "safety", "ifactor", "dfactor"):
            if options.get(k) is not None:
                setattr(o, k, float(options[k]))
        if options.get("max_num_steps") is not None:
            o.max_num_steps = int(options["max_num_steps"])
        self.trace = None
        if options.get("_trace"):           # diagnostics (tests): keep the step sequence (t0, dt, accepted, error ratio)
            self.trace = np.zeros((int(options["_trace"]), 4), dtype=np.float64)
            o.trace_capacity = self.trace.shape[0]
            o.trace = self.trace.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        self.replay = None
        if options.get("_replay") is not None:      # verification (tests): force a recorded step sequence, rows of (dt, accepted)
            self.replay = np.ascontiguousarray(np.asarray(options["_replay"], dtype=np.float64).reshape(-1, 2))
            o.replay_count = self.replay.shape[0]
            o.replay = self.replay.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        self.opt = o


class _FusedDopri5(torch.autograd.Function):
    """method='dopri5': forward = ncde_dopri5_forward; backward = ncde_dopri5_adjoint (one adaptive reverse solve per output
    interval, adjoint.py:37-145).  Both calls synchronise: the number of attempts depends on the data."""

    @staticmethod
    def forward(ctx, z0, coeffs, cfg, *params):
        spec, ad = cfg["spec"], cfg["adaptive"]
        z0c = z0.detach().contiguous()
        p = build_problem(coeffs, cfg["interp"], z0c, spec, "rk4", _lib.OUT_INTERVAL, cfg["flags"])
        out = torch.empty(z0.shape[0], len(ad.tv), z0.shape[1], dtype=torch.float32, device=z0.device)
        lib = _lib.lib()
        stats = _lib.NcdeAdaptiveStats()
        with torch.cuda.device(z0.device):
            need = _lib.check(lib.ncde_dopri5_workspace_bytes(ctypes.byref(p), ctypes.byref(ad.ts), 0), "ncde_dopri5_workspace_bytes")
            ws = _workspace_bytes(int(need), z0.device)
            rc = lib.ncde_dopri5_forward(ctypes.byref(p), ctypes.byref(ad.ts), ctypes.byref(ad.opt), out.data_ptr(), ws.data_ptr(),
                                         ws.numel(), _stream_ptr(), ctypes.byref(stats))
        if rc == -1:
            raise AssertionError(lib.ncde_last_error_string().decode())      # the reference asserts (rk_common.py:232-233, 195)
        _lib.check(rc, "ncde_dopri5_forward")
        cfg["stats_forward"] = (stats.nfe, stats.n_accepted, stats.n_rejected)
        if ad.trace is not None and cfg["func"] is not None:
            cfg["func"].dopri5_trace = ad.trace[:stats.n_accepted + stats.n_rejected].copy()
        if cfg["func"] is not None and hasattr(cfg["func"], "nfe"):
            cfg["func"].nfe += stats.nfe
        ctx.cfg, ctx.coeffs, ctx.z0_shape = cfg, coeffs, z0.shape
        ctx.save_for_backward(out, *params)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        cfg = ctx.cfg
        out, *params = ctx.saved_tensors
        spec, ad = cfg["spec"], cfg["adaptive_backward"]
        dev = out.device
        grad_out = grad_out.contiguous().float()
        p = build_problem(ctx.coeffs, cfg["interp"], out[:, 0], spec, "rk4", _lib.OUT_INTERVAL, cfg["flags"])
        uniq = spec.unique_params()
        gbuf = {id(q): torch.empty_like(q, memory_format=torch.contiguous_format) for q in uniq}
        g = _lib.NcdeGrads()
        grad_z0 = torch.empty(ctx.z0_shape, dtype=torch.float32, device=dev)
        g.grad_z0 = grad_z0.data_ptr()
        # the adjoint's parameters (adjoint.py:176-189): those that require a gradient and, if given, are listed in adjoint_params; the
        # others get a NULL destination = they are not part of the augmented state, hence not of the mixed error norm
        keep = cfg["adjoint_param_ids"]
        live = {id(q) for q, needs in zip(params, ctx.needs_input_grad[3:]) if needs and (keep is None or id(q) in keep)}
        ptr = lambda q: gbuf[id(q)].data_ptr() if id(q) in live else None      # noqa: E731
        for i, (w, b) in enumerate(spec.layers):
            g.grad_layer_W[i], g.grad_layer_b[i] = ptr(w), ptr(b)
        g.grad_Wo, g.grad_bo = ptr(spec.Wo), ptr(spec.bo)
        lib = _lib.lib()
        stats = _lib.NcdeAdaptiveStats()
        with torch.cuda.device(dev):
            need = _lib.check(lib.ncde_dopri5_workspace_bytes(ctypes.byref(p), ctypes.byref(ad.ts), 1), "ncde_dopri5_workspace_bytes")
            ws = _workspace_bytes(int(need), dev)
            rc = lib.ncde_dopri5_adjoint(ctypes.byref(p), ctypes.byref(ad.ts), ctypes.byref(ad.opt), out.data_ptr(), grad_out.data_ptr(),
                                         ctypes.byref(g), ws.data_ptr(), ws.numel(), _stream_ptr(), ctypes.byref(stats))
        if rc == -1:
            raise AssertionError(lib.ncde_last_error_string().decode())
        _lib.check(rc, "ncde_dopri5_adjoint")
        cfg["stats_backward"] = (stats.nfe, stats.n_accepted, stats.n_rejected)
        if cfg["func"] is not None and hasattr(cfg["func"], "nfe"):
            cfg["func"].nfe += stats.nfe
        if ad.trace is not None and cfg["func"] is not None:      # diagnostics (tests): the reverse solve's own step sequence
            cfg["func"].dopri5_trace_backward = ad.trace[:stats.n_accepted + stats.n_rejected].copy()
        grads = [gbuf[id(q)] if id(q) in live else None for q in params]
        return (grad_z0 if ctx.needs_input_grad[0] else None, None, None, *grads)


class _FusedDopri5Taped(torch.autograd.Function):
    """method='dopri5' with adjoint=False (torchdiffeq.odeint under autograd, torchcde/solver.py:224-225): forward =
    ncde_dopri5_forward_record (the adaptive solve keeping a record of its accepted steps), backward = ncde_dopri5_backward
    (reverse-mode sweep over that record incl. the gradient of the first step size).  The backward evaluates no vector field
    the reference would count: func.nfe grows by the forward's evaluations only, as under autograd."""

    @staticmethod
    def forward(ctx, z0, coeffs, cfg, *params):
        spec, ad = cfg["spec"], cfg["adaptive"]
        z0c = z0.detach().contiguous()
        p = build_problem(coeffs, cfg["interp"], z0c, spec, "rk4", _lib.OUT_INTERVAL, cfg["flags"])
        out = torch.empty(z0.shape[0], len(ad.tv), z0.shape[1], dtype=torch.float32, device=z0.device)
        lib = _lib.lib()
        stats = _lib.NcdeAdaptiveStats()
        # nothing to differentiate (torch.no_grad(), frozen inputs): the plain solve, no step record at all
        taped = cfg.get("needs_grad", True) and any(ctx.needs_input_grad)
        rec = None
        with torch.cuda.device(z0.device):
            need = _lib.check(lib.ncde_dopri5_workspace_bytes(ctypes.byref(p), ctypes.byref(ad.ts), 0), "ncde_dopri5_workspace_bytes")
            ws = _workspace_bytes(int(need), z0.device)
            if not taped:
                rc = lib.ncde_dopri5_forward(ctypes.byref(p), ctypes.byref(ad.ts), ctypes.byref(ad.opt), out.data_ptr(), ws.data_ptr(),
                                             ws.numel(), _stream_ptr(), ctypes.byref(stats))
            else:
                # The record holds one start state per ACCEPTED step; its default size is a guess when the options give no min_step
                # (ncde_dopri5_record_bytes).  A solve that outgrows it reports NCDE_ERR_WORKSPACE: repeat it with twice the record
                # -- the forward is deterministic -- where the reference's autograd tape would simply have grown.
                rbytes = int(_lib.check(lib.ncde_dopri5_record_bytes(ctypes.byref(p), ctypes.byref(ad.ts), ctypes.byref(ad.opt)), "ncde_dopri5_record_bytes"))
                rbytes = int(cfg.get("record_bytes") or rbytes)
                for _ in range(8):
                    rec = torch.empty(rbytes, dtype=torch.uint8, device=z0.device)
                    rc = lib.ncde_dopri5_forward_record(ctypes.byref(p), ctypes.byref(ad.ts), ctypes.byref(ad.opt), out.data_ptr(), rec.data_ptr(),
                                                        rec.numel(), ws.data_ptr(), ws.numel(), _stream_ptr(), ctypes.byref(stats))
                    if rc != -3 or b"record" not in (lib.ncde_last_error_string() or b""):
                        break
                    rec = None
                    rbytes *= 2
        if rc == -1:
            raise AssertionError(lib.ncde_last_error_string().decode())      # the reference asserts (rk_common.py:232-233, 195)
        _lib.check(rc, "ncde_dopri5_forward_record" if taped else "ncde_dopri5_forward")
        cfg["stats_forward"] = (stats.nfe, stats.n_accepted, stats.n_rejected)
        if ad.trace is not None and cfg["func"] is not None:
            cfg["func"].dopri5_trace = ad.trace[:stats.n_accepted + stats.n_rejected].copy()
        if cfg["func"] is not None and hasattr(cfg["func"], "nfe"):
            cfg["func"].nfe += stats.nfe
        ctx.cfg, ctx.coeffs, ctx.z0_shape = cfg, coeffs, z0.shape
        if taped:
            ctx.save_for_backward(z0c, rec, *params)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        cfg = ctx.cfg
        z0c, rec, *params = ctx.saved_tensors
        spec, ad = cfg["spec"], cfg["adaptive"]
        dev = z0c.device
        grad_out = grad_out.contiguous().float()
        p = build_problem(ctx.coeffs, cfg["interp"], z0c, spec, "rk4", _lib.OUT_INTERVAL, cfg["flags"])
        uniq = spec.unique_params()
        gbuf = {id(q): torch.empty_like(q, memory_format=torch.contiguous_format) for q in uniq}
        g = _lib.NcdeGrads()
        grad_z0 = torch.empty(ctx.z0_shape, dtype=torch.float32, device=dev)
        g.grad_z0 = grad_z0.data_ptr()
        for i, (w, b) in enumerate(spec.layers):
            g.grad_layer_W[i], g.grad_layer_b[i] = gbuf[id(w)].data_ptr(), gbuf[id(b)].data_ptr()
        g.grad_Wo, g.grad_bo = gbuf[id(spec.Wo)].data_ptr(), gbuf[id(spec.bo)].data_ptr()
        lib = _lib.lib()
        with torch.cuda.device(dev):
            need = _lib.check(lib.ncde_dopri5_workspace_bytes(ctypes.byref(p), ctypes.byref(ad.ts), 2), "ncde_dopri5_workspace_bytes")
            ws = _workspace_bytes(int(need), dev)
            rc = lib.ncde_dopri5_backward(ctypes.byref(p), ctypes.byref(ad.ts), ctypes.byref(ad.opt), rec.data_ptr(), rec.numel(),
                                          grad_out.data_ptr(), ctypes.byref(g), ws.data_ptr(), ws.numel(), _stream_ptr())
        _lib.check(rc, "ncde_dopri5_backward")
        grads = [gbuf[id(q)] if needs else None for q, needs in zip(params, ctx.needs_input_grad[3:])]
        return (grad_z0 if ctx.needs_input_grad[0] else None, None, None, *grads)


_DOPRI5_OPTIONS = ("min_step", "max_step", "first_step", "safety", "ifactor", "dfactor", "max_num_steps", "_trace", "_replay",
                   "_record_bytes")      # _record_bytes: initial size of the step record of adjoint=False (default: the library's estimate)


def _run_unfused(reason, X, func, z0, t, adjoint, vector_field_type, method, options, adjoint_params, rtol, atol, adjoint_rtol,
                 adjoint_atol, adjoint_options, warned=False):
    """Outside the fused kernels: the own unfused torch-op solver, same algorithms (unfused.py), on the tensors' device.
    warned: the caller has already been through the fused path's argument checks (a shape without a fused kernel is discovered after
    them), so the reference's warnings -- unexpected arguments, control buffers requiring gradients -- are not repeated."""
    from . import unfused
    options = dict(options)
    if not z0.is_cuda:
        raise NotImplementedError("cdeint needs tensors on the GPU (there is no CPU fallback); z0 is on %s" % z0.device)
    adaptive_cfg = None
    if method == "dopri5":      # adaptive dopri5 on the unfused path too (same options as the fused one)
        if options.pop("norm", None) is not None:
            raise NotImplementedError("a custom error norm is not supported (the reference's rms / mixed norms are built in)")
        aopt = {k: options.pop(k) for k in list(options) if k in _DOPRI5_OPTIONS and not k.startswith("_")}
        for k in [k for k in options if k.startswith("_")]:
            options.pop(k)
        bopt = dict(aopt) if adjoint_options is None else {k: v for k, v in adjoint_options.items() if k in _DOPRI5_OPTIONS and not k.startswith("_")}
        adaptive_cfg = {"rtol": rtol, "atol": atol, "options": aopt, "adjoint_rtol": rtol if adjoint_rtol is None else adjoint_rtol,
                        "adjoint_atol": atol if adjoint_atol is None else adjoint_atol, "adjoint_options": bopt}
    step = options.pop("step_size", None)
    if "grid_constructor" in options:
        raise NotImplementedError("options['grid_constructor'] is not supported; give options={'step_size': h}")
    options.pop("perturb", None)
    for k in options:
        if not warned:
            warnings.warn("cdeint: Unexpected arguments {}".format({k: options[k]}))
    if adjoint and not warned:
        ids = set(id(q) for q in adjoint_params) if adjoint_params is not None else set()
        for buffer in X.buffers():
            if buffer.requires_grad and id(buffer) not in ids:      # the reference's warning (solver.py:207-221)
                warnings.warn("One of the inputs to the control path X requires gradients but is not listed in "
                              "`options['adjoint_params']`. It will not receive a gradient when using the adjoint method.")
    unfused.warn_once(reason)
    if torch.is_tensor(step):
        step = step.item()
    return unfused.cdeint_unfused(X, func, z0, t, adjoint, vector_field_type, method, step, adjoint_params, adaptive_cfg)


def _no_kernel_reason(p, passes, dopri5_ts=None):
    """None if the library has a fused kernel for every pass in `passes` (0 forward, 1 continuous adjoint, 2 exact discrete
    backward) of problem `p`, else a sentence naming the shape.  Asked BEFORE the forward runs: a model
    the reference trains -- hidden_dim up to 256, hidden_hidden_dim up to 196, configurations.json5:34-35 -- must not pass its
    forward and then fail inside loss.backward()."""
    lib = _lib.lib()
    for ps in passes:
        if dopri5_ts is not None:
            rc = lib.ncde_dopri5_workspace_bytes(ctypes.byref(p), ctypes.byref(dopri5_ts), ps)
        else:
            rc = lib.ncde_workspace_bytes(ctypes.byref(p), ps)
        if rc == -2:      # NCDE_ERR_UNSUPPORTED (anything else is reported by the call itself)
            what = ("forward", "continuous adjoint", "exact discrete backward")[ps]
            widths = [int(p.layer_out[l]) for l in range(p.n_layers)]
            return "no fused %s kernel for hidden=%d, layer widths %s, channels=%d: %s" % (
                what, p.hidden, widths, p.channels, (lib.ncde_last_error_string() or b"").decode())
    return None


def cdeint(X, func, z0, t, adjoint=True, vector_field_type="matmul", **kwargs):
    r"""Solve ``z_t = z_{t_0} + \int f(z_s) dX_s``; returns ``[batch, len(t), hidden]`` like the reference.

    Same arguments as ``torchcde.cdeint`` (solver.py:140): X a LinearInterpolation / NaturalCubicSpline (default integer grid
    or a user knot grid), ``func`` any ``nn.Module (t, z) -> [..., H, C]``, ``method`` in {euler, midpoint, rk4} (with
    ``options={'step_size': h}``) or dopri5, CUDA tensors, ``t`` any monotone times.
    The FUSED kernels run when ``func`` exposes ``fused_spec()`` (the package's vector fields), tensors are fp32, ``t``
    increases and nothing upstream of the control path needs a gradient: the reference's NeuralCDE setting -- default grid,
    step 1, t = X.interval or X.grid_points -- on the shape-specialised / batch-tiled kernels, any other time axis on their
    plan-driven forms.  Every other request the reference accepts -- and shapes no fused kernel covers (training with hidden
    widths > 256, checked here before the forward) -- runs on the unfused torch-op solver on the GPU (unfused.py), with one
    UserWarning per reason.
    """
    if vector_field_type not in ("matmul", "evaluate", "derivative"):
        raise ValueError("vector_field_type string not recognised")
    kwargs.setdefault("atol", 1e-6)
    kwargs.setdefault("rtol", 1e-4)
    method = kwargs.pop("method", None)
    options = dict(kwargs.pop("options", None) or {})
    options_in = dict(options)      # (as given: what the unfused solver is handed if the call ends up there)
    flags = kwargs.pop("kernel_flags", 0)
    adjoint_params = kwargs.pop("adjoint_params", None)
    atol, rtol = kwargs.pop("atol"), kwargs.pop("rtol")
    adjoint_rtol, adjoint_atol = kwargs.pop("adjoint_rtol", None), kwargs.pop("adjoint_atol", None)
    adjoint_options = kwargs.pop("adjoint_options", None)
    adjoint_method = kwargs.pop("adjoint_method", None)
    for k in kwargs:
        warnings.warn("cdeint: Unexpected arguments {}".format({k: kwargs[k]}))  # misc.py:9-11
    if method is None:
        method = "dopri5"
    if method not in _ALL_METHODS:
        raise ValueError('Invalid method "{}". Must be one of {}'.format(method, '{"' + '", "'.join(_ALL_METHODS) + '"}.'))
    if method not in _FIXED_METHODS and method != "dopri5":
        raise NotImplementedError("method '%s': the fixed-step solvers %s and adaptive dopri5 are implemented" % (method, _FIXED_METHODS))
    if adjoint_method is not None and adjoint_method != method:
        raise NotImplementedError("adjoint_method != method is not implemented")
    if not isinstance(X, (LinearInterpolation, NaturalCubicSpline)):
        raise NotImplementedError("X must be ncde_amd.LinearInterpolation or ncde_amd.NaturalCubicSpline")

    z0_in = z0      # (z0 is flattened further down; the unfused solver takes it as given)

    def unfused_(reason, warned=False):
        return _run_unfused(reason, X, func, z0_in, t, adjoint, vector_field_type, method, options_in, adjoint_params, rtol, atol,
                            adjoint_rtol, adjoint_atol, adjoint_options, warned)

    reason = _unfused_reason(X, func, z0, t, adjoint, adjoint_params, method)
    if reason is not None:
        return unfused_(reason)
    adaptive = method == "dopri5"
    if adaptive:
        for k in list(options):
            if k not in _DOPRI5_OPTIONS and k != "norm":
                warnings.warn("cdeint: Unexpected arguments {}".format({k: options.pop(k)}))
        if options.pop("norm", None) is not None:
            raise NotImplementedError("a custom error norm is outside the fused path (the reference's rms / mixed norms are built in)")
        options.setdefault("step_size", 1.0)      # unused by the adaptive solver; keeps the common argument checks below uniform
    step = options.pop("step_size", None)
    if "grid_constructor" in options:
        raise NotImplementedError("options['grid_constructor'] is outside the fused path; give options={'step_size': h}")
    if step is None:
        raise NotImplementedError("options={'step_size': h} is required on the fused path (without it torchdiffeq steps "
                                  "from output time to output time, solvers.py:69-71)")
    if torch.is_tensor(step):
        step = step.item()
    if not float(step) > 0.0:
        raise ValueError("step_size must be positive")
    if options.pop("perturb", False):
        warnings.warn("cdeint: options['perturb'] is ignored by the fused fixed-step kernels (stage times are exact knots/fractions)")
    ad_options = {k: options.pop(k) for k in list(options) if adaptive and k in _DOPRI5_OPTIONS}
    for k in options:
        warnings.warn("cdeint: Unexpected arguments {}".format({k: options[k]}))
    if not torch.is_tensor(z0):
        raise NotImplementedError("tuple-valued z0 is outside the fused path")
    batch_shape = z0.shape[:-1]          # any number of batch dimensions, as the reference allows (flattened for the kernels)
    if z0.dim() < 1:
        raise ValueError("z0 must have a hidden dimension")
    ap = set(id(q) for q in adjoint_params) if adjoint_params is not None else None
    for buffer in X.buffers():
        if not buffer.requires_grad:
            continue
        if adjoint and (ap is None or id(buffer) not in ap):       # the reference's warning (solver.py:207-221)
            warnings.warn("One of the inputs to the control path X requires gradients but is not listed in "
                          "`options['adjoint_params']`. It will not receive a gradient when using the adjoint method.")
        else:
            # adjoint=False tapes the solve in the reference, so the path's coefficients (and whatever produced them)
            # would receive gradients; the fused backward has no dL/dcoeffs -- refuse rather than drop it silently
            raise NotImplementedError("cdeint: gradients with respect to the control path's coefficients are not "
                                      "implemented on the fused path; detach() the coefficients")
    coeffs = X.fused_coeffs
    if coeffs.dim() < 2 or tuple(coeffs.shape[:-2]) != tuple(batch_shape):
        raise ValueError("batch dimensions of X %s != batch dimensions of z0 %s" % (tuple(coeffs.shape[:-2]), tuple(batch_shape)))
    _check_tensor(z0, "z0")
    _check_tensor(coeffs, "coeffs")
    if len(batch_shape) != 1:
        z0 = z0.reshape(-1, z0.shape[-1])
        coeffs = coeffs.reshape(-1, coeffs.shape[-2], coeffs.shape[-1])
    if coeffs.stride(2) != 1:
        coeffs = coeffs.contiguous()
    if coeffs.shape[0] != z0.shape[0]:
        raise ValueError("batch of X (%d) != batch of z0 (%d)" % (coeffs.shape[0], z0.shape[0]))
    spec = _field_spec(func)
    if spec.mode != vector_field_type:
        raise ValueError("vector_field_type='%s' but func was built for '%s'" % (vector_field_type, spec.mode))
    uniq = spec.unique_params()
    for q in uniq:
        _check_tensor(q, "a vector-field parameter")
        if not q.is_contiguous():
            raise NotImplementedError("vector-field parameters must be contiguous")
    if adjoint and ap is not None:
        ap = ap & set(id(q) for q in uniq)
    else:
        ap = None
    # does anything need a gradient?  (torch.no_grad() evaluation with trainable parameters does not: needs_input_grad inside an
    # autograd.Function mirrors requires_grad whatever the grad mode, so the decision is taken here -- ADVICE round 4)
    needs_grad = torch.is_grad_enabled() and (z0.requires_grad or any(q.requires_grad for q in uniq))
    # (a family pinned by a development flag keeps the old contract -- the forward runs, a missing backward kernel is reported when a
    # backward is actually asked for: the caller named the kernels, rerouting to the unfused solver would not be what was asked)
    pinned = flags & (_lib.FLAG_FORCE_GENERIC | _lib.FLAG_FORCE_FAST | _lib.FLAG_FORCE_TILED)
    passes = (0,) + (((1,) if adjoint else (2,)) if (needs_grad and not pinned) else ())
    if adaptive:
        if spec.kind != "original" or spec.mode != "matmul":
            raise NotImplementedError("method='dopri5' runs the original vector field with the matmul input only")
        bopt = dict(ad_options) if adjoint_options is None else {k: v for k, v in adjoint_options.items() if k in _DOPRI5_OPTIONS}
        cfg = {"spec": spec, "interp": X.interp_name, "flags": flags, "func": func, "adjoint_param_ids": ap, "needs_grad": needs_grad,
               "adaptive": _AdaptiveSpec(X, t, rtol, atol, ad_options), "record_bytes": ad_options.get("_record_bytes"),
               "adaptive_backward": _AdaptiveSpec(X, t, rtol if adjoint_rtol is None else adjoint_rtol,
                                                  atol if adjoint_atol is None else adjoint_atol, bopt)}
        why = _no_kernel_reason(build_problem(coeffs, X.interp_name, z0.detach(), spec, "rk4", _lib.OUT_INTERVAL, flags), passes, cfg["adaptive"].ts)
        if why is not None:
            return unfused_(why, warned=True)
        out = (_FusedDopri5 if adjoint else _FusedDopri5Taped).apply(z0, coeffs.detach(), cfg, *uniq)
        if len(batch_shape) != 1:
            out = out.reshape(*batch_shape, out.shape[-2], out.shape[-1])
        return out
    flags = _coop_flags(flags, z0.device)
    output = _time_mode(X, t) if float(step) == 1.0 else None
    stages = {"euler": 1, "midpoint": 2, "rk4": 4}[method]
    plan = None
    if output is None:
        tq = torch.as_tensor(t)
        assert tq.dim() == 1, "t must be one dimensional"
        plan = _time_plan(X, tq, method, step, z0.device)
        output = _lib.OUT_TIMES
        nfe, nfe_adj = stages * plan[1][1], stages * plan[1][2]
    else:
        nfe = nfe_adj = stages * (X.n_knots - 1)
    why = _no_kernel_reason(build_problem(coeffs, X.interp_name, z0.detach(), spec, method, output, flags, plan), passes)
    if why is not None:
        return unfused_(why, warned=True)
    cfg = {"spec": spec, "interp": X.interp_name, "method": method, "output": output, "flags": flags, "plan": plan, "needs_grad": needs_grad,
           "adjoint": bool(adjoint), "func": func, "nfe_per_solve": nfe, "nfe_adjoint": nfe_adj, "adjoint_param_ids": ap}
    out = _FusedCdeint.apply(z0, coeffs.detach(), cfg, *uniq)
    if hasattr(func, "nfe"):
        func.nfe += nfe
    if len(batch_shape) != 1:
        out = out.reshape(*batch_shape, out.shape[-2], out.shape[-1])
    return out
