"""The UNFUSED solver: ``cdeint`` for requests the fused HIP kernels do not cover -- an arbitrary ``func`` (any
``nn.Module (t, z) -> [..., H, C]``, /root/reference/modules/torchcde/torchcde/solver.py:102-137), decreasing output times
(torchdiffeq/_impl/misc.py:262-271), gradients with respect to the control path or the output times, non-fp32 tensors.

It is an own restatement in torch ops of the same algorithms the kernels implement -- the fixed-grid solvers of
torchdiffeq (solvers.py:78-119, 166-172; fixed_grid.py:6-29; the 3/8 rule of rk_common.py:106-114), adaptive dopri5
(rk_common.py:41-86, 216-305; dopri5.py:5-36; misc.py:33-103; interp.py:4-61 -- round 4: the gated fields and the evaluate /
derivative inputs with ``method='dopri5'``, which the fused adaptive kernels do not cover) and the continuous adjoint
(adjoint.py:37-145: one reverse solve of (vjp_t, y, a, g_theta) per output interval, stage VJPs by ``torch.autograd.grad``) --
running on the device the tensors live on (the GPU: ``cdeint`` refuses CPU tensors, there is no CPU fallback), one torch kernel
per elementary operation.  It is the reference's speed class, not the fused kernels'; ``cdeint`` warns once per reason when it
takes this path.
"""
import math
import warnings

import torch

_WARNED = set()


def warn_once(reason):
    if reason not in _WARNED:
        _WARNED.add(reason)
        warnings.warn("cdeint: running the unfused torch-op solver (%s); the fused MI355X kernels need a vector field exposing "
                      "fused_spec(), fp32 CUDA tensors, increasing output times and a detached control path" % reason, UserWarning)


class ControlledField:
    """g(t, z) = f(t, z) dX/dt(t)  (matmul), or f(t, [z, X(t)]) / f(t, [z, dX/dt(t)])  -- solver.py:112-137."""

    def __init__(self, X, func, mode):
        self.X, self.func, self.mode = X, func, mode

    def __call__(self, t, z):
        if self.mode == "matmul":
            dX = self.X.derivative(t)
            return (self.func(t, z) @ dX.unsqueeze(-1)).squeeze(-1)
        return self.func(t, torch.cat([z, getattr(self.X, self.mode)(t)], dim=-1))


# ---- fixed-grid solvers on a TUPLE state (the adjoint integrates (vjp_t, y, a, g_theta...) together) -------------------------

def _axpy(ys, alpha, ks):
    return tuple(y + alpha * k for y, k in zip(ys, ks))


def _step(f, method, t0, dt, t1, y0):
    """Increment dy of one step (fixed_grid.py:6-29, rk_common.py:106-114)."""
    if method == "euler":
        return tuple(dt * k for k in f(t0, y0))
    if method == "midpoint":
        half = 0.5 * dt
        ymid = _axpy(y0, half, f(t0, y0))
        return tuple(dt * k for k in f(t0 + half, ymid))
    if method == "rk4":      # the 3/8 rule
        third = 1.0 / 3.0
        k1 = f(t0, y0)
        k2 = f(t0 + dt * third, tuple(y + dt * k * third for y, k in zip(y0, k1)))
        k3 = f(t0 + dt * (2.0 * third), tuple(y + dt * (b - a * third) for y, a, b in zip(y0, k1, k2)))
        k4 = f(t1, tuple(y + dt * (a - b + c) for y, a, b, c in zip(y0, k1, k2, k3)))
        return tuple((a + 3.0 * (b + c) + d) * dt * 0.125 for a, b, c, d in zip(k1, k2, k3, k4))
    raise ValueError(method)


def _grid(t, step_size):
    """The solver's own time grid (solvers.py:69-87): the output times themselves without a step size, else
    t0 + arange(ceil((t_end - t0) / h + 1)) * h with the last point moved onto t_end."""
    if step_size is None:
        return t
    n = int(torch.ceil((t[-1] - t[0]) / step_size + 1).item())
    g = torch.arange(0, n, dtype=t.dtype, device=t.device) * step_size + t[0]
    return torch.cat([g[:-1], t[-1:]])


def solve_fixed(f, y0, t, method, step_size):
    """y(t[i]) for every output time; f maps (t, tuple state) -> tuple; t increasing or decreasing (solved in negated time with the
    field's sign flipped, misc.py:262-271).  Returns a tuple of tensors [len(t), ...].  Differentiable torch ops throughout."""
    if t.numel() > 1 and bool(t[0] > t[1]):
        t = -t
        inner = f
        f = lambda s, y: tuple(-k for k in inner(-s, y))      # noqa: E731
    grid = _grid(t, step_size)
    assert bool(grid[0] == t[0]) and bool(grid[-1] == t[-1])      # solvers.py:96
    out = [[y] for y in y0]
    j = 1
    y = tuple(y0)
    for i in range(grid.numel() - 1):
        g0, g1 = grid[i], grid[i + 1]
        dy = _step(f, method, g0, g1 - g0, g1, y)
        y1 = tuple(a + b for a, b in zip(y, dy))
        while j < t.numel() and bool(g1 >= t[j]):      # outputs between grid states: linear interpolation (solvers.py:108-116, 166-172)
            tj = t[j]
            if bool(tj == g1):
                pick = y1
            elif bool(tj == g0):
                pick = y
            else:
                slope = (tj - g0) / (g1 - g0)
                pick = tuple(a + slope * (b - a) for a, b in zip(y, y1))
            for o, v in zip(out, pick):
                o.append(v)
            j += 1
        y = y1
    return tuple(torch.stack(o, dim=0) for o in out)


# ---- adaptive Dormand-Prince 5(4) on a TUPLE state --------------------------------------------------------------------------

_DP_ALPHA = (1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0)
_DP_BETA = ((1 / 5,), (3 / 40, 9 / 40), (44 / 45, -56 / 15, 32 / 9), (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
            (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656), (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84))
_DP_CERR = (35 / 384 - 1951 / 21600, 0.0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720, -2187 / 6784 - -12231 / 42400,
            11 / 84 - 649 / 6300, -1.0 / 60.0)
_DP_CMID = (6025192743 / 30085553152 / 2, 0.0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
            187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2)


def _rms_norm(parts):
    """rms over EVERY element of the (flattened) state: one norm for the whole batch (misc.py:18-19).  0-dim tensor (differentiable:
    autograd goes through the norms of the initial-step rule when the solve is taped)."""
    return (sum(p.pow(2).sum() for p in parts) / max(1, sum(p.numel() for p in parts))).sqrt()


def _mixed_norm(parts):
    """The adjoint's norm over (vjp_t, y, a, g_theta...): max(|vjp_t|, rms(y), rms(a), max_p rms(g_p))  (adjoint.py:239-242)."""
    rms = lambda q: q.pow(2).mean().sqrt() if q.numel() else q.new_zeros(())      # noqa: E731
    vals = [parts[0].abs().max(), rms(parts[1]), rms(parts[2])]
    if len(parts) > 3:
        vals.append(torch.stack([rms(q) for q in parts[3:]]).max())
    return torch.stack(vals).max()


def _before(t):
    """The largest representable time below t, gradient of the identity (misc.py:318-324)."""
    return t + (torch.nextafter(t, torch.full_like(t, -math.inf)) - t).detach()


def solve_dopri5(f, y0, t, rtol, atol, options, norm, stats=None):
    """y at every output time, adaptive steps, 4th-order dense output (steps are NOT clipped to output times).  f maps
    (t in the state dtype, tuple state) -> tuple.  Step control in fp64 and without gradients (misc.py:84-97 is @no_grad); the
    initial step of _select_initial_step IS differentiable, as in the reference.  Returns a tuple of [len(t), ...] tensors."""
    if t.numel() > 1 and bool(t[0] > t[1]):
        t = -t
        inner = f
        f = lambda s, y: tuple(-k for k in inner(-s, y))      # noqa: E731
    opt = dict(options or {})
    f64 = torch.float64
    dtype = y0[0].dtype
    dev = y0[0].device
    t = t.to(f64)
    min_step, max_step = float(opt.get("min_step", 0.0)), float(opt.get("max_step", math.inf))
    safety, ifactor, dfactor = float(opt.get("safety", 0.9)), float(opt.get("ifactor", 10.0)), float(opt.get("dfactor", 0.2))
    max_num_steps = int(opt.get("max_num_steps", 2 ** 31 - 1))
    nfe = [0]

    def ev(tt, y, before=False):
        tt = tt.to(dtype)
        nfe[0] += 1
        return f(_before(tt) if before else tt, y)

    y = tuple(y0)
    f0 = ev(t[0], y)
    if opt.get("first_step") is None:      # misc.py:33-74 (differentiable, unlike every later step size)
        scale = tuple(atol + q.abs() * rtol for q in y)
        d0 = norm(tuple(q / sc for q, sc in zip(y, scale)))
        d1 = norm(tuple(k / sc for k, sc in zip(f0, scale)))
        h0 = d0.new_tensor(1e-6) if (float(d0) < 1e-5 or float(d1) < 1e-5) else 0.01 * d0 / d1
        f1 = ev(t[0].to(dtype) + h0, tuple(q + h0 * k for q, k in zip(y, f0)))
        d2 = norm(tuple((b - a) / sc for a, b, sc in zip(f0, f1, scale))) / h0
        if float(d1) <= 1e-15 and float(d2) <= 1e-15:
            h1 = torch.max(h0.new_tensor(1e-6), h0 * 1e-3)
        else:
            h1 = (0.01 / torch.max(d1, d2)) ** (1.0 / 5.0)
        dt = torch.min(100 * h0, h1).to(f64)
    else:
        dt = torch.as_tensor(opt["first_step"], dtype=f64, device=dev)
    t0 = t[0]
    t_prev, interp = t[0], None
    out = [[q] for q in y]
    n_acc = n_rej = 0
    for i in range(1, t.numel()):
        n = 0
        while bool(t[i] > t0):
            if n >= max_num_steps:
                raise AssertionError("max_num_steps exceeded (%d)" % max_num_steps)
            n += 1
            t1 = t0 + dt
            if not bool(t1 > t0):
                raise AssertionError("underflow in dt %g" % float(dt))
            t0s, dts, t1s = t0.to(dtype), dt.to(dtype), t1.to(dtype)
            ks = [f0]
            yi = y
            for a_i, b_i in zip(_DP_ALPHA, _DP_BETA):      # rk_common.py:61-73
                yi = tuple(q + sum(k[j] * (b * dts) for k, b in zip(ks, b_i)) for j, q in enumerate(y))
                ks.append(ev(t1s, yi, before=True) if a_i == 1.0 else ev(t0s + a_i * dts, yi))
            y1, f1 = yi, ks[-1]      # c_sol = (beta[-1], 0): the input of the last stage IS the solution
            err = tuple(sum(k[j] * (c * dts) for k, c in zip(ks, _DP_CERR)) for j in range(len(y)))
            with torch.no_grad():
                ratio = float(norm(tuple(e / (atol + rtol * torch.max(a.abs(), b.abs())) for e, a, b in zip(err, y, y1))))
            if not math.isfinite(ratio):
                raise AssertionError("non-finite values in state `y`")
            accept = ratio <= 1.0
            if float(dt) > max_step:
                accept = False
            if float(dt) <= min_step:
                accept = True
            if accept:
                n_acc += 1
                ym = tuple(q + sum(k[j] * (c * dts) for k, c in zip(ks, _DP_CMID)) for j, q in enumerate(y))
                interp = []
                for j in range(len(y)):      # interp.py:4-61
                    fa, fb = ks[0][j], f1[j]
                    a = 2 * dts * (fb - fa) - 8 * (y1[j] + y[j]) + 16 * ym[j]
                    b = dts * (5 * fa - 3 * fb) + 18 * y[j] + 14 * y1[j] - 32 * ym[j]
                    c = dts * (fb - 4 * fa) - 11 * y[j] - 5 * y1[j] + 16 * ym[j]
                    interp.append((y[j], dts * fa, c, b, a))
                t_prev, t0, y, f0 = t0, t1, y1, f1
            else:
                n_rej += 1
            with torch.no_grad():      # misc.py:84-97
                if ratio == 0.0:
                    factor = ifactor
                else:
                    factor = min(ifactor, max(safety / ratio ** 0.2, 1.0 if ratio < 1.0 else dfactor))
                dt = (dt.detach() * factor).clamp(min_step, max_step)
        x = ((t[i] - t_prev) / (t0 - t_prev)).to(dtype)
        for o, cf in zip(out, interp):
            total = cf[0] + x * cf[1]
            xp = x
            for c in cf[2:]:
                xp = xp * x
                total = total + xp * c
            o.append(total)
    if stats is not None:
        stats.update(nfe=stats.get("nfe", 0) + nfe[0], accepted=stats.get("accepted", 0) + n_acc, rejected=stats.get("rejected", 0) + n_rej)
    return tuple(torch.stack(o, dim=0) for o in out)


class _AdjointDopri5(torch.autograd.Function):
    """Continuous adjoint with the adaptive solver (adjoint.py:37-145): forward under no_grad; backward = one adaptive reverse solve
    per output interval of (vjp_t, y, a, g_theta...) with the mixed norm, a and y reset / incremented at every output time."""

    @staticmethod
    def forward(ctx, cfg, z0, t, *params):
        with torch.no_grad():
            (y,) = solve_dopri5(lambda s, st: (cfg["field"](s, st[0]),), (z0,), t, cfg["rtol"], cfg["atol"], cfg["options"], _rms_norm, cfg["stats"])
        ctx.cfg = cfg
        ctx.save_for_backward(t, y, *params)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        cfg = ctx.cfg
        t, y, *params = ctx.saved_tensors
        field = cfg["field"]
        params = tuple(params)
        with torch.no_grad():
            def aug(s, state):
                yy, aa = state[1], state[2]
                with torch.enable_grad():
                    s_ = s.detach().requires_grad_(True)      # (the reference's time tensor requires grad here: vjp_t is part of the norm)
                    y_ = yy.detach().requires_grad_(True)
                    fe = field(s_, y_)
                    vj = torch.autograd.grad(fe, (s_, y_) + params, -aa, allow_unused=True, retain_graph=False)
                vt = torch.zeros_like(s) if vj[0] is None else vj[0]
                vy = torch.zeros_like(yy) if vj[1] is None else vj[1]
                vp = tuple(torch.zeros_like(p) if v is None else v for p, v in zip(params, vj[2:]))
                return (vt, fe.detach(), vy) + vp

            t_needs = ctx.needs_input_grad[2]
            state = [torch.zeros((), dtype=y.dtype, device=y.device), y[-1], grad_y[-1]] + [torch.zeros_like(p) for p in params]
            time_vjps = torch.empty(t.numel(), dtype=t.dtype, device=t.device) if t_needs else None
            for i in range(t.numel() - 1, 0, -1):
                if t_needs:      # dL/dt_i through the end point of the interval: f(t_i, y_i) . dL/dy_i  (adjoint.py:112-123, every solver)
                    dcur = (field(t[i].to(y.dtype), y[i]).reshape(-1) * grad_y[i].reshape(-1)).sum()
                    state[0] = state[0] - dcur
                    time_vjps[i] = dcur
                sol = solve_dopri5(aug, tuple(state), t[i - 1:i + 1].flip(0), cfg["adjoint_rtol"], cfg["adjoint_atol"], cfg["adjoint_options"],
                                   _mixed_norm, cfg["stats_backward"])
                state = [s_[1] for s_ in sol]
                state[1] = y[i - 1]
                state[2] = state[2] + grad_y[i - 1]
            if t_needs:
                time_vjps[0] = state[0]      # adjoint.py:136
        gp = [g if need else None for g, need in zip(state[3:], ctx.needs_input_grad[3:])]
        return (None, state[2] if ctx.needs_input_grad[1] else None, time_vjps, *gp)


class _Adjoint(torch.autograd.Function):
    """Continuous adjoint (adjoint.py:37-145): forward under no_grad, backward = one reverse solve per output interval of the
    augmented state (vjp_t, y, a, g_theta...), a and y reset / incremented at every output time."""

    @staticmethod
    def forward(ctx, cfg, z0, t, *params):
        with torch.no_grad():
            (y,) = solve_fixed(lambda s, st: (cfg["field"](s, st[0]),), (z0,), t, cfg["method"], cfg["step"])
        ctx.cfg = cfg
        ctx.save_for_backward(t, y, *params)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        cfg = ctx.cfg
        t, y, *params = ctx.saved_tensors
        field, method, step = cfg["field"], cfg["method"], cfg["step"]
        t_needs = ctx.needs_input_grad[2]
        params = tuple(params)
        with torch.no_grad():
            def aug(s, state):
                yy, aa = state[1], state[2]
                with torch.enable_grad():
                    s_ = s.detach().requires_grad_(True)
                    y_ = yy.detach().requires_grad_(True)
                    fe = field(s_ if t_needs else s_.detach(), y_)
                    vj = torch.autograd.grad(fe, (s_, y_) + params, -aa, allow_unused=True, retain_graph=False)
                vt = torch.zeros_like(s) if vj[0] is None else vj[0]
                vy = torch.zeros_like(yy) if vj[1] is None else vj[1]
                vp = tuple(torch.zeros_like(p) if v is None else v for p, v in zip(params, vj[2:]))
                return (vt, fe.detach(), vy) + vp

            state = [torch.zeros((), dtype=y.dtype, device=y.device), y[-1], grad_y[-1]] + [torch.zeros_like(p) for p in params]
            time_vjps = torch.empty(t.numel(), dtype=t.dtype, device=t.device) if t_needs else None
            for i in range(t.numel() - 1, 0, -1):
                if t_needs:      # dL/dt_i through the end point of the interval: f(t_i, y_i) . dL/dy_i
                    dcur = (field(t[i], y[i]).reshape(-1) * grad_y[i].reshape(-1)).sum()
                    state[0] = state[0] - dcur
                    time_vjps[i] = dcur
                sol = solve_fixed(aug, tuple(state), t[i - 1:i + 1].flip(0), method, step)
                state = [s_[1] for s_ in sol]
                state[1] = y[i - 1]
                state[2] = state[2] + grad_y[i - 1]
            if t_needs:
                time_vjps[0] = state[0]
        gp = [g if need else None for g, need in zip(state[3:], ctx.needs_input_grad[3:])]
        return (None, state[2] if ctx.needs_input_grad[1] else None, time_vjps, *gp)


def cdeint_unfused(X, func, z0, t, adjoint, mode, method, step_size, adjoint_params=None, adaptive=None):
    """-> [..., len(t), H].  `method` in {euler, midpoint, rk4}: step_size None = step from output time to output time;
    `method` = 'dopri5': `adaptive` = {rtol, atol, options, adjoint_rtol, adjoint_atol, adjoint_options}."""
    field = ControlledField(X, func, mode)
    t = torch.as_tensor(t, device=z0.device)
    if not t.is_floating_point():
        t = t.to(z0.dtype)
    params = []
    if adjoint:
        if adjoint_params is None:
            adjoint_params = tuple(func.parameters()) if isinstance(func, torch.nn.Module) else ()
        seen = set()
        for p in adjoint_params:      # de-duplicated, requires_grad only (adjoint.py:176-183)
            if torch.is_tensor(p) and p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                params.append(p)
    if method == "dopri5":
        stats, stats_b = {}, {}
        if adjoint:
            cfg = dict(adaptive, field=field, stats=stats, stats_backward=stats_b)
            y = _AdjointDopri5.apply(cfg, z0, t, *params)
        else:      # adjoint=False: autograd tapes the torch ops of the solve itself
            (y,) = solve_dopri5(lambda s, st: (field(s, st[0]),), (z0,), t, adaptive["rtol"], adaptive["atol"], adaptive["options"], _rms_norm, stats)
    elif adjoint:
        cfg = {"field": field, "method": method, "step": step_size}
        y = _Adjoint.apply(cfg, z0, t, *params)
    else:
        (y,) = solve_fixed(lambda s, st: (field(s, st[0]),), (z0,), t, method, step_size)
    dims = list(range(1, y.dim() - 1))
    return y.permute(*dims, 0, y.dim() - 1)      # time to dim -2 (solver.py:227-229)
