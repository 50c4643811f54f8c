"""The UNFUSED solver: ``cdeint`` for requests the fused HIP kernels do not cover -- an arbitrary ``func`` (any
``nn.Module (t, z) -> [..., H, C]``, /root/reference/modules/torchcde/torchcde/solver.py:102-137), decreasing output times
(torchdiffeq/_impl/misc.py:262-271), gradients with respect to the control path or the output times, non-fp32 tensors.

It is an own restatement in torch ops of the same algorithms the kernels implement -- the fixed-grid solvers of
torchdiffeq (solvers.py:78-119, 166-172; fixed_grid.py:6-29; the 3/8 rule of rk_common.py:106-114) and the continuous adjoint
(adjoint.py:37-145: one reverse solve of (vjp_t, y, a, g_theta) per output interval, stage VJPs by ``torch.autograd.grad``) --
running on the device the tensors live on (the GPU: ``cdeint`` refuses CPU tensors, there is no CPU fallback), one torch kernel
per elementary operation.  It is the reference's speed class, not the fused kernels'; ``cdeint`` warns once per reason when it
takes this path.
"""
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


def cdeint_unfused(X, func, z0, t, adjoint, mode, method, step_size, adjoint_params=None):
    """-> [..., len(t), H].  `method` in {euler, midpoint, rk4}; step_size None = step from output time to output time."""
    field = ControlledField(X, func, mode)
    t = torch.as_tensor(t, device=z0.device)
    if not t.is_floating_point():
        t = t.to(z0.dtype)
    if adjoint:
        if adjoint_params is None:
            adjoint_params = tuple(func.parameters()) if isinstance(func, torch.nn.Module) else ()
        seen, params = set(), []
        for p in adjoint_params:      # de-duplicated, requires_grad only (adjoint.py:176-183)
            if torch.is_tensor(p) and p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                params.append(p)
        cfg = {"field": field, "method": method, "step": step_size}
        y = _Adjoint.apply(cfg, z0, t, *params)
    else:
        (y,) = solve_fixed(lambda s, st: (field(s, st[0]),), (z0,), t, method, step_size)
    dims = list(range(1, y.dim() - 1))
    return y.permute(*dims, 0, y.dim() - 1)      # time to dim -2 (solver.py:227-229)
