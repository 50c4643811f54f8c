"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package never does (it fails loudly when the HIP library is missing).

This is a plain PyTorch-CPU fp32 restatement of the op sequence the reference executes for
``torchcde.cdeint(X, func, z0, t, adjoint=True, method in {rk4, midpoint, euler}, options={'step_size': 1})``
on the default integer knot grid.  No autograd is used: the adjoint sweep uses hand-written VJPs.

PARITY PIN: this file is checked against the *imported reference itself* (vendored torchcde 0.2.0 /
torchdiffeq 0.2.1 under /root/reference/modules) by ``oracle/gen_golden.py``, which also writes the
golden vectors in tests/golden/*.npz that tests/test_oracle_golden.py re-checks on every run.

Reference lines restated (relative to /root/reference):
  * knot index / linear derivative    modules/torchcde/torchcde/interpolation_linear.py:195-198, 212-219, 231-234
  * cubic derivative                   modules/torchcde/torchcde/interpolation_cubic.py:291-305, 315-322, 331-336
  * f_theta (MLP, shared inner layer)  src/ncde/vector_fields/base.py:64-69, 83-92, 97-104
  * contraction f(z) dX/dt             modules/torchcde/torchcde/solver.py:112-137
  * fixed grid loop / output pick      modules/torchdiffeq/torchdiffeq/_impl/solvers.py:78-87, 94-119, 166-172
  * Euler / midpoint / RK4 (3/8 rule)  modules/torchdiffeq/torchdiffeq/_impl/fixed_grid.py:6-29, rk_common.py:106-114
  * time reversal                      modules/torchdiffeq/torchdiffeq/_impl/misc.py:152-159, 262-271
  * adjoint sweep                      modules/torchdiffeq/torchdiffeq/_impl/adjoint.py:37-145
"""
import math

import torch

_ONE_THIRD = 1 / 3
_TWO_THIRDS = 2 / 3


def _f32(x):
    return torch.tensor(float(x), dtype=torch.float32)


class Field:
    """Vector field f_theta.

    ``layers`` is a list of (W, b) applied as Linear+ReLU (the "inner net", H or H+C -> HH -> ... -> HH); the same
    (W, b) tensors may appear several times (the reference repeats ONE inner layer nl-1 times, base.py:66-68).
    ``Wo, bo`` is the tanh head.  Variants (src/ncde/vector_fields/gating.py, solver.py:112-137):
      kind  'original' : M = tanh(Wo hh + bo)
            'minimal'  : M = sigmoid(Wg hh + bg) * tanh(Wo hh + bo)                       (gating.py:7-32)
            'gru'      : M = sigmoid(Wg net(u) + bg) * tanh(Wo net(sigmoid(Wr u + br) * u) + bo)   (gating.py:35-61)
      mode  'matmul'    : u = z, M viewed [H, C], output M . dX/dt        (row index of the heads = h*C + c)
            'evaluate'  : u = [z, X(t)],     output M  (heads have H rows, no contraction)   (solver.py:123-126)
            'derivative': u = [z, dX/dt(t)], output M
    """

    def __init__(self, layers, Wo, bo, hidden, channels, kind="original", mode="matmul", Wg=None, bg=None, Wr=None, br=None):
        self.layers = [(torch.as_tensor(w), torch.as_tensor(b)) for w, b in layers]
        self.Wo = torch.as_tensor(Wo)
        self.bo = torch.as_tensor(bo)
        self.H = hidden
        self.C = channels
        self.kind, self.mode = kind, mode
        self.Wg = None if Wg is None else torch.as_tensor(Wg)
        self.bg = None if bg is None else torch.as_tensor(bg)
        self.Wr = None if Wr is None else torch.as_tensor(Wr)
        self.br = None if br is None else torch.as_tensor(br)
        assert kind in ("original", "minimal", "gru") and mode in ("matmul", "evaluate", "derivative")
        assert self.Wo.shape[0] == (hidden * channels if mode == "matmul" else hidden)

    @staticmethod
    def original(p, hidden, channels, num_layers):
        """OriginalVectorField parameter dict {W0,b0,W1,b1,Wo,bo} -> Field (W1 shared nl-1 times)."""
        return Field.variant(p, hidden, channels, num_layers)

    @staticmethod
    def variant(p, hidden, channels, num_layers, kind="original", mode="matmul"):
        """Parameter dict {W0,b0,[W1,b1],Wo,bo,[Wg,bg],[Wr,br]} -> Field."""
        t = {k: torch.as_tensor(v) for k, v in p.items()}
        layers = [(t["W0"], t["b0"])]
        layers += [(t["W1"], t["b1"])] * (num_layers - 1)
        return Field(layers, t["Wo"], t["bo"], hidden, channels, kind, mode, t.get("Wg"), t.get("bg"), t.get("Wr"), t.get("br"))

    def unique_params(self):
        """De-duplicated parameter tensors in nn.Module.parameters() order of the reference modules: inner net, then
        (reset net), (sigmoid head), tanh head (adjoint.py:176-183; gating.py registers reset, sigmoid, tanh in that order)."""
        seen, out = set(), []
        for w, b in self.layers:
            for p in (w, b):
                if id(p) not in seen:
                    seen.add(id(p))
                    out.append(p)
        if self.kind == "gru":
            out += [self.Wr, self.br]
        if self.kind in ("minimal", "gru"):
            out += [self.Wg, self.bg]
        out += [self.Wo, self.bo]
        return out

    def _net(self, u):
        acts = [u]
        x = u
        for w, b in self.layers:
            x = torch.relu(torch.addmm(b, x, w.t()))
            acts.append(x)
        return acts

    # -- forward: dz/dt for state z and control input cin (dX/dt for matmul/derivative, X(t) for evaluate) --------------
    def g(self, z, cin, save=False):
        u = z if self.mode == "matmul" else torch.cat([z, cin], dim=-1)
        sv = {"u": u}
        if self.kind == "gru":
            rg = torch.sigmoid(torch.addmm(self.br, u, self.Wr.t()))
            sv["rg"] = rg
            acts_i = self._net(u)
            acts_r = self._net(rg * u)
            sv["acts_i"], sv["acts_r"] = acts_i, acts_r
            hh_s, hh_t = acts_i[-1], acts_r[-1]
        else:
            acts = self._net(u)
            sv["acts_i"] = sv["acts_r"] = acts
            hh_s = hh_t = acts[-1]
        th = torch.tanh(torch.addmm(self.bo, hh_t, self.Wo.t()))
        sv["th"] = th
        if self.kind == "original":
            m = th
        else:
            sg = torch.sigmoid(torch.addmm(self.bg, hh_s, self.Wg.t()))
            sv["sg"] = sg
            m = sg * th
        if self.mode == "matmul":
            out = (m.view(-1, self.H, self.C) @ cin.unsqueeze(-1)).squeeze(-1)
        else:
            out = m
        if save:
            return out, sv
        return out

    # -- VJP of g wrt (z, params) for cotangent c [B, H] ----------------------------------------------------------------
    def g_vjp(self, sv, cin, c):
        grads = {}

        def acc(p, v):
            if id(p) in grads:
                grads[id(p)] = grads[id(p)] + v
            else:
                grads[id(p)] = v

        def net_bwd(acts, dxl):
            for li in range(len(self.layers) - 1, -1, -1):
                w, b = self.layers[li]
                dpre = dxl * (acts[li + 1] > 0).to(dxl.dtype)
                acc(b, dpre.sum(0))
                acc(w, dpre.t() @ acts[li])
                dxl = dpre @ w
            return dxl

        if self.mode == "matmul":
            dm = (c.unsqueeze(-1) * cin.unsqueeze(-2)).reshape(-1, self.H * self.C)
        else:
            dm = c
        th = sv["th"]
        if self.kind == "original":
            dpt = dm * (1 - th * th)
        else:
            sg = sv["sg"]
            dpt = (dm * sg) * (1 - th * th)
            dps = (dm * th) * (sg * (1 - sg))
        acc(self.bo, dpt.sum(0))
        acc(self.Wo, dpt.t() @ sv["acts_r"][-1])
        if self.kind == "original":
            du = net_bwd(sv["acts_r"], dpt @ self.Wo)
        elif self.kind == "minimal":
            acc(self.bg, dps.sum(0))
            acc(self.Wg, dps.t() @ sv["acts_i"][-1])
            du = net_bwd(sv["acts_i"], dpt @ self.Wo + dps @ self.Wg)
        else:
            acc(self.bg, dps.sum(0))
            acc(self.Wg, dps.t() @ sv["acts_i"][-1])
            d_ru = net_bwd(sv["acts_r"], dpt @ self.Wo)          # cotangent of rg * u
            du = net_bwd(sv["acts_i"], dps @ self.Wg)
            rg, u = sv["rg"], sv["u"]
            du = du + d_ru * rg
            dpr = (d_ru * u) * (rg * (1 - rg))
            acc(self.br, dpr.sum(0))
            acc(self.Wr, dpr.t() @ u)
            du = du + dpr @ self.Wr
        dz = du[:, :self.H]
        return dz, [grads[id(p)] for p in self.unique_params()]


class Control:
    """Control path. kind in {'linear', 'cubic'} ('rectilinear' data uses 'linear' evaluation, src/ncde/ncde.py:12-15).
    ``t`` = the knot grid the coefficients were built on (None = the default integer grid linspace(0, T-1, T),
    interpolation_linear.py:195-196 / interpolation_cubic.py:291-292)."""

    def __init__(self, coeffs, kind, t=None):
        self.kind = kind
        coeffs = torch.as_tensor(coeffs)
        self.coeffs = coeffs
        if kind == "linear":
            self.n_pieces = coeffs.shape[-2] - 1
            self.channels = coeffs.shape[-1]
        elif kind == "cubic":
            self.n_pieces = coeffs.shape[-2]
            ch = coeffs.shape[-1] // 4
            self.channels = ch
            self.a, self.b = coeffs[..., :ch], coeffs[..., ch:2 * ch]
            self.two_c, self.three_d = coeffs[..., 2 * ch:3 * ch], coeffs[..., 3 * ch:]
        else:
            raise ValueError(kind)
        self.n_knots = self.n_pieces + 1
        self.default_grid = t is None
        if t is None:
            t = torch.linspace(0, self.n_knots - 1, self.n_knots, dtype=coeffs.dtype)
        self.t = torch.as_tensor(t, dtype=coeffs.dtype)
        assert self.t.shape == (self.n_knots,)
        if kind == "linear":
            # (c[1:]-c[:-1]) / (t[1:]-t[:-1])  (interpolation_linear.py:198)
            self.derivs = (coeffs[..., 1:, :] - coeffs[..., :-1, :]) / (self.t[1:] - self.t[:-1]).unsqueeze(-1)

    def x0(self):
        return self.coeffs[..., 0, :] if self.kind == "linear" else self.a[..., 0, :]

    def piece(self, t):
        """bucketize(t, knots, right=False) - 1, clamped: the LEFT piece at an exact knot
        (interpolation_linear.py:212-219, interpolation_cubic.py:315-322)."""
        if self.default_grid:
            idx = int(math.ceil(float(t))) - 1
        else:
            idx = int(torch.bucketize(torch.as_tensor(t, dtype=self.t.dtype), self.t)) - 1
        return max(0, min(idx, self.n_pieces - 1))

    def evaluate(self, t):
        """X(t) (interpolation_linear.py:221-229, interpolation_cubic.py:324-329)."""
        idx = self.piece(t)
        frac = t - self.t[idx]
        if self.kind == "linear":
            prev, nxt = self.coeffs[..., idx, :], self.coeffs[..., idx + 1, :]
            diff_t = self.t[idx + 1] - self.t[idx]
            return prev + frac * (nxt - prev) / diff_t
        inner = 0.5 * self.two_c[..., idx, :] + self.three_d[..., idx, :] * frac / 3
        inner = self.b[..., idx, :] + inner * frac
        return self.a[..., idx, :] + inner * frac

    def field_input(self, t, mode):
        """What the vector field receives besides z: dX/dt(t) ('matmul', 'derivative') or X(t) ('evaluate')."""
        return self.evaluate(t) if mode == "evaluate" else self.derivative(t)

    def derivative(self, t):
        idx = self.piece(t)
        if self.kind == "linear":
            return self.derivs[..., idx, :]
        frac = t - self.t[idx]
        inner = self.two_c[..., idx, :] + self.three_d[..., idx, :] * frac
        return self.b[..., idx, :] + inner * frac

    def second_derivative(self, t):
        """d/dt of derivative(t): zero for a piecewise-linear path (the knot index carries no gradient), two_c + 2 three_d
        frac for the cubic spline -- what autograd sends to t in adjoint.py:95-98 (the vjp_t component of the state)."""
        idx = self.piece(t)
        if self.kind == "linear":
            return torch.zeros_like(self.derivs[..., idx, :])
        frac = t - self.t[idx]
        inner = self.two_c[..., idx, :] + self.three_d[..., idx, :] * frac
        return inner + self.three_d[..., idx, :] * frac


def stage_plan(method):
    if method not in ("rk4", "midpoint", "euler"):
        raise ValueError('Invalid method "{}"'.format(method))
    return method


def _step(fn, method, t0, dt, t1, y0):
    """One fixed step on a tuple state.  fn(t, state) -> tuple of derivatives."""
    if method == "euler":
        f0 = fn(t0, y0)
        return tuple(y + dt * f for y, f in zip(y0, f0))
    if method == "midpoint":
        half_dt = 0.5 * dt
        f0 = fn(t0, y0)
        ymid = tuple(y + f * half_dt for y, f in zip(y0, f0))
        fm = fn(t0 + half_dt, ymid)
        return tuple(y + dt * f for y, f in zip(y0, fm))
    # rk4 = torchdiffeq's rk4_alt_step_func (3/8 rule)
    k1 = fn(t0, y0)
    k2 = fn(t0 + dt * _ONE_THIRD, tuple(y + dt * a * _ONE_THIRD for y, a in zip(y0, k1)))
    k3 = fn(t0 + dt * _TWO_THIRDS, tuple(y + dt * (b - a * _ONE_THIRD) for y, a, b in zip(y0, k1, k2)))
    k4 = fn(t1, tuple(y + dt * (a - b + c) for y, a, b, c in zip(y0, k1, k2, k3)))
    return tuple(y + (a + 3 * (b + c) + d) * dt * 0.125 for y, a, b, c, d in zip(y0, k1, k2, k3, k4))


def solve_forward(control, field, z0, method="rk4", sequence=False, nfe=None):
    """z at t = [0, T-1] (sequence=False -> [B,2,H]) or at every knot (sequence=True -> [B,T,H])."""
    stage_plan(method)
    z0 = torch.as_tensor(z0)
    T = control.n_knots

    def fn(t, state):
        if nfe is not None:
            nfe[0] += 1
        return (field.g(state[0], control.field_input(t, field.mode)),)

    ys = [z0]
    y = (z0,)
    for n in range(T - 1):
        t0, t1 = _f32(n), _f32(n + 1)
        y = _step(fn, method, t0, t1 - t0, t1, y)
        if sequence:
            ys.append(y[0])
    if not sequence:
        ys.append(y[0])
    return torch.stack(ys, dim=1)


def solve_adjoint(control, field, z_out, grad_out, method="rk4", sequence=False, nfe=None):
    """Continuous-adjoint reverse sweep of adjoint.py:37-145.

    z_out / grad_out: [B, n_out, H] with n_out = 2 (interval) or T (every knot).
    Returns (dL/dz0 [B,H], [dL/dparam ...] in Field.unique_params() order).
    """
    stage_plan(method)
    z_out = torch.as_tensor(z_out)
    grad_out = torch.as_tensor(grad_out)
    T = control.n_knots
    params = field.unique_params()

    def fn(s, state):
        # _ReverseFunc: -base(-s, .)   (misc.py:152-159); base = augmented_dynamics with cotangent -a
        if nfe is not None:
            nfe[0] += 1
        y, a = state[0], state[1]
        t = -s
        dx = control.field_input(t, field.mode)
        f, saved = field.g(y, dx, save=True)
        vjp_y, vjp_p = field.g_vjp(saved, dx, -a)
        return (-f, -vjp_y) + tuple(-v for v in vjp_p)

    y = z_out[:, -1]
    a = grad_out[:, -1].clone()
    g = tuple(torch.zeros_like(p) for p in params)

    def sweep(n_hi, n_lo, y, a, g):
        state = (y, a) + g
        for n in range(n_hi, n_lo, -1):          # reverse step n -> n-1, in negated time s: -n -> -(n-1)
            s0, s1 = _f32(-n), _f32(-(n - 1))
            state = _step(fn, method, s0, s1 - s0, s1, state)
        return state[0], state[1], tuple(state[2:])

    if sequence:
        for i in range(T - 1, 0, -1):
            y, a, g = sweep(i, i - 1, y, a, g)
            y = z_out[:, i - 1]                  # reset to the stored forward value (adjoint.py:132)
            a = a + grad_out[:, i - 1]           # (adjoint.py:133)
    else:
        y, a, g = sweep(T - 1, 0, y, a, g)
        a = a + grad_out[:, 0]
    return a, list(g)


def _forward_stages(control, field, z0, method):
    """Forward solve keeping, per step, the stage times, the stage inputs and dt."""
    stage_plan(method)
    z0 = torch.as_tensor(z0)
    T = control.n_knots
    steps = []
    y = z0
    for n in range(T - 1):
        t0, t1 = _f32(n), _f32(n + 1)
        dt = t1 - t0
        if method == "euler":
            ts, Ys = [t0], [y]
            y = y + dt * field.g(y, control.field_input(t0, field.mode))
        elif method == "midpoint":
            half = 0.5 * dt
            k1 = field.g(y, control.field_input(t0, field.mode))
            ym = y + k1 * half
            ts, Ys = [t0, t0 + half], [y, ym]
            y = y + dt * field.g(ym, control.field_input(t0 + half, field.mode))
        else:
            k1 = field.g(y, control.field_input(t0, field.mode))
            y2 = y + dt * k1 * _ONE_THIRD
            k2 = field.g(y2, control.field_input(t0 + dt * _ONE_THIRD, field.mode))
            y3 = y + dt * (k2 - k1 * _ONE_THIRD)
            k3 = field.g(y3, control.field_input(t0 + dt * _TWO_THIRDS, field.mode))
            y4 = y + dt * (k1 - k2 + k3)
            k4 = field.g(y4, control.field_input(t1, field.mode))
            ts, Ys = [t0, t0 + dt * _ONE_THIRD, t0 + dt * _TWO_THIRDS, t1], [y, y2, y3, y4]
            y = y + (k1 + 3 * (k2 + k3) + k4) * dt * 0.125
        steps.append((ts, Ys, dt))

    return steps


def stage_record(control, field, z0, method="rk4"):
    """Stage inputs of the forward solve as the C-ABI's stage record: [(T-1)*S, B, H] (step-major, stage-minor)."""
    return torch.stack([Y for _, Ys, _ in _forward_stages(control, field, z0, method) for Y in Ys], dim=0)


def solve_discrete_backward(control, field, z0, grad_out, method="rk4", sequence=False):
    """Exact gradient of the DISCRETISED solve (what ``cdeint(..., adjoint=False)`` + autograd computes:
    modules/torchcde/torchcde/solver.py:224 picks ``odeint``; backprop runs through
    solvers.py:94-119 and fixed_grid.py:6-29 / rk_common.py:106-114).  Hand-written reverse sweep:
    the stage inputs of every step are recomputed from the forward solve, then each step is
    transposed stage by stage (no autograd).

    grad_out: [B, n_out, H].  Returns (dL/dz0 [B,H], [dL/dparam ...] in Field.unique_params() order).
    """
    grad_out = torch.as_tensor(grad_out)
    T = control.n_knots
    params = field.unique_params()
    steps = _forward_stages(control, field, z0, method)
    g = [torch.zeros_like(p) for p in params]

    def pull(t, Y, ck):
        """cotangent ck of k = g(t, Y)  ->  cotangent of Y; parameter gradients accumulated."""
        dx = control.field_input(t, field.mode)
        _, saved = field.g(Y, dx, save=True)
        dY, dp = field.g_vjp(saved, dx, ck)
        for i, v in enumerate(dp):
            g[i] = g[i] + v
        return dY

    a = grad_out[:, -1].clone()
    for n in range(T - 2, -1, -1):
        ts, Ys, dt = steps[n]
        if method == "euler":
            a = a + pull(ts[0], Ys[0], dt * a)
        elif method == "midpoint":
            dYm = pull(ts[1], Ys[1], dt * a)
            dY1 = pull(ts[0], Ys[0], (0.5 * dt) * dYm)
            a = a + dYm + dY1
        else:
            ck4 = a * dt * 0.125
            dY4 = pull(ts[3], Ys[3], ck4)
            ck3 = 3 * ck4 + dt * dY4
            dY3 = pull(ts[2], Ys[2], ck3)
            ck2 = 3 * ck4 - dt * dY4 + dt * dY3
            dY2 = pull(ts[1], Ys[1], ck2)
            ck1 = ck4 + dt * dY4 - (dt * _ONE_THIRD) * dY3 + (dt * _ONE_THIRD) * dY2
            dY1 = pull(ts[0], Ys[0], ck1)
            a = a + dY4 + dY3 + dY2 + dY1
        if sequence and n > 0:
            a = a + grad_out[:, n]
    a = a + grad_out[:, 0]           # row 0 of the solution is z0 itself
    return a, g


# ======================================================================================================================
# General time axis: any increasing output times t, any step_size, user knot grids (the rest of the cdeint call surface).
# Restates  solvers.py:78-87 (grid from step_size), :94-119 (loop + output pick), :166-172 (linear interpolation of the
# output between the two bracketing grid states)  and  adjoint.py:116-133 (one reverse solve per output interval, each
# with its OWN grid starting at t[i]).  All time arithmetic in the dtype of ``t`` (fp32 here), exactly as torch does it.
# ======================================================================================================================
def fixed_time_grid(t, step_size):
    """solvers.py:78-87: arange(0, ceil((t[-1]-t[0])/step + 1)) * step + t[0], last entry := t[-1]."""
    t = torch.as_tensor(t)
    start, end = t[0], t[-1]
    niters = torch.ceil((end - start) / step_size + 1).item()
    grid = torch.arange(0, niters, dtype=t.dtype) * step_size + start
    grid[-1] = t[-1]
    return grid


def _linear_interp(t0, t1, y0, y1, t):
    if t == t0:
        return y0
    if t == t1:
        return y1
    slope = (t - t0) / (t1 - t0)
    return y0 + slope * (y1 - y0)


def solve_forward_times(control, field, z0, t, method="rk4", step_size=1.0, nfe=None):
    """z at the times t (increasing, t[0] = start of the solve) -> [B, len(t), H]."""
    stage_plan(method)
    z0 = torch.as_tensor(z0)
    t = torch.as_tensor(t)
    if not t.is_floating_point():
        t = t.to(z0.dtype)
    grid = fixed_time_grid(t, step_size)      # in the dtype of t (fp32 or fp64), as torch does
    assert grid[0] == t[0] and grid[-1] == t[-1]

    def fn(tt, state):
        if nfe is not None:
            nfe[0] += 1
        return (field.g(state[0], control.field_input(tt.to(z0.dtype), field.mode)),)     # misc.py:181: t.to(y.dtype)

    sol = [z0]
    j = 1
    y0 = z0
    for t0, t1 in zip(grid[:-1], grid[1:]):
        y1 = _step(fn, method, t0, t1 - t0, t1, (y0,))[0]
        while j < len(t) and t1 >= t[j]:
            sol.append(_linear_interp(t0, t1, y0, y1, t[j]))
            j += 1
        y0 = y1
    return torch.stack(sol, dim=1)


def solve_adjoint_times(control, field, t, z_out, grad_out, method="rk4", step_size=1.0, nfe=None):
    """Continuous adjoint for arbitrary output times: for i = len(t)-1 .. 1 one reverse solve from t[i] to t[i-1] on its
    own grid in negated time (adjoint.py:116-133, misc.py:262-271), y reset to the stored z_out[i-1], a += grad_out[i-1]."""
    stage_plan(method)
    z_out = torch.as_tensor(z_out)
    grad_out = torch.as_tensor(grad_out)
    t = torch.as_tensor(t)
    if not t.is_floating_point():
        t = t.to(z_out.dtype)
    params = field.unique_params()

    def fn(s, state):
        if nfe is not None:
            nfe[0] += 1
        y, a = state[0], state[1]
        tt = -(s.to(y.dtype))
        dx = control.field_input(tt, field.mode)
        f, saved = field.g(y, dx, save=True)
        vjp_y, vjp_p = field.g_vjp(saved, dx, -a)
        return (-f, -vjp_y) + tuple(-v for v in vjp_p)

    y = z_out[:, -1]
    a = grad_out[:, -1].clone()
    g = tuple(torch.zeros_like(p) for p in params)
    for i in range(len(t) - 1, 0, -1):
        grid = fixed_time_grid(-t[i - 1:i + 1].flip(0), step_size)
        state = (y, a) + g
        for s0, s1 in zip(grid[:-1], grid[1:]):
            state = _step(fn, method, s0, s1 - s0, s1, state)
        a, g = state[1], tuple(state[2:])
        y = z_out[:, i - 1]
        a = a + grad_out[:, i - 1]
    return a, list(g)


def _forward_stages_times(control, field, z0, t, method, step_size):
    z0 = torch.as_tensor(z0)
    t = torch.as_tensor(t)
    if not t.is_floating_point():
        t = t.to(z0.dtype)
    grid = fixed_time_grid(t, step_size)
    steps = []
    y = z0
    _inp = control.field_input

    class _C:      # the control path sees the stage time cast to the state dtype (misc.py:181)
        @staticmethod
        def field_input(tt, mode):
            return _inp(tt.to(z0.dtype), mode)
    control = _C
    for t0, t1 in zip(grid[:-1], grid[1:]):
        dt = t1 - t0
        if method == "euler":
            ts, Ys = [t0], [y]
            y1 = y + dt * field.g(y, control.field_input(t0, field.mode))
        elif method == "midpoint":
            half = 0.5 * dt
            k1 = field.g(y, control.field_input(t0, field.mode))
            ym = y + k1 * half
            ts, Ys = [t0, t0 + half], [y, ym]
            y1 = y + dt * field.g(ym, control.field_input(t0 + half, field.mode))
        else:
            k1 = field.g(y, control.field_input(t0, field.mode))
            y2 = y + dt * k1 * _ONE_THIRD
            k2 = field.g(y2, control.field_input(t0 + dt * _ONE_THIRD, field.mode))
            y3 = y + dt * (k2 - k1 * _ONE_THIRD)
            k3 = field.g(y3, control.field_input(t0 + dt * _TWO_THIRDS, field.mode))
            y4 = y + dt * (k1 - k2 + k3)
            k4 = field.g(y4, control.field_input(t1, field.mode))
            ts, Ys = [t0, t0 + dt * _ONE_THIRD, t0 + dt * _TWO_THIRDS, t1], [y, y2, y3, y4]
            y1 = y + (k1 + 3 * (k2 + k3) + k4) * dt * 0.125
        steps.append((ts, Ys, dt, t0, t1))
        y = y1
    return grid, steps


def stage_record_times(control, field, z0, t, method="rk4", step_size=1.0):
    _, steps = _forward_stages_times(control, field, z0, t, method, step_size)
    return torch.stack([Y for _, Ys, _, _, _ in steps for Y in Ys], dim=0)


def solve_discrete_backward_times(control, field, z0, t, grad_out, method="rk4", step_size=1.0):
    """Exact gradient of the discretised solve for arbitrary output times (autograd through solvers.py:94-119 incl. the
    linear interpolation of the outputs, :166-172).  grad_out: [B, len(t), H]."""
    grad_out = torch.as_tensor(grad_out)
    t = torch.as_tensor(t)
    if not t.is_floating_point():
        t = t.to(grad_out.dtype)
    params = field.unique_params()
    grid, steps = _forward_stages_times(control, field, z0, t, method, step_size)
    # which outputs each step emits (solvers.py:108-116)
    emits = [[] for _ in steps]
    j = 1
    for n, (_, _, _, t0, t1) in enumerate(steps):
        while j < len(t) and t1 >= t[j]:
            emits[n].append(j)
            j += 1
    g = [torch.zeros_like(p) for p in params]

    def pull(tt, Y, ck):
        dx = control.field_input(tt.to(Y.dtype), field.mode)
        _, saved = field.g(Y, dx, save=True)
        dY, dp = field.g_vjp(saved, dx, ck.to(Y.dtype))
        for i, v in enumerate(dp):
            g[i] = g[i] + v
        return dY

    a = torch.zeros_like(grad_out[:, 0])      # cotangent of the state after the step being transposed
    for n in range(len(steps) - 1, -1, -1):
        ts, Ys, dt, t0, t1 = steps[n]
        a_y0 = torch.zeros_like(a)
        for jj in emits[n]:
            gj = grad_out[:, jj]
            if t[jj] == t0:
                a_y0 = a_y0 + gj
            elif t[jj] == t1:
                a = a + gj
            else:
                slope = (t[jj] - t0) / (t1 - t0)
                a = a + slope * gj
                a_y0 = a_y0 + (gj - slope * gj)
        if method == "euler":
            a = a + pull(ts[0], Ys[0], dt * a)
        elif method == "midpoint":
            dYm = pull(ts[1], Ys[1], dt * a)
            dY1 = pull(ts[0], Ys[0], (0.5 * dt) * dYm)
            a = a + dYm + dY1
        else:
            ck4 = a * dt * 0.125
            dY4 = pull(ts[3], Ys[3], ck4)
            ck3 = 3 * ck4 + dt * dY4
            dY3 = pull(ts[2], Ys[2], ck3)
            ck2 = 3 * ck4 - dt * dY4 + dt * dY3
            dY2 = pull(ts[1], Ys[1], ck2)
            ck1 = ck4 + dt * dY4 - (dt * _ONE_THIRD) * dY3 + (dt * _ONE_THIRD) * dY2
            dY1 = pull(ts[0], Ys[0], ck1)
            a = a + dY4 + dY3 + dY2 + dY1
        a = a + a_y0
    a = a + grad_out[:, 0]
    return a, g


def time_plan_words(control, t, method, step_size):
    """The time plan of csrc/ncde_timeplan.hip (layout: csrc/ncde_common.h) restated with torch's own arithmetic: the
    checker for the C++ builder (bit for bit).  t: 1-D tensor (fp32 or fp64: the grid arithmetic runs in its dtype)."""
    import numpy as np
    t = torch.as_tensor(t)
    S = {"rk4": 4, "midpoint": 2, "euler": 1}[method]
    pw = 3 + 3 * S
    f32 = torch.float32

    def stage_words(tt):
        tt = tt.to(f32)
        idx = control.piece(tt)
        frac = tt - control.t[idx]
        kdt = control.t[idx + 1] - control.t[idx]
        return [np.int32(idx), np.float32(frac).view(np.int32), np.float32(kdt).view(np.int32)]

    def step_words(t0, t1, neg):
        dt = t1 - t0
        if method == "rk4":
            ts = [t0, t0 + dt * _ONE_THIRD, t0 + dt * _TWO_THIRDS, t1]
        elif method == "midpoint":
            ts = [t0, t0 + 0.5 * dt]
        else:
            ts = [t0]
        out = []
        for x in ts:
            out += stage_words(-x if neg else x)
        return np.float32(dt.to(f32)).view(np.int32), out

    grid = fixed_time_grid(t, step_size)
    n_fwd, nt = len(grid) - 1, len(t)
    fwd, outs = [], [np.int32(0), np.int32(0)]
    j = 1
    for t0, t1 in zip(grid[:-1], grid[1:]):
        dtw, sw = step_words(t0, t1, False)
        first, cnt = j, 0
        while j < nt and t1 >= t[j]:
            if t[j] == t0:
                outs += [np.int32(0), np.int32(0)]
            elif t[j] == t1:
                outs += [np.int32(1), np.int32(0)]
            else:
                slope = (t[j] - t0) / (t1 - t0)
                outs += [np.int32(2), np.float32(slope.to(f32)).view(np.int32)]
            j += 1
            cnt += 1
        fwd += [dtw, np.int32(first), np.int32(cnt)] + sw
    adj = []
    n_adj = 0
    for i in range(nt - 1, 0, -1):
        g = fixed_time_grid(-t[i - 1:i + 1].flip(0), step_size)
        for q, (s0, s1) in enumerate(zip(g[:-1], g[1:])):
            dtw, sw = step_words(s0, s1, True)
            adj += [dtw, np.int32(i - 1 if q == len(g) - 2 else -1), np.int32(0)] + sw
            n_adj += 1
    off_fwd = 8
    off_out = off_fwd + n_fwd * pw
    off_adj = off_out + 2 * nt
    head = [np.int32(0x4e43504c), S, n_fwd, n_adj, nt, off_fwd, off_out, off_adj]
    return np.array([np.int32(x) for x in head + fwd + outs + adj], dtype=np.int32), n_fwd, n_adj


# ======================================================================================================================
# Adaptive Dormand-Prince 5(4) ("dopri5"): NeuralCDE(solver="dopri5") (src/ncde/ncde.py:129-134, options {"min_step": 0.5})
# and the toy's default method (experiments/sim_bm_toy_example.py:54-57).  Restates, on FLAT state vectors as the reference
# does (misc.py:131-139, 204-214):
#   tableau, mid-point weights          torchdiffeq/_impl/dopri5.py:5-36
#   one RK step + error estimate        rk_common.py:41-86        (stage times in the STATE dtype, state matmul over stages)
#   step control                        rk_common.py:216-305      (time in fp64; accept iff ratio <= 1, min/max step overrides)
#   initial step, error ratio, next dt  misc.py:33-103
#   dense output                        interp.py:4-61, rk_common.py:307-313, 198-205
#   stage time perturbation             misc.py:168-191 (alpha = 1 stages are evaluated just BEFORE t1)
#   ONE error norm for the whole batch  misc.py:18-19 (rms over every element of the state)
#   adjoint: one adaptive reverse solve per output interval over (vjp_t, y, a, g_theta) with the mixed norm
#                                       adjoint.py:37-145, 235-247
# ======================================================================================================================
_DP_ALPHA = torch.tensor([1 / 5, 3 / 10, 4 / 5, 8 / 9, 1., 1.], dtype=torch.float64)
_DP_BETA = [
    torch.tensor([1 / 5], dtype=torch.float64),
    torch.tensor([3 / 40, 9 / 40], dtype=torch.float64),
    torch.tensor([44 / 45, -56 / 15, 32 / 9], dtype=torch.float64),
    torch.tensor([19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729], dtype=torch.float64),
    torch.tensor([9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656], dtype=torch.float64),
    torch.tensor([35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84], dtype=torch.float64),
]
_DP_C_ERROR = torch.tensor([35 / 384 - 1951 / 21600, 0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
                            -2187 / 6784 - -12231 / 42400, 11 / 84 - 649 / 6300, -1. / 60.], dtype=torch.float64)
_DP_C_MID = torch.tensor([6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
                          187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2], dtype=torch.float64)


def _rms(x):
    return x.pow(2).mean().sqrt()


class Dopri5:
    """RKAdaptiveStepsizeODESolver specialised to the Dormand-Prince tableau, on a flat state.
    ``func(t, y, perturb)``: t a 0-dim tensor ALREADY in the state dtype; perturb in {0: none, -1: just before t}."""

    def __init__(self, func, y0, rtol, atol, norm=_rms, min_step=0.0, max_step=float("inf"), first_step=None, safety=0.9,
                 ifactor=10.0, dfactor=0.2, max_num_steps=2 ** 31 - 1):
        f64 = torch.float64
        self.func, self.y0, self.norm = func, y0, norm
        self.rtol, self.atol = torch.as_tensor(rtol, dtype=f64), torch.as_tensor(atol, dtype=f64)
        self.min_step, self.max_step = torch.as_tensor(min_step, dtype=f64), torch.as_tensor(max_step, dtype=f64)
        self.first_step = None if first_step is None else torch.as_tensor(first_step, dtype=f64)
        self.safety, self.ifactor, self.dfactor = (torch.as_tensor(v, dtype=f64) for v in (safety, ifactor, dfactor))
        self.max_num_steps = max_num_steps
        dt_ = y0.dtype
        self.alpha = _DP_ALPHA.to(dt_)
        self.beta = [b.to(dt_) for b in _DP_BETA]
        self.c_error, self.mid = _DP_C_ERROR.to(dt_), _DP_C_MID.to(dt_)
        self.n_accept = self.n_reject = 0
        self.trace = []        # (t0, dt, accepted, error_ratio) per attempt
        self.tape = None       # set to [] to record every ACCEPTED step (dopri5_discrete_backward)
        self.init_info = None  # what _select_initial_step computed (for the gradient of the first step size)
        self.out_steps = []    # per output time i >= 1: (index of the accepted step it was interpolated in, x)

    def _f(self, t, y, perturb=0):
        t = t.to(y.dtype)
        if perturb < 0:
            t = torch.nextafter(t, torch.tensor(-math.inf))
        return self.func(t, y)

    def _select_initial_step(self, t0, f0):
        y0 = self.y0
        dtype = y0.dtype
        t0 = t0.to(dtype)
        scale = self.atol + torch.abs(y0) * self.rtol
        d0, d1 = self.norm(y0 / scale), self.norm(f0 / scale)
        if d0 < 1e-5 or d1 < 1e-5:
            h0 = torch.tensor(1e-6, dtype=dtype)
        else:
            h0 = 0.01 * d0 / d1
        y1 = y0 + h0 * f0
        f1 = self._f(t0 + h0, y1)
        d2 = self.norm((f1 - f0) / scale) / h0
        if d1 <= 1e-15 and d2 <= 1e-15:
            h1 = torch.max(torch.tensor(1e-6, dtype=dtype), h0 * 1e-3)
        else:
            h1 = (0.01 / max(d1, d2)) ** (1. / float(4 + 1))
        self.init_info = {"t0": t0, "f0": f0, "scale": scale, "d0": d0, "d1": d1, "d2": d2, "h0": h0, "h1": h1, "y1": y1, "f1": f1,
                          "h0_const": bool(d0 < 1e-5 or d1 < 1e-5), "h1_const": bool(d1 <= 1e-15 and d2 <= 1e-15)}
        return torch.min(100 * h0, h1).to(torch.float64)

    def _rk_step(self, y0, f0, t0, dt, t1):
        t0, dt, t1 = t0.to(y0.dtype), dt.to(y0.dtype), t1.to(y0.dtype)
        k = torch.empty(*f0.shape, 7, dtype=y0.dtype)
        k[..., 0] = f0
        self._stage_times = []
        for i, (alpha_i, beta_i) in enumerate(zip(self.alpha, self.beta)):
            if alpha_i == 1.:
                ti, perturb = t1, -1
            else:
                ti, perturb = t0 + alpha_i * dt, 0
            yi = y0 + k[..., :i + 1].matmul(beta_i * dt).view_as(f0)
            k[..., i + 1] = self._f(ti, yi, perturb)
            self._stage_times.append(torch.nextafter(ti, torch.tensor(-math.inf)) if perturb < 0 else ti)
        y1 = yi                                 # c_sol = (beta[-1], 0): the last stage input IS the solution
        return y1, k[..., -1], k.matmul(dt * self.c_error), k

    def _optimal_step_size(self, last_step, error_ratio):
        if error_ratio == 0:
            return last_step * self.ifactor
        dfactor = self.dfactor
        if error_ratio < 1:
            dfactor = torch.ones((), dtype=last_step.dtype)
        error_ratio = error_ratio.type_as(last_step)
        exponent = torch.tensor(5, dtype=last_step.dtype).reciprocal()
        factor = torch.min(self.ifactor, torch.max(self.safety / error_ratio ** exponent, dfactor))
        return last_step * factor

    def _adaptive_step(self, st):
        y0, f0, _, t0, dt, interp = st
        t1 = t0 + dt
        assert t0 + dt > t0, "underflow in dt {}".format(dt.item())
        assert torch.isfinite(y0).all(), "non-finite values in state `y`"
        y1, f1, y1_error, k = self._rk_step(y0, f0, t0, dt, t1)
        error_tol = self.atol + self.rtol * torch.max(y0.abs(), y1.abs())
        error_ratio = self.norm(y1_error / error_tol)
        accept = bool(error_ratio <= 1)
        if dt > self.max_step:
            accept = False
        if dt <= self.min_step:
            accept = True
        self.trace.append((float(t0), float(dt), accept, float(error_ratio)))
        if accept:
            self.n_accept += 1
            dtf = dt.type_as(y0)
            y_mid = y0 + k.matmul(dtf * self.mid).view_as(y0)
            fa, fb = k[..., 0], k[..., -1]
            a = 2 * dtf * (fb - fa) - 8 * (y1 + y0) + 16 * y_mid
            b = dtf * (5 * fa - 3 * fb) + 18 * y0 + 14 * y1 - 32 * y_mid
            c = dtf * (fb - 4 * fa) - 11 * y0 - 5 * y1 + 16 * y_mid
            interp = [y0, dtf * fa, c, b, a]
            t_next, y_next, f_next = t1, y1, f1
            if self.tape is not None:
                self.tape.append({"t0": t0, "dt": dt, "y0": y0, "k": k, "ts": list(self._stage_times), "attempt": len(self.trace) - 1})
        else:
            self.n_reject += 1
            t_next, y_next, f_next = t0, y0, f0
        dt_next = self._optimal_step_size(dt, error_ratio).clamp(self.min_step, self.max_step)
        return (y_next, f_next, t0, t_next, dt_next, interp)

    def integrate(self, t):
        t = torch.as_tensor(t).to(torch.float64)
        sol = [self.y0]
        f0 = self._f(t[0], self.y0)
        first = self._select_initial_step(t[0], f0) if self.first_step is None else self.first_step
        st = (self.y0, f0, t[0], t[0], first, [self.y0] * 5)
        for i in range(1, len(t)):
            n = 0
            while t[i] > st[3]:
                assert n < self.max_num_steps, "max_num_steps exceeded"
                st = self._adaptive_step(st)
                n += 1
            coeffs, t0, t1 = st[5], st[2], st[3]
            assert (t0 <= t[i]) & (t[i] <= t1)
            x = ((t[i] - t0) / (t1 - t0)).to(coeffs[0].dtype)
            self.out_steps.append((self.n_accept - 1, x))
            total = coeffs[0] + x * coeffs[1]
            xp = x
            for cf in coeffs[2:]:
                xp = xp * x
                total = total + xp * cf
            sol.append(total)
        return torch.stack(sol, dim=0)


def dopri5_forward(control, field, z0, t, rtol, atol, options=None, stats=None):
    """cdeint(..., method='dopri5') forward: z at the times t -> [B, len(t), H]; stats gets nfe / accepted / rejected."""
    z0 = torch.as_tensor(z0)
    nfe = [0]

    def func(tt, y):
        nfe[0] += 1
        return field.g(y, control.field_input(tt, field.mode))

    sv = Dopri5(func, z0, rtol, atol, **(options or {}))
    sol = sv.integrate(t)
    if stats is not None:
        stats.update(nfe=nfe[0], accepted=sv.n_accept, rejected=sv.n_reject, trace=sv.trace)
    return sol.permute(1, 0, 2).contiguous()


def dopri5_adjoint(control, field, t, z_out, grad_out, rtol, atol, options=None, stats=None, vjp="hand"):
    """Continuous adjoint with the adaptive solver (adjoint.py:37-145): per output interval a fresh dopri5 solve of the
    flat augmented state [vjp_t, y, a, g_theta...] in negated time, mixed norm max(|t|, rms(y), rms(a), max_p rms(g_p)).

    vjp = "hand": the hand-written VJPs used everywhere else in this oracle.  vjp = "autograd": the stage VJP through
    torch.autograd.grad as adjoint.py:95-98 does -- same values up to summation order, but an ADAPTIVE solve amplifies
    last-bit differences into different step sequences (steps cluster at the kinks of a piecewise-linear control, where a
    1e-7 change of dt moves a stage across a knot), so gen_golden.py pins the step-control logic with "autograd" (step
    sequence identical to the reference, gradients bit-level) and the hand VJPs at solver-tolerance level."""
    z_out, grad_out = torch.as_tensor(z_out), torch.as_tensor(grad_out)
    t = torch.as_tensor(t).to(torch.float64)
    params = field.unique_params()
    B, H = z_out.shape[0], z_out.shape[2]
    sizes = [1, B * H, B * H] + [p.numel() for p in params]
    nfe = [0]
    acc = {"accepted": 0, "rejected": 0, "trace": []}

    def unpack(v):
        out, o = [], 0
        for n in sizes:
            out.append(v[o:o + n])
            o += n
        return out

    def norm(v):
        parts = unpack(v)
        return max([parts[0].abs().max(), _rms(parts[1]), _rms(parts[2])] + [max([_rms(q) for q in parts[3:]])])

    def func(s, v):                      # _ReverseFunc(mul = -1) around augmented_dynamics evaluated at -s
        nfe[0] += 1
        parts = unpack(v)
        y, a = parts[1].view(B, H), parts[2].view(B, H)
        tt = -s
        dx = control.field_input(tt, field.mode)
        if vjp == "autograd":
            with torch.enable_grad():
                yg = y.detach().requires_grad_(True)
                tg = tt.detach().requires_grad_(True)
                for q in params:
                    q.requires_grad_(True)
                f = field.g(yg, control.field_input(tg, field.mode))
                vjp_t, vjp_y, *vjp_p = torch.autograd.grad(f, (tg, yg) + tuple(params), -a, allow_unused=True)
                for q in params:
                    q.requires_grad_(False)
            f = f.detach()
            vjp_p = [torch.zeros_like(q) if gq is None else gq for q, gq in zip(params, vjp_p)]
            vjp_t = torch.zeros(1, dtype=v.dtype) if vjp_t is None else vjp_t.reshape(1)
        else:
            f, saved = field.g(y, dx, save=True)
            vjp_y, vjp_p = field.g_vjp(saved, dx, -a)
            # vjp_t: adjoint.py:75-98 calls func with a time tensor that (through the in-place requires_grad_ on the detached
            # alias) DOES require grad, so the cubic spline's dependence on t lands in the first state component and, via
            # |vjp_t|, in the error norm.  matmul mode: (-a)^T M d2X/dt2 summed over the batch.
            if control.kind == "cubic":
                if field.mode != "matmul":
                    raise NotImplementedError("hand vjp_t only for the matmul input")
                m = saved["th"] if field.kind == "original" else saved["sg"] * saved["th"]
                ddx = ((-a).unsqueeze(-1) * m.view(-1, field.H, field.C)).sum(1)
                vjp_t = (ddx * control.second_derivative(tt)).sum().reshape(1)
            else:
                vjp_t = torch.zeros(1, dtype=v.dtype)
        flat = torch.cat([vjp_t, f.reshape(-1), vjp_y.reshape(-1)] + [q.reshape(-1) for q in vjp_p])
        return -1.0 * flat

    y = z_out[:, -1]
    a = grad_out[:, -1].clone()
    g = [torch.zeros_like(p) for p in params]
    vt = torch.zeros(1, dtype=y.dtype)       # vjp_t: carried across the output intervals like a and g (adjoint.py:131)
    for i in range(len(t) - 1, 0, -1):
        v0 = torch.cat([vt, y.reshape(-1), a.reshape(-1)] + [q.reshape(-1) for q in g])
        sv = Dopri5(func, v0, rtol, atol, norm=norm, **(options or {}))
        v1 = sv.integrate(-t[i - 1:i + 1].flip(0))[1]
        acc["accepted"] += sv.n_accept
        acc["rejected"] += sv.n_reject
        acc["trace"] += sv.trace
        parts = unpack(v1)
        vt = parts[0]
        a = parts[2].view(B, H)
        g = [q.view_as(p) for q, p in zip(parts[3:], params)]
        y = z_out[:, i - 1]
        a = a + grad_out[:, i - 1]
    if stats is not None:
        stats.update(nfe=nfe[0], **acc)
    return a, g


def dopri5_discrete_backward(control, field, z0, t, grad_out, rtol, atol, options=None, stats=None):
    """cdeint(..., method='dopri5', adjoint=False): reverse-mode through the taped adaptive solve, by hand.  What the reference's
    autograd differentiates (rk_common.py:216-305 under torchdiffeq.odeint):
      * every ACCEPTED step -- the six stage evaluations (rk_common.py:41-86), with FSAL: k1 of a step IS k7 of the previous one;
      * the 4th-order dense output that produces the solution at the requested times (interp.py:4-61, rk_common.py:307-313);
      * NOT the step-size controller: _optimal_step_size is @torch.no_grad() (misc.py:84-97), so every dt but the first is a constant;
      * the FIRST step size: _select_initial_step (misc.py:33-74) is differentiable in (y0, theta), and it reaches the solution
        through step 1 itself (y_i = y0 + dt sum beta k, the fit's dt terms), through every later step's start time
        t0_m = t[0] + dt_1 + (constants) -- the dense-output abscissa x = (t - t0)/(t1 - t0) of every output, and, for a control whose
        derivative depends on t (cubic), every stage time -- provided the first attempt was accepted (otherwise dt_2 is a constant).
    Returns (dL/dz0, [dL/dtheta]) for L = sum(z_out * grad_out); stats gets the forward's nfe / step counts."""
    z0, grad_out = torch.as_tensor(z0), torch.as_tensor(grad_out)
    t = torch.as_tensor(t).to(torch.float64)
    params = field.unique_params()
    nfe = [0]

    def func(tt, y):
        nfe[0] += 1
        return field.g(y, control.field_input(tt, field.mode))

    sv = Dopri5(func, z0, rtol, atol, **(options or {}))
    sv.tape = []
    sol = sv.integrate(t).permute(1, 0, 2).contiguous()
    tape, M = sv.tape, len(sv.tape)
    outs = [[] for _ in range(M)]
    for j, (m, x) in enumerate(sv.out_steps):
        outs[m].append((j + 1, x))
    dtype = z0.dtype
    beta, alpha, mid = sv.beta, sv.alpha, sv.mid
    gth = [torch.zeros_like(p) for p in params]

    def vjp(tt, y, c):
        """cotangent c of k = f(tt, y) -> (ybar, tbar); theta-bar accumulates"""
        cin = control.field_input(tt, field.mode)
        _, saved = field.g(y, cin, save=True)
        dy, dp = field.g_vjp(saved, cin, c)
        for q, v in zip(gth, dp):
            q += v
        tb = 0.0
        if control.kind == "cubic":
            if field.mode != "matmul":
                raise NotImplementedError("time gradient of the field input only for the matmul mode")
            mth = saved["th"] if field.kind == "original" else saved["sg"] * saved["th"]
            ddx = (c.unsqueeze(-1) * mth.view(-1, field.H, field.C)).sum(1)
            tb = float((ddx * control.second_derivative(tt)).sum())
        return dy, tb

    # is dt_1 a differentiable function of (y0, theta)?  (no first_step option, and the very first attempt was accepted)
    delta_active = sv.first_step is None and M > 0 and tape[0]["attempt"] == 0
    Yb1 = torch.zeros_like(z0)           # cotangent of the step's y1 (= the next step's y0)
    Kb_next = torch.zeros_like(z0)       # cotangent of the step's k7 coming from its use as the next step's k1 (FSAL)
    Tbar, dtbar1 = 0.0, 0.0              # d/d(t0 of the steps m >= 2) summed; d/d(dt_1) of step 1's own arithmetic
    for m in range(M - 1, -1, -1):
        st = tape[m]
        y0, k, dt64 = st["y0"], st["k"], st["dt"]
        dtf = dt64.to(dtype)
        Kb = [torch.zeros_like(y0) for _ in range(7)]
        Kb[6] = Kb[6] + Kb_next
        yb0 = torch.zeros_like(y0)
        yb1 = Yb1.clone()
        dtb, tb0 = 0.0, 0.0
        k1, k7 = k[..., 0], k[..., 6]
        if outs[m]:
            y1 = y0 + k[..., :6].matmul(beta[5] * dtf)
            y_mid = y0 + k.matmul(dtf * mid)
            ca = 2 * dtf * (k7 - k1) - 8 * (y1 + y0) + 16 * y_mid
            cb = dtf * (5 * k1 - 3 * k7) + 18 * y0 + 14 * y1 - 32 * y_mid
            cc = dtf * (k7 - 4 * k1) - 11 * y0 - 5 * y1 + 16 * y_mid
            cd = dtf * k1
            ab, bb, cbb, db, eb = (torch.zeros_like(y0) for _ in range(5))
            for j, x in outs[m]:
                g = grad_out[:, j]
                x2 = x * x
                x3 = x2 * x
                x4 = x3 * x
                eb += g
                db += x * g
                cbb += x2 * g
                bb += x3 * g
                ab += x4 * g
                xbar = float((g * (cd + 2 * x * cc + 3 * x2 * cb + 4 * x3 * ca)).double().sum())
                if m >= 1:
                    Tbar += xbar * (-1.0 / float(dt64))            # x = (t - t0)/dt_m, t0 moves with dt_1
                else:
                    dtbar1 += xbar * (-float(x) / float(dt64))     # x = (t - t[0])/dt_1
            ymb = 16 * ab - 32 * bb + 16 * cbb
            yb0 += -8 * ab + 18 * bb - 11 * cbb + eb + ymb
            yb1 += -8 * ab + 14 * bb - 5 * cbb
            Kb[0] += dtf * (-2 * ab + 5 * bb - 4 * cbb + db)
            Kb[6] += dtf * (2 * ab - 3 * bb + cbb)
            dtb += float((ab * 2 * (k7 - k1) + bb * (5 * k1 - 3 * k7) + cbb * (k7 - 4 * k1) + db * k1).double().sum())
            for j in range(7):
                Kb[j] += (dtf * mid[j]) * ymb
            dtb += float((ymb * k.matmul(mid)).double().sum())
        for i in range(6, 0, -1):          # stage i: k[..., i] = f(ts[i-1], y0 + k[..., :i] @ (beta[i-1] dt))
            yi = y0 + k[..., :i].matmul(beta[i - 1] * dtf)
            dy, tbi = vjp(st["ts"][i - 1], yi, Kb[i])
            ybi = dy + yb1 if i == 6 else dy
            yb0 += ybi
            for jj in range(i):
                Kb[jj] += (beta[i - 1][jj] * dtf) * ybi
            dtb += float((ybi * k[..., :i].matmul(beta[i - 1])).double().sum()) + float(alpha[i - 1]) * tbi
            tb0 += tbi
        if m >= 1:
            Tbar += tb0
        else:
            dtbar1 += dtb
        Yb1, Kb_next = yb0, Kb[0]
    dz0 = Yb1 + grad_out[:, 0]             # the solution at t[0] is z0 itself
    f0b = Kb_next                          # cotangent of f0 = f(t[0], y0)
    t0f = t[0].to(dtype)
    if delta_active:
        ii = sv.init_info
        dbar = dtbar1 + Tbar               # dL/d(dt_1)
        h0, h1, d0, d1, d2, scale, f0, f1, y1p = (ii[k_] for k_ in ("h0", "h1", "d0", "d1", "d2", "scale", "f0", "f1", "y1"))
        n_el = z0.numel()
        h0b = h1b = 0.0
        if float(100 * h0) <= float(h1):
            h0b = 100.0 * dbar
        else:
            h1b = dbar
        d1b = d2b = 0.0
        if h1b != 0.0 and not ii["h1_const"]:
            mx = max(float(d1), float(d2))
            mxb = -0.2 * float(h1) / mx * h1b
            if float(d1) >= float(d2):
                d1b += mxb
            else:
                d2b += mxb
        elif h1b != 0.0:
            h0b += 1e-3 * h1b if float(h0 * 1e-3) > 1e-6 else 0.0
        scb = torch.zeros_like(z0)
        if d2b != 0.0:
            n2 = float(d2) * float(h0)                      # d2 = rms((f1 - f0)/scale) / h0
            h0b += -float(d2) / float(h0) * d2b
            q = (f1 - f0) / scale
            qb = (d2b / float(h0)) * q / (n_el * n2)
            f1b = qb / scale
            f0b = f0b - qb / scale
            scb = scb - qb * q / scale
            dy1, tbp = vjp(ii["t0"] + h0, y1p, f1b)
            dz0 = dz0 + dy1
            f0b = f0b + float(h0) * dy1
            h0b += float((dy1 * f0).double().sum()) + tbp
        if h0b != 0.0 and not ii["h0_const"]:
            d0b = 0.01 / float(d1) * h0b
            d1b += -float(h0) / float(d1) * h0b
            q0 = z0 / scale
            q0b = d0b * q0 / (n_el * float(d0))
            dz0 = dz0 + q0b / scale
            scb = scb - q0b * q0 / scale
        if d1b != 0.0:
            q1 = f0 / scale
            q1b = d1b * q1 / (n_el * float(d1))
            f0b = f0b + q1b / scale
            scb = scb - q1b * q1 / scale
        dz0 = dz0 + scb * float(sv.rtol) * torch.sign(z0)
    dy0, _ = vjp(t0f, z0, f0b)
    dz0 = dz0 + dy0
    if stats is not None:
        stats.update(nfe=nfe[0], accepted=sv.n_accept, rejected=sv.n_reject, trace=sv.trace, delta_active=bool(delta_active),
                     first_step=float(tape[0]["dt"]) if M else None)
    return sol, dz0, gth
