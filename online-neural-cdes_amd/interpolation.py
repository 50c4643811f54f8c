"""Control paths: host-side mirrors of ``torchcde.LinearInterpolation`` / ``torchcde.NaturalCubicSpline``
(/root/reference/modules/torchcde/torchcde/interpolation_linear.py:183-234, interpolation_cubic.py:268-336).

They keep the reference's constructor, ``grid_points``, ``interval``, ``evaluate`` and ``derivative``
(the last two in plain torch ops, used outside the solve, e.g. for ``h0 = Linear(X(0))``).  Inside
``cdeint`` the fused HIP kernels read the raw coefficient tensor and evaluate dX/dt on chip, so
the ``[B, T-1, C]`` derivative tensor the reference materialises in its constructor is never built.
"""
import weakref

import torch


class _Tagged(torch.Tensor):
    """A time tensor that remembers which grid of which control it is (avoids a device sync in cdeint)."""


def _tag(t, kind, owner):
    t = t.as_subclass(_Tagged)
    t._ncde_kind = kind
    t._ncde_owner = weakref.ref(owner)
    return t


_GRID_CACHE = {}   # (n_knots, dtype, device) -> (t, [t0, t_last]): the default integer grid is built once, not per forward


def _default_grid(n_knots, like):
    key = (n_knots, like.dtype, like.device)
    hit = _GRID_CACHE.get(key)
    if hit is None:
        t = torch.linspace(0, n_knots - 1, n_knots, dtype=like.dtype, device=like.device)
        hit = (t, torch.stack([t[0], t[-1]]))
        if len(_GRID_CACHE) > 64:
            _GRID_CACHE.clear()
        _GRID_CACHE[key] = hit
    return hit


class _ControlBase(torch.nn.Module):
    interp_name = None

    def _setup_t(self, t, n_knots, like):
        self._default_grid = t is None
        self._interval = None
        if t is None:
            t, self._interval = _default_grid(n_knots, like)
        self.register_buffer("_t", t)

    @property
    def grid_points(self):
        return _tag(self._t, "knots", self)

    @property
    def interval(self):
        iv = self._interval if self._interval is not None else torch.stack([self._t[0], self._t[-1]])
        return _tag(iv, "interval", self)

    def _at_first_knot(self, t):
        """X(t) at t = first knot of the default grid is the first coefficient row itself (fractional part 0): no
        bucketize / gather kernels for the ``h0 = Linear(X(0))`` of every forward (ncde.py:170-198)."""
        return self._default_grid and isinstance(t, (int, float)) and t == 0

    def _interpret_t(self, t, n_pieces):
        t = torch.as_tensor(t, dtype=self._t.dtype, device=self._t.device)
        index = torch.bucketize(t.detach(), self._t.detach()).sub(1).clamp(0, n_pieces - 1)
        return t - self._t[index], index

    def forward(self, t):  # convenience, not part of the reference interface
        return self.evaluate(t)


class LinearInterpolation(_ControlBase):
    """Piecewise linear path through ``coeffs[..., T, C]`` (output of linear_interpolation_coeffs;
    rectilinear data is simply a longer such tensor)."""

    interp_name = "linear"

    def __init__(self, coeffs, t=None, **kwargs):
        super().__init__(**kwargs)
        self._setup_t(t, coeffs.size(-2), coeffs)
        self.register_buffer("_coeffs", coeffs)

    @property
    def fused_coeffs(self):
        return self._coeffs

    @property
    def n_knots(self):
        return self._coeffs.size(-2)

    @property
    def channels(self):
        return self._coeffs.size(-1)

    def evaluate(self, t):
        if self._at_first_knot(t):
            return self._coeffs[..., 0, :]
        frac, index = self._interpret_t(t, self._coeffs.size(-2) - 1)
        prev_c = self._coeffs[..., index, :]
        next_c = self._coeffs[..., index + 1, :]
        dt = self._t[index + 1] - self._t[index]
        return prev_c + frac.unsqueeze(-1) * (next_c - prev_c) / dt.unsqueeze(-1)

    def derivative(self, t):
        _, index = self._interpret_t(t, self._coeffs.size(-2) - 1)
        dt = self._t[index + 1] - self._t[index]
        return (self._coeffs[..., index + 1, :] - self._coeffs[..., index, :]) / dt.unsqueeze(-1)


class NaturalCubicSpline(_ControlBase):
    """Natural cubic spline from ``coeffs[..., T-1, 4C] = a | b | 2c | 3d`` (natural_cubic_coeffs output)."""

    interp_name = "cubic"

    def __init__(self, coeffs, t=None, **kwargs):
        super().__init__(**kwargs)
        channels = coeffs.size(-1) // 4
        if channels * 4 != coeffs.size(-1):
            raise ValueError("Passed invalid coeffs.")
        self._setup_t(t, coeffs.size(-2) + 1, coeffs)
        self.register_buffer("_coeffs", coeffs)
        self._channels = channels

    @property
    def fused_coeffs(self):
        return self._coeffs

    @property
    def n_knots(self):
        return self._coeffs.size(-2) + 1

    @property
    def channels(self):
        return self._channels

    def _parts(self, index):
        c = self._channels
        row = self._coeffs[..., index, :]
        return row[..., :c], row[..., c:2 * c], row[..., 2 * c:3 * c], row[..., 3 * c:]

    def evaluate(self, t):
        if self._at_first_knot(t):
            return self._coeffs[..., 0, :self._channels]
        frac, index = self._interpret_t(t, self._coeffs.size(-2))
        frac = frac.unsqueeze(-1)
        a, b, two_c, three_d = self._parts(index)
        inner = 0.5 * two_c + three_d * frac / 3
        inner = b + inner * frac
        return a + inner * frac

    def derivative(self, t):
        frac, index = self._interpret_t(t, self._coeffs.size(-2))
        frac = frac.unsqueeze(-1)
        _, b, two_c, three_d = self._parts(index)
        return b + (two_c + three_d * frac) * frac
