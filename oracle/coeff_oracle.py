"""ORACLE (test infrastructure, NOT product code) -- numpy restatement of the reference's control-path coefficient
builders, the step right before the hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  It is the checker for the
GPU builders of csrc/ncde_prepare.hip (bit-exact comparison) and builds host-side inputs for the solve oracle.

PARITY PIN: ``oracle/gen_golden.py`` compares every function here with the imported reference (golden g8:
rectilinear and the spline bit-exact, with and without missing values; NaN linear fill <= 1e-6) and
tests/test_host_cpu.py re-checks against the committed fixtures on every run.

Reference lines restated (relative to /root/reference/modules/torchcde/torchcde):
  * rectilinear prep      interpolation_linear.py:85-128
  * NaN linear fill       interpolation_linear.py:13-82, 131-180
  * forward fill          misc.py:103-126
  * natural cubic coeffs  interpolation_cubic.py:7-53, 77-165 (tridiagonal solve: misc.py:13-67)
"""
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
import ncde_amd  # noqa: E402  (only its numpy workload generators: ncde_amd.data)

from ncde_amd.data import (linear_weights, make_field_weights, make_readin_weights, make_variant_weights, normal,  # noqa: E402,F401
                           synthetic_series, uniform01)   # so that `import coeff_oracle as data` is a superset namespace


def forward_fill(x):
    """Forward fill NaNs along axis -2 of x[..., length, channels]; leading NaNs are left in place."""
    x = np.asarray(x)
    mask = np.isnan(x)
    if not mask.any():
        return x
    length = x.shape[-2]
    idx = np.where(~mask, np.arange(length).reshape(-1, 1), 0)
    idx = np.maximum.accumulate(idx, axis=-2)
    return np.take_along_axis(x, idx, axis=-2)


def rectilinear_prep(x, time_index):
    """[..., L, C] -> [..., 2L-1, C]: forward fill, repeat every row twice, advance the time channel
    by one slot, drop the last row.  Linear interpolation of the result is the rectilinear path."""
    x = forward_fill(x)
    rep = np.repeat(x, 2, axis=-2)
    rep[..., :-1, time_index] = rep[..., 1:, time_index].copy()
    return np.ascontiguousarray(rep[..., :-1, :])


def linear_interpolation_coeffs(x, rectilinear=None):
    """Host-side mirror of torchcde.linear_interpolation_coeffs on the default integer time grid
    (interpolation_linear.py:131-180): optional rectilinear preparation (``rectilinear`` = index of the
    time channel), then every remaining NaN is filled by linear interpolation between its observed
    neighbours, with the first/last observation extended to the ends; all-NaN channels become zero."""
    x = np.array(x, dtype=np.float32, copy=True)
    if rectilinear is not None:
        assert isinstance(rectilinear, int) and 0 <= rectilinear < x.shape[-1], "bad time channel index"
        assert not np.isnan(x[..., rectilinear]).any(), "There exist nan values in the time column which is not allowed."
        x = rectilinear_prep(x, rectilinear)
    if not np.isnan(x).any():
        return x
    flat = x.reshape(-1, x.shape[-2], x.shape[-1])
    grid = np.arange(x.shape[-2], dtype=np.float64)
    for b in range(flat.shape[0]):
        for c in range(flat.shape[2]):
            col = flat[b, :, c]
            bad = np.isnan(col)
            if not bad.any():
                continue
            if bad.all():
                col[:] = 0.0
            else:
                col[bad] = np.interp(grid[bad], grid[~bad], col[~bad].astype(np.float64)).astype(np.float32)
    return x


def _natural_cubic_series_missing(x):
    """One scalar series x[L] with NaNs = missing -> (a, b, 2c, 3d)[L-1] on every unit interval, following
    interpolation_cubic.py:77-165 (``natural_cubic_coeffs``: ends filled from the first/last observation, spline on
    the observed knots, then re-expanded around the left end of every unit interval)."""
    f32 = np.float32
    L = x.shape[0]
    obs = np.where(~np.isnan(x))[0]
    if obs.size == 0:
        z = np.zeros(L - 1, dtype=f32)
        return z, z.copy(), z.copy(), z.copy()
    x = x.copy()
    x[:obs[0]] = x[obs[0]]
    x[obs[-1] + 1:] = x[obs[-1]]
    kn = np.where(~np.isnan(x))[0]
    tk = kn.astype(f32)
    xk = x[kn]
    m = kn.size
    if m == 2:
        a_k = xk[:1]
        b_k = (xk[1:] - xk[:1]) / (tk[1:] - tk[:1])
        c_k = np.zeros(1, dtype=f32)
        d_k = np.zeros(1, dtype=f32)
    else:
        td = tk[1:] - tk[:-1]
        r = (f32(1) / td).astype(f32)
        r2 = r * r
        three = f32(3) * (xk[1:] - xk[:-1])
        six = f32(2) * three
        scaled = three * r2
        diag = np.empty(m, dtype=f32)
        diag[:-1] = r
        diag[-1] = 0
        diag[1:] += r
        diag *= f32(2)
        rhs = np.empty(m, dtype=f32)
        rhs[:-1] = scaled
        rhs[-1] = 0
        rhs[1:] += scaled
        nd = np.empty(m, dtype=f32)
        nb = np.empty(m, dtype=f32)
        nd[0], nb[0] = diag[0], rhs[0]
        for i in range(1, m):
            w = f32(r[i - 1] / nd[i - 1])
            nd[i] = f32(diag[i] - f32(w * r[i - 1]))
            nb[i] = f32(rhs[i] - f32(w * nb[i - 1]))
        kd = np.empty(m, dtype=f32)
        kd[m - 1] = f32(nb[m - 1] / nd[m - 1])
        for i in range(m - 2, -1, -1):
            kd[i] = f32(f32(nb[i] - f32(r[i] * kd[i + 1])) / nd[i])
        a_k = xk[:-1]
        b_k = kd[:-1]
        c_k = ((six * r - f32(4) * kd[:-1]) - f32(2) * kd[1:]) * r
        d_k = (-six * r + f32(3) * (kd[:-1] + kd[1:])) * r2
    a = np.empty(L - 1, dtype=f32)
    b = np.empty(L - 1, dtype=f32)
    c2 = np.empty(L - 1, dtype=f32)
    d3 = np.empty(L - 1, dtype=f32)
    k = 0
    for time in range(L - 1):
        while k + 1 < m - 1 and kn[k + 1] <= time:
            k += 1
        off = f32(tk[k] - f32(time))
        a_in = f32(f32(f32(f32(0.5) * c_k[k]) - f32(f32(d_k[k] * off) / f32(3))) * off)
        a[time] = f32(a_k[k] + f32(f32(a_in - b_k[k]) * off))
        b[time] = f32(b_k[k] + f32(f32(f32(d_k[k] * off) - c_k[k]) * off))
        c2[time] = f32(c_k[k] - f32(f32(f32(2) * d_k[k]) * off))
        d3[time] = d_k[k]
    return a, b, c2, d3


def natural_cubic_coeffs(x):
    """Natural cubic spline through x[..., L, C] on the integer grid t = 0..L-1; NaNs are missing values.

    Returns [..., L-1, 4C] = a || b || 2c || 3d per piece, the layout NaturalCubicSpline consumes
    (interpolation_cubic.py:189, 294-298).  fp32 arithmetic in the reference's operation order.
    """
    x = np.asarray(x, dtype=np.float32)
    if np.isnan(x).any():
        lead = x.shape[:-2]
        L, C = x.shape[-2:]
        flat = x.reshape(-1, L, C)
        out = np.empty((flat.shape[0], L - 1, 4 * C), dtype=np.float32)
        with np.errstate(all="ignore"):
            for i in range(flat.shape[0]):
                for c in range(C):
                    a, b, c2, d3 = _natural_cubic_series_missing(flat[i, :, c])
                    out[i, :, c], out[i, :, C + c], out[i, :, 2 * C + c], out[i, :, 3 * C + c] = a, b, c2, d3
        return out.reshape(*lead, L - 1, 4 * C)
    xt = np.swapaxes(x, -1, -2)  # [..., C, L]
    length = xt.shape[-1]
    f32 = np.float32
    if length == 2:
        a = xt[..., :1]
        b = xt[..., 1:] - xt[..., :1]
        two_c = np.zeros_like(a)
        three_d = np.zeros_like(a)
    else:
        recip = np.ones(length - 1, dtype=f32)  # 1/(t[i+1]-t[i]) on the integer grid
        recip_sq = recip * recip
        three_diff = f32(3) * (xt[..., 1:] - xt[..., :-1])
        six_diff = f32(2) * three_diff
        scaled = three_diff * recip_sq
        diag = np.empty(length, dtype=f32)
        diag[:-1] = recip
        diag[-1] = 0
        diag[1:] += recip
        diag *= f32(2)
        rhs = np.empty_like(xt)
        rhs[..., :-1] = scaled
        rhs[..., -1] = 0
        rhs[..., 1:] += scaled
        # Thomas algorithm, upper = lower = recip
        new_b = [rhs[..., 0]]
        new_d = [np.broadcast_to(diag[0], rhs[..., 0].shape).astype(f32)]
        for i in range(1, length):
            w = recip[i - 1] / new_d[i - 1]
            new_d.append((diag[i] - w * recip[i - 1]).astype(f32))
            new_b.append((rhs[..., i] - w * new_b[i - 1]).astype(f32))
        outs = [None] * length
        outs[length - 1] = new_b[length - 1] / new_d[length - 1]
        for i in range(length - 2, -1, -1):
            outs[i] = (new_b[i] - recip[i] * outs[i + 1]) / new_d[i]
        kd = np.stack(outs, axis=-1).astype(f32)
        a = xt[..., :-1]
        b = kd[..., :-1]
        two_c = (six_diff * recip - f32(4) * kd[..., :-1] - f32(2) * kd[..., 1:]) * recip
        three_d = (-six_diff * recip + f32(3) * (kd[..., :-1] + kd[..., 1:])) * recip_sq
    parts = [np.swapaxes(p, -1, -2) for p in (a, b, two_c, three_d)]
    return np.ascontiguousarray(np.concatenate(parts, axis=-1).astype(f32))



def make_rectilinear_coeffs(batch, length, channels, missing=0.3, seed=1234, batch_offset=0):
    """coeffs[batch, 2*length-1, channels+1] for the rectilinear configs (cfg2/3/5)."""
    x = synthetic_series(batch, length, channels, missing=missing, seed=seed, batch_offset=batch_offset)
    return rectilinear_prep(x, time_index=0)


def make_linear_coeffs(batch, length, channels, seed=1234, batch_offset=0):
    """coeffs[batch, length, channels+1] for plain linear interpolation (no missing values)."""
    return synthetic_series(batch, length, channels, missing=0.0, seed=seed, batch_offset=batch_offset)


def make_cubic_coeffs(batch, length, channels, seed=1234, batch_offset=0):
    """coeffs[batch, length-1, 4*(channels+1)] natural cubic (cfg4)."""
    x = synthetic_series(batch, length, channels, missing=0.0, seed=seed, batch_offset=batch_offset)
    return natural_cubic_coeffs(x)


