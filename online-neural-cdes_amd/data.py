"""Deterministic synthetic workloads: raw irregular series and model weights.

Everything here is numpy only and independent of torch's RNG so that this container, the GPU
box and every rank of a multi-GPU run regenerate bit-identical inputs.  Control-path
coefficients are built from the raw series by the GPU builders (coefficients.py -> csrc/ncde_prepare.hip);
the numpy restatements of the reference's builders used to check them live in oracle/coeff_oracle.py
(test infrastructure).
"""
import math

import numpy as np

_U64 = np.uint64


def _splitmix64(x):
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = x + _U64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        return z ^ (z >> _U64(31))


def uniform01(seed, n, stream=0):
    """n doubles in [0, 1), a pure function of (seed, stream, index)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([seed], dtype=_U64) * _U64(0x2545F4914F6CDD1D)
                           + _U64(stream) * _U64(0xD1342543DE82EF95))[0]
        idx = np.arange(n, dtype=_U64) + base
    return (_splitmix64(idx) >> _U64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(seed, n, stream=0):
    """n standard normals (Box-Muller in fp64)."""
    u1 = uniform01(seed, n, stream=2 * stream)
    u2 = uniform01(seed, n, stream=2 * stream + 1)
    return np.sqrt(-2.0 * np.log1p(-u1)) * np.cos(2.0 * math.pi * u2)


def linear_weights(seed, stream, fan_out, fan_in):
    """(W [fan_out, fan_in], b [fan_out]) ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)), fp32 (torch Linear scale)."""
    bound = 1.0 / math.sqrt(fan_in)
    w = (uniform01(seed, fan_out * fan_in, stream=2 * stream) * 2.0 - 1.0) * bound
    b = (uniform01(seed, fan_out, stream=2 * stream + 1) * 2.0 - 1.0) * bound
    return w.reshape(fan_out, fan_in).astype(np.float32), b.astype(np.float32)


# ----------------------------------------------------------------------------------------------
# synthetic workloads (BASELINE.json configs)
# ----------------------------------------------------------------------------------------------
def synthetic_series(batch, length, channels, missing=0.0, seed=1234, batch_offset=0):
    """Raw irregular series x[batch, length, channels+1] fp32: channel 0 is time i/length, the rest
    are scaled random walks with a fraction ``missing`` of entries NaN (never in the first row).

    ``batch_offset`` selects a slice of one global deterministic batch, so rank r of an N-GPU run
    draws samples [r*B/N, (r+1)*B/N) of exactly the data a 1-GPU run would see.
    """
    n = length * channels
    out = np.empty((batch, length, channels + 1), dtype=np.float32)
    t = (np.arange(length, dtype=np.float64) / length).astype(np.float32)
    for i in range(batch):
        g = batch_offset + i
        incr = normal(seed, n, stream=2 * g).reshape(length, channels)
        walk = np.cumsum(incr, axis=0) / math.sqrt(length)
        if missing > 0.0:
            drop = uniform01(seed, n, stream=4 * g + 1_000_003).reshape(length, channels) < missing
            drop[0, :] = False
            walk = np.where(drop, np.nan, walk)
        out[i, :, 0] = t
        out[i, :, 1:] = walk.astype(np.float32)
    return out


def make_field_weights(hidden, hidden_hidden, in_channels, seed=0, layer_dims=None):
    """Vector-field parameters as a dict of fp32 arrays.

    Default architecture = the reference's OriginalVectorField (src/ncde/vector_fields/base.py:64-69,
    97-101): W0[HH,H], W1[HH,HH] (shared by all inner layers), Wo[H*C,HH].  ``layer_dims`` gives a
    general un-shared stack instead (toy CDEFunc, experiments/sim_bm_toy_example.py:10-30).
    """
    p = {}
    if layer_dims is None:
        p["W0"], p["b0"] = linear_weights(seed, 1, hidden_hidden, hidden)
        p["W1"], p["b1"] = linear_weights(seed, 2, hidden_hidden, hidden_hidden)
        p["Wo"], p["bo"] = linear_weights(seed, 3, hidden * in_channels, hidden_hidden)
    else:
        dims = [hidden] + list(layer_dims)
        for i in range(len(layer_dims)):
            p[f"W{i}"], p[f"b{i}"] = linear_weights(seed, 10 + i, dims[i + 1], dims[i])
        p["Wo"], p["bo"] = linear_weights(seed, 3, hidden * in_channels, dims[-1])
    return p


def make_variant_weights(hidden, hidden_hidden, in_channels, seed=0, kind="original", mode="matmul"):
    """Parameters of the reference's vector-field variants (src/ncde/vector_fields/gating.py, base.py:56-69):
    the inner net takes H (matmul) or H+C (evaluate / derivative) inputs, the heads have H*C (matmul) or H rows;
    'minimal' adds the sigmoid head (Wg, bg), 'gru' additionally the reset net (Wr, br)."""
    d0 = hidden if mode == "matmul" else hidden + in_channels
    rows = hidden * in_channels if mode == "matmul" else hidden
    p = {}
    p["W0"], p["b0"] = linear_weights(seed, 1, hidden_hidden, d0)
    p["W1"], p["b1"] = linear_weights(seed, 2, hidden_hidden, hidden_hidden)
    p["Wo"], p["bo"] = linear_weights(seed, 3, rows, hidden_hidden)
    if kind in ("minimal", "gru"):
        p["Wg"], p["bg"] = linear_weights(seed, 6, rows, hidden_hidden)
    if kind == "gru":
        p["Wr"], p["br"] = linear_weights(seed, 7, d0, d0)
    return p


def make_readin_weights(hidden, in_channels, out_dim, seed=0):
    wi, bi = linear_weights(seed, 4, hidden, in_channels)
    wf, bf = linear_weights(seed, 5, out_dim, hidden)
    return {"Wi": wi, "bi": bi, "Wf": wf, "bf": bf}
