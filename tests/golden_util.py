"""Load golden fixtures (outputs of the imported reference, written by oracle/gen_golden.py)."""
import json
import os

import numpy as np

import coeff_oracle as data  # noqa: F401  workload generators (ncde_amd.data) + numpy restatement of the coefficient builders
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SOLVE_CASES = [
    "g1_toy_rk4_seq", "g1_toy_midpoint_seq", "g1_toy_euler_seq",
    "g2_rect_rk4_final", "g2_rect_rk4_seq", "g2_rect_midpoint_final",
    "g3_cubic_midpoint_final", "g3_cubic_midpoint_seq", "g3_cubic_rk4_final", "g3_cubic_rk4_seq",
    "g4_wide_rk4_final",
    "g6_nl1_rk4_seq", "g6_nl4_rk4_seq", "g6_T2_rk4_final", "g6_linear_euler_seq",
]


VARIANT_CASES = ["g9_%s_%s_%s" % (k, m, tail) for k in ("original", "minimal", "gru") for m in ("matmul", "evaluate", "derivative")
                 for tail in ("linear_rk4", "cubic_midpoint") if not (k == "original" and m == "matmul")]


def _z0_from(c0, rw):
    return (c0 @ rw["Wi"].T + rw["bi"]).astype(np.float32)


def load_case(name):
    """-> dict(meta, coeffs, z0, params{...}, layers=[(Wname,bname)...], expect{z_out,dz0,grad_out,d*})."""
    f = np.load(os.path.join(GOLD, name + ".npz"))
    meta = json.loads(str(f["meta"]))
    expect = {k: f[k] for k in f.files if k not in ("meta", "coeffs", "z0") and not k.startswith("p_")}
    if "coeffs" in f.files:
        src = f
    elif "inputs_in" in meta:
        src = np.load(os.path.join(GOLD, meta["inputs_in"] + ".npz"))
    else:
        src = None
    if src is not None:
        coeffs, z0 = src["coeffs"], src["z0"]
        params = {k[2:]: src[k] for k in src.files if k.startswith("p_")}
    else:
        assert name == "g4_wide_rk4_final"
        coeffs = data.make_rectilinear_coeffs(4, 40, 79, missing=0.6, seed=99)
        params = data.make_field_weights(128, 128, 80, seed=3)
        z0 = _z0_from(coeffs[:, 0], data.make_readin_weights(128, 80, 1, seed=3))
    d = meta["dims"]
    if meta["field"] in ("original", "variant"):
        layers = [("W0", "b0")] + [("W1", "b1")] * (d["nl"] - 1)
    else:
        layers = [("W0", "b0"), ("W1", "b1")]
    return {"meta": meta, "coeffs": coeffs, "z0": z0, "params": params, "layers": layers, "expect": expect,
            "H": d["H"], "C": d["C"]}


def oracle_field(case):
    import ncde_oracle as orc
    import torch
    t = {k: torch.from_numpy(v) for k, v in case["params"].items()}
    m = case["meta"]
    return orc.Field([(t[w], t[b]) for w, b in case["layers"]], t["Wo"], t["bo"], case["H"], case["C"],
                     m.get("field_kind", "original"), m.get("field_mode", "matmul"), t.get("Wg"), t.get("bg"), t.get("Wr"), t.get("br"))


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
