"""Range-fault path of the split-fp16 kernels under a sweep of outlier magnitudes (incl. inf and NaN) in one sample: forward and
adjoint outputs must equal the all-split-bf16 run on the faulted tile (NaN-aware), stay untouched elsewhere, and nothing may hang."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import gpu_util
import test_gpu_parity as T
from ncde_amd import _lib
case = T._seeded_case("linear", "rk4", True, B=37, L=6, C=20, H=32, HH=32, nl=3, seed=555)
plain = gpu_util.run_case(case, need_grads=False)["z_out"]
eq = lambda a, b: np.array_equal(a, b, equal_nan=True)
bad = 0
for scale in (1e2, 1e3, 3e3, 1e4, 1e5, 1e8, 1e30, np.inf, np.nan):
    big = dict(case); big["z0"] = case["z0"].copy()
    big["z0"][21] = big["z0"][21] * scale if np.isfinite(scale) else scale
    r16 = gpu_util.run_case(big, need_grads=False)["z_out"]
    rbf = gpu_util.run_case(big, flags=_lib.FLAG_SPLIT_BF16, need_grads=False)["z_out"]
    same_tile = eq(r16[16:32], rbf[16:32])
    others = eq(r16[:16], plain[:16]) and eq(r16[32:], plain[32:])
    def rel(a, b):      # per sample, relative to that sample's largest finite magnitude; NaN / inf patterns must coincide
        if not np.array_equal(np.isfinite(a), np.isfinite(b)):
            return np.inf
        fin = np.isfinite(b)
        d = np.where(fin, np.abs(np.where(fin, a, 0) - np.where(fin, b, 0)), 0.0)
        sc = np.where(fin, np.abs(b), 0.0).reshape(b.shape[0], -1).max(axis=1)
        return float((d.reshape(b.shape[0], -1).max(axis=1) / np.maximum(sc, 1e-30)).max())
    close = rel(r16[16:32], rbf[16:32])
    # adjoint on the forward's own output
    z = rbf.copy()
    a16 = gpu_util.run_adjoint_direct(big, z)
    abf = gpu_util.run_adjoint_direct(big, z, flags=_lib.FLAG_SPLIT_BF16)
    adj_tile = eq(a16["dz0"][16:32], abf["dz0"][16:32])
    adj_close = rel(a16["dz0"], abf["dz0"])
    ok = others and (same_tile or close < 2e-5) and (adj_tile or adj_close < 2e-5)
    bad += not ok
    print("outlier x %-8g forward: faulted tile %s bf16 run (max rel %.1e), other tiles untouched %s | adjoint tile %s (max rel %.1e)%s" % (
        scale, "==" if same_tile else "~", close, others, "==" if adj_tile else "~", adj_close, "" if ok else "  <-- FAIL"))
print("failures:", bad)
