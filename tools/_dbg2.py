import os, sys
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from ncde_amd import _lib
if os.environ.get("VARIANT"): _lib.LIB_PATH = os.path.join(ROOT, "variants", os.environ["VARIANT"])
import gpu_util, golden_util as gu
import test_gpu_parity as T
case = T._seeded_case("linear", "rk4", False, B=21, L=9, C=20, H=32, HH=32, nl=3, seed=120)
ex = case["expect"]
ref = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=0, stages=case["stage_record"])
for i in range(12):
    r = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=64, stages=case["stage_record"])
    d = np.abs(r["dz0"] - ref["dz0"]) / np.abs(ref["dz0"]).max()
    bad = np.argwhere(d > 1e-5)
    print(i, "max rel diff dz0 %.1e" % d.max(), "bad entries (sample, h):", bad.tolist()[:12], {k: "%.1e" % (np.abs(r["grads"][k] - ref["grads"][k]).max() / np.abs(ref["grads"][k]).max()) for k in r["grads"]})
