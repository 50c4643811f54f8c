import os, sys
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from ncde_amd import _lib
if os.environ.get("VARIANT"): _lib.LIB_PATH = os.path.join(ROOT, "variants", os.environ["VARIANT"])
import gpu_util, golden_util as gu
import test_gpu_parity as T
for (interp, method, seq, B, L) in (("cubic", "euler", False, 16, 4), ("linear", "rk4", False, 16, 4)):
    case = T._seeded_case(interp, method, seq, B=B, L=L, C=20, H=32, HH=32, nl=3, seed=77)
    ex = case["expect"]
    ref = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=64)
    for i in range(16):
        r = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=0)
        d = np.abs(r["dz0"] - ref["dz0"]) / np.abs(ref["dz0"]).max()
        bad = np.argwhere(d > 1e-5)
        g = {k: (np.abs(r["grads"][k] - ref["grads"][k]).max() / np.abs(ref["grads"][k]).max()) for k in r["grads"]}
        if d.max() > 1e-5 or max(g.values()) > 1e-5:
            print(interp, method, i, "dz0 %.1e" % d.max(), "bad (sample,h):", bad.tolist()[:10], {k: "%.0e" % v for k, v in g.items()})
            for k in ("Wo", "bo", "W1", "W0"):
                dd = np.abs(r["grads"][k] - ref["grads"][k]) / np.abs(ref["grads"][k]).max()
                bb = np.argwhere(dd > 1e-5)
                print("     ", k, "bad count", len(bb), "first", bb[:6].tolist())
print("done")
