import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import golden_util as gu, gpu_util
import test_gpu_parity as T
interp, method, seq = sys.argv[1], sys.argv[2], sys.argv[3] == "1"
case = T._seeded_case(interp, method, seq, B=21, L=9, C=20, H=32, HH=32, nl=3, seed=120)
ex = case["expect"]
for flags, name in ((0, "v2"), (8, "v1")):
    r1 = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=flags)
    r2 = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=flags)
    e = np.abs(r1["dz0"] - ex["dz0"]).max(axis=1) / np.abs(ex["dz0"]).max()
    print(name, "deterministic", np.array_equal(r1["dz0"], r2["dz0"]), "dz0 err per sample", np.array2string(e, precision=1))
    print("   grads:", {k: "%.1e" % gu.relerr(r1["grads"][k], ex["d" + k]) for k in r1["grads"]})
r = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=0)
err = np.abs(r["dz0"] - ex["dz0"]) / np.abs(ex["dz0"]).max()
bad = np.argwhere(err > 1e-4)
print("bad (sample,h):", bad.tolist()[:40])
for k in r["grads"]:
    e = np.abs(r["grads"][k] - ex["d" + k]) / np.abs(ex["d" + k]).max()
    idx = np.argwhere(e > 1e-3)
    print(k, "n_bad", len(idx), "of", e.size, "first", idx[:6].tolist())
