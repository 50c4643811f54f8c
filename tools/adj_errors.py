"""Development: gradient errors of the specialised adjoint variants on a seeded case (isolated on the oracle's z)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import gpu_util
import test_gpu_parity as tg
from ncde_amd import _lib
interp, method, seq = sys.argv[1], sys.argv[2], sys.argv[3] == "1"
B = int(sys.argv[4]) if len(sys.argv) > 4 else 21
L = int(sys.argv[5]) if len(sys.argv) > 5 else 9
case = tg._seeded_case(interp, method, seq, B=B, L=L, C=20, H=32, HH=32, nl=3, seed=120)
ex = case["expect"]
for fl, nm in ((0, "v3"), (_lib.FLAG_ADJOINT_V4, "v4")):
    iso = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=fl)
    print(nm, {k: float("%.3g" % e) for k, e in tg._grad_errors(case, iso).items()})
    isod = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=fl, stages=case["stage_record"])
    print(nm, "disc", {k: float("%.3g" % e) for k, e in tg._grad_errors(case, isod, "bp_").items()})
if os.environ.get("DUMP"):
    iso = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=_lib.FLAG_ADJOINT_V4)
    np.set_printoptions(linewidth=250, precision=3, suppress=True)
    d = iso["dz0"] - ex["dz0"]
    print("dz0 err by h (max over samples):"); print(np.abs(d).max(axis=0))
    print("dz0 err by sample:"); print(np.abs(d).max(axis=1))
    for k in ("W1", "W0", "b1", "b0"):
        e = np.abs(iso["grads"][k] - ex["d" + k]); r = np.abs(ex["d" + k]).max()
        print(k, "rel err by row:", (e.reshape(e.shape[0], -1).max(axis=1) / r))
    e = np.abs(iso["grads"]["Wo"] - ex["dWo"]).reshape(32, 20, 32).max(axis=2) / np.abs(ex["dWo"]).max()
    print("Wo err [h][c]:"); print(e)
