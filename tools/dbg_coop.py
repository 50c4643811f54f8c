"""Cooperative sweep vs the per-workgroup sweep vs the oracle, per sample / per parameter (development).
    python tools/dbg_coop.py B L C H HH nl interp method seq"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import test_gpu_parity as T
import gpu_util
from ncde_amd import _lib
B, L, C, H, HH, nl = [int(x) for x in sys.argv[1:7]]
interp, method, seq = sys.argv[7], sys.argv[8], sys.argv[9] == "1"
case = T._seeded_case(interp, method, seq, B=B, L=L, C=C, H=H, HH=HH, nl=nl, seed=900 + C)
ex = case["expect"]
print(gpu_util.kernel_names(case))
np.set_printoptions(precision=1, linewidth=220)
for tag, kw, pre in (("adjoint", {}, ""), ("discrete", {"stages": case["stage_record"]}, "bp_")):
    r = gpu_util.run_adjoint_direct(case, ex["z_out"], **kw)
    o = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=_lib.FLAG_NO_COOP, **kw)
    d = np.abs(r["dz0"] - ex[pre + "dz0"]).max(axis=1) / np.abs(ex[pre + "dz0"]).max()
    print(tag, "coop dz0 per-sample err (first 48):", d[:48], "max", d.max(), "nan", int(np.isnan(r["dz0"]).sum()))
    du = np.abs(r["dz0"] - ex[pre + "dz0"]).max(axis=0) / np.abs(ex[pre + "dz0"]).max()
    print(tag, "per-unit err:", du)
    for k, v in r["grads"].items():
        e = ex[pre + "d" + k]
        print("   ", k, "coop", float(np.abs(v - e).max() / (np.abs(e).max() + 1e-30)), "old", float(np.abs(o["grads"][k] - e).max() / (np.abs(e).max() + 1e-30)))
