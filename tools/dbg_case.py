"""Per-sample error report of one seeded case: dispatched kernels vs the oracle, continuous adjoint and exact discrete backward.
    python tools/dbg_case.py B L C H HH nl interp method seq [flags]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import test_gpu_parity as T
import gpu_util
B, L, C, H, HH, nl = [int(x) for x in sys.argv[1:7]]
interp, method, seq = sys.argv[7], sys.argv[8], sys.argv[9] == "1"
flags = int(sys.argv[10], 0) if len(sys.argv) > 10 else 0
case = T._seeded_case(interp, method, seq, B=B, L=L, C=C, H=H, HH=HH, nl=nl, seed=500 + B)
ex = case["expect"]
print(gpu_util.kernel_names(case, flags=flags))
for tag, kw, pre in (("adjoint", {}, ""), ("discrete", {"stages": case["stage_record"]}, "bp_")):
    r = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=flags, **kw)
    d = np.abs(r["dz0"] - ex[pre + "dz0"]).max(axis=1) / np.abs(ex[pre + "dz0"]).max()
    print(tag, "dz0 per-sample err:", np.array2string(d, precision=1, max_line_width=200))
    for k, v in r["grads"].items():
        e = ex[pre + "d" + k]
        print("   ", k, float(np.abs(v - e).max() / (np.abs(e).max() + 1e-30)))
