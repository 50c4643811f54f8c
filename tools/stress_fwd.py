"""Repetition stress of the specialised forward kernels (default: split-fp16 + range-fault re-execution): N runs, bit-identical, within tolerance."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import gpu_util, golden_util as gu
import test_gpu_parity as tg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bad = 0
for shape in ((20, 32, 32, 3), (4, 64, 64, 3)):
    C, H, HH, nl = shape
    for interp in ("linear", "cubic"):
        for method in ("rk4", "midpoint", "euler"):
            for seq in (False, True):
                for (B, L) in ((16, 4), (37, 7)):
                    case = tg._seeded_case(interp, method, seq, B=B, L=L, C=C, H=H, HH=HH, nl=nl, seed=77)
                    ex = case["expect"]
                    first, same, errs = None, True, []
                    for _ in range(N):
                        r = gpu_util.run_case(case, need_grads=False)
                        errs.append(gu.relerr(r["z_out"], ex["z_out"]))
                        if first is None:
                            first = r
                        else:
                            same &= np.array_equal(first["z_out"], r["z_out"])
                    ok = max(errs) < 2e-5 and same
                    bad += not ok
                    if not ok:
                        print(shape, interp, method, seq, B, L, "max err %.2g" % max(errs), "identical" if same else "NOT identical", " <-- FAIL")
print("kernel", first["kernels"][0], "failures:", bad)
