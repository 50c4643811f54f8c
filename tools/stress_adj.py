"""Repetition stress of the specialised adjoint kernels: every instantiation, N runs each, all bit-identical and within tolerance."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import gpu_util
import test_gpu_parity as tg
from ncde_amd import _lib
if os.environ.get("VARIANT"): _lib.LIB_PATH = os.path.join(ROOT, "variants", os.environ["VARIANT"])
FLAGS = _lib.FLAG_ADJOINT_V4 if "v4" in sys.argv else (64 if "bf16" in sys.argv else 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
bad = 0
for interp in ("linear", "cubic"):
    for method in ("rk4", "midpoint", "euler"):
        for seq in (False, True):
            for (B, L) in ((16, 4), (37, 7)):
                case = tg._seeded_case(interp, method, seq, B=B, L=L, C=20, H=32, HH=32, nl=3, seed=77)
                ex = case["expect"]
                for disc in (False, True):
                    kw = {"stages": case["stage_record"]} if disc else {}
                    first = None
                    errs = []
                    same = True
                    for _ in range(N):
                        iso = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=FLAGS, **kw)
                        errs.append(max(tg._grad_errors(case, iso, "bp_" if disc else "").values()))
                        if first is None:
                            first = iso
                        else:
                            same &= np.array_equal(first["dz0"], iso["dz0"]) and all(np.array_equal(first["grads"][k], iso["grads"][k]) for k in iso["grads"])
                    ok = max(errs) < 2e-5 and same
                    bad += not ok
                    print(interp, method, "seq" if seq else "final", "B%d L%d" % (B, L), "disc" if disc else "cont", "max err %.2g" % max(errs),
                          "identical" if same else "NOT identical", "" if ok else "  <-- FAIL")
print("failures:", bad)
