"""Repetition stress of the specialised adjoint kernels: every instantiation, N runs each, all bit-identical and within tolerance.
usage: python tools/stress_adj.py [N] [bf16|v4] [shape=C,H,HH,nl] [flags=0x...]   (default shape 20,32,32,3; e.g. shape=4,64,64,3 for
ncde_adj_h64 -- add flags=0x2000 for its two-tile variant --, shape=20,32,32,1 for the other layer counts, shape=5,16,15,3 for a
zero-padded problem)"""
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
SHAPE = (20, 32, 32, 3)
for arg in sys.argv[1:]:
    if arg.startswith("shape="):
        SHAPE = tuple(int(v) for v in arg[6:].split(","))
    if arg.startswith("flags="):
        FLAGS |= int(arg[6:], 0)
C_, H_, HH_, NL_ = SHAPE
print("shape (C, H, HH, nl) =", SHAPE, "flags %#x" % FLAGS, "runs", N)
bad = 0
for interp in ("linear", "cubic"):
    for method in ("rk4", "midpoint", "euler"):
        for seq in (False, True):
            for (B, L) in ((16, 4), (37, 7)):
                case = tg._seeded_case(interp, method, seq, B=B, L=L, C=C_, H=H_, HH=HH_, nl=NL_, seed=77)
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
