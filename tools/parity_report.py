"""Print the HIP-vs-reference-golden error table (run on the GPU box): python tools/parity_report.py [flags]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import golden_util as gu  # noqa: E402
import gpu_util  # noqa: E402

flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
print(f"{'case':28s} {'z':>9s} {'dz0':>9s}  dtheta...")
for name in gu.SOLVE_CASES:
    case = gu.load_case(name)
    res = gpu_util.run_case(case, flags=flags)
    ex = case["expect"]
    errs = []
    for pname in case["meta"]["param_names"]:
        g = res["grads"][pname]
        if "d" + pname in ex:
            errs.append((pname, gu.relerr(g, ex["d" + pname])))
        else:
            errs.append((pname, gu.relerr(g[::16], ex["d" + pname + "__rows16"])))
    print(f"{name:28s} {gu.relerr(res['z_out'], ex['z_out']):9.2e} {gu.relerr(res['dz0'], ex['dz0']):9.2e}  "
          + " ".join(f"{n}:{e:.1e}" for n, e in errs))
