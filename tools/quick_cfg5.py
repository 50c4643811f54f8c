import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import ncde_amd, bench
from ncde_amd import _lib, solver
c = dict(bench.CONFIGS["cfg5"]); c["L"] = 60     # short series: the per-stage time is what matters
B = 4096
coeffs = bench.make_inputs(c, B, 0, torch.device("cuda", 0))
model, fw, rw = bench.make_model(c, "cuda")
spec = model.func.fused_spec()
with torch.no_grad():
    z0 = model.initial_linear(coeffs[:, 0]).contiguous()
lib = _lib.lib()
T = coeffs.shape[1]
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    p = solver.build_problem(coeffs, "linear", z0, spec, "rk4", _lib.OUT_INTERVAL, 0)
    ws = torch.zeros(int(lib.ncde_workspace_bytes(ctypes.byref(p), 0)), dtype=torch.uint8, device="cuda")
    out = torch.empty(B, 2, c["H"], device="cuda")
    ms = ctypes.c_float()
    _lib.check(lib.ncde_time_kernel(ctypes.byref(p), 0, out.data_ptr(), None, None, ws.data_ptr(), ws.numel(), None, 2, ctypes.byref(ms)), "time")
    print("forward %.2f ms -> %.1f us per stage" % (ms.value, ms.value * 1e3 / ((T - 1) * 4)))
