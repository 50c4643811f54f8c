"""development: cycles per pair-stage of ncde_dwo_h2 by phase, from the instrumented build (tools/build_dw2prof.sh).
usage: python tools/prof_dw2.py [variants/dw2prof.so] [L] [flags]   (the gradients of such a build are garbage)"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from ncde_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "variants/dw2prof.so")
import ncde_amd, bench
from ncde_amd import solver
c = dict(bench.CONFIGS["cfg5"]); c["L"] = int(sys.argv[2]) if len(sys.argv) > 2 else 60
flags = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
B = c["B"]; dev = torch.device("cuda", 0)
coeffs = bench.make_inputs(c, B, 0, dev)
model, fw, rw = bench.make_model(c, "cuda")
spec = model.func.fused_spec()
with torch.no_grad():
    z0 = model.initial_linear(coeffs[:, 0]).contiguous()
lib = _lib.lib()
p = solver.build_problem(coeffs, "linear", z0, spec, "rk4", _lib.OUT_INTERVAL, flags)
out = torch.randn(B, 2, c["H"], device=dev); gout = torch.randn(B, 2, c["H"], device=dev) / B
ws = solver._workspace(p, 1, dev)
g = _lib.NcdeGrads(); gz0 = torch.zeros_like(z0); g.grad_z0 = gz0.data_ptr()
ms = ctypes.c_float()
_lib.check(lib.ncde_time_kernel(ctypes.byref(p), 1, out.data_ptr(), gout.data_ptr(), ctypes.byref(g), ws.data_ptr(), ws.numel(), None, 1, ctypes.byref(ms)), "time")
nwv = 8
per = gz0.view(-1)[:1280 * nwv * 8].view(1280, nwv, 8)[:, :, :5].cpu().numpy()
print("backward %.2f ms; ncde_dwo_h2 cycles per pair-stage, mean over workgroups, by wave x [issue DMA | P + epilogue | dWo | DMA wait | barrier]" % ms.value)
print(np.array2string(per.mean(axis=0), precision=0, suppress_small=True), " total", per.mean(axis=0).sum(axis=1).round())
