"""development: phase-cycle breakdown of ncde_adj_h64 at cfg4 from the instrumented build (tools/build_f64prof.sh): run on the GPU box.
usage: python tools/prof_cfg4.py [variants/f64prof.so] [solver]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from ncde_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "variants/f64prof.so")
import ncde_amd, bench
from ncde_amd import solver
c = dict(bench.CONFIGS["cfg4"])
if len(sys.argv) > 2:
    c["solver"] = sys.argv[2]
B, dev = c["B"], torch.device("cuda", 0)
coeffs = bench.make_inputs(c, B, 0, dev)
model, fw, rw = bench.make_model(c, "cuda")
spec = model.func.fused_spec()
with torch.no_grad():
    z0 = model.initial_linear(coeffs[:, 0, :c["C"]]).contiguous()
lib = _lib.lib()
gout = torch.randn(B, 2, 64, device=dev) / B
theta = sum(q.numel() for q in spec.unique_params())
for flags, label in ((_lib.FLAG_TILED_NS1, "NS1"), (_lib.FLAG_TILED_NS2, "NS2")):
    p = solver.build_problem(coeffs, "cubic", z0, spec, c["solver"], _lib.OUT_INTERVAL, flags)
    out = torch.randn(B, 2, 64, device=dev)
    ws = solver._workspace(p, 1, dev)
    g = _lib.NcdeGrads(); gz0 = torch.zeros_like(z0); g.grad_z0 = gz0.data_ptr()
    ms = ctypes.c_float()
    _lib.check(lib.ncde_time_kernel(ctypes.byref(p), 1, out.data_ptr(), gout.data_ptr(), ctypes.byref(g), ws.data_ptr(), ws.numel(), None, 3, ctypes.byref(ms)), "time")
    n_wg = B // (16 * (2 if label == "NS2" else 1))
    gp = ws[: n_wg * theta * 4].view(torch.float32).view(n_wg, theta)[:, :32].cpu().numpy().reshape(n_wg, 4, 8)
    per = gp.mean(axis=0)
    print(label, "adjoint %.3f ms; cycles per stage by wave x [hidden fwd | out + dxl | reduce + dWo | hidden bwd | vy + bookkeeping | - | - | barrier wait]" % ms.value)
    print(np.array2string(per, precision=0, suppress_small=True), " total/stage", per.sum(axis=1).round())
