"""Time forward kernel variants on cfg2 (run on the GPU box): python tools/time_variants.py [flags...]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ncde_amd
from ncde_amd import _lib, solver
import bench
c = dict(bench.CONFIGS["cfg2"])
B = 4096
coeffs = torch.from_numpy(bench.make_inputs(c, B, 0)).cuda()
model, fw, rw = bench.make_model(c, "cuda")
spec = model.func.fused_spec()
with torch.no_grad():
    z0 = model.initial_linear(coeffs[:, 0]).contiguous()
lib = _lib.lib()
ref = None
for flags in [int(x, 0) for x in (sys.argv[1:] or ["0", "0x400"])]:
    p = solver.build_problem(coeffs, "linear", z0, spec, "rk4", _lib.OUT_INTERVAL, flags)
    ws = torch.zeros(int(lib.ncde_workspace_bytes(ctypes.byref(p), 0)), dtype=torch.uint8, device="cuda")
    out = torch.empty(B, 2, 32, device="cuda")
    ms = ctypes.c_float()
    _lib.check(lib.ncde_time_kernel(ctypes.byref(p), 0, out.data_ptr(), None, None, ws.data_ptr(), ws.numel(), None, 5, ctypes.byref(ms)), "time")
    o = out.cpu().numpy()
    if ref is None:
        ref = o
    print("flags 0x%x: %.4f ms/launch, max |diff| vs first %.3e" % (flags, ms.value, np.abs(o - ref).max()))
