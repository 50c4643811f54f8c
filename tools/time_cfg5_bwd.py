"""cfg5 backward (sweep + pass B) kernel time through ncde_time_kernel on a SHORT series (per-stage time is what matters):
usage: time_cfg5_bwd.py [L=60] [flags,flags,...] [B=4096]   -- one line per flag set (forward and backward kernel time)"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import ncde_amd, bench
from ncde_amd import _lib, solver
if os.environ.get("NCDE_LIB"): _lib.LIB_PATH = os.path.join(ROOT, os.environ["NCDE_LIB"])      # (a variant build, e.g. variants/x.so)
c = dict(bench.CONFIGS["cfg5"]); c["L"] = int(sys.argv[1]) if len(sys.argv) > 1 else 60
flag_sets = [int(f, 0) for f in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
coeffs = bench.make_inputs(c, B, 0, torch.device("cuda", 0))
model, fw, rw = bench.make_model(c, "cuda")
spec = model.func.fused_spec()
with torch.no_grad():
    z0 = model.initial_linear(coeffs[:, 0]).contiguous()
lib = _lib.lib()
T = coeffs.shape[1]
for flags in flag_sets:
    p = solver.build_problem(coeffs, "linear", z0, spec, "rk4", _lib.OUT_INTERVAL, flags)
    out = torch.empty(B, 2, c["H"], device="cuda")
    ws0 = torch.zeros(int(lib.ncde_workspace_bytes(ctypes.byref(p), 0)), dtype=torch.uint8, device="cuda")
    _lib.check(lib.ncde_forward(ctypes.byref(p), out.data_ptr(), ws0.data_ptr(), ws0.numel(), None), "fwd")
    msf = ctypes.c_float()
    _lib.check(lib.ncde_time_kernel(ctypes.byref(p), 0, out.data_ptr(), None, None, ws0.data_ptr(), ws0.numel(), None, 2, ctypes.byref(msf)), "time fwd")
    gout = torch.randn_like(out)
    uniq = spec.unique_params()
    gbuf = {id(q): torch.empty_like(q) for q in uniq}
    g = _lib.NcdeGrads()
    gz0 = torch.empty_like(z0)
    g.grad_z0 = gz0.data_ptr()
    for i, (w, b) in enumerate(spec.layers):
        g.grad_layer_W[i], g.grad_layer_b[i] = gbuf[id(w)].data_ptr(), gbuf[id(b)].data_ptr()
    g.grad_Wo, g.grad_bo = gbuf[id(spec.Wo)].data_ptr(), gbuf[id(spec.bo)].data_ptr()
    ws = torch.zeros(int(lib.ncde_workspace_bytes(ctypes.byref(p), 1)), dtype=torch.uint8, device="cuda")
    ms = ctypes.c_float()
    _lib.check(lib.ncde_time_kernel(ctypes.byref(p), 1, out.data_ptr(), gout.data_ptr(), ctypes.byref(g), ws.data_ptr(), ws.numel(), None, 2, ctypes.byref(ms)), "time")
    name = (lib.ncde_kernel_name(ctypes.byref(p), 1) or b"?").decode()
    print("flags 0x%x  B=%d %s: forward %.2f ms, backward %.2f ms -> %.1f us per stage (T=%d)" % (flags, B, name, msf.value, ms.value, ms.value * 1e3 / ((T - 1) * 4), T))
