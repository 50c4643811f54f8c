import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, ncde_amd
from ncde_amd import _lib, solver
import bench
c = dict(bench.CONFIGS["cfg2"]); B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
coeffs = bench.make_inputs(c, B, 0, torch.device("cuda", 0))
model, fw, rw = bench.make_model(c, "cuda")
spec = model.func.fused_spec()
with torch.no_grad():
    z0 = model.initial_linear(coeffs[:, 0]).contiguous()
lib = _lib.lib()
p = solver.build_problem(coeffs, "linear", z0, spec, "rk4", _lib.OUT_INTERVAL, 0)
nwg = (B + 15) // 16
fb = ((nwg * 4 + 255) // 256) * 256
ws = torch.zeros(int(lib.ncde_workspace_bytes(ctypes.byref(p), 0)), dtype=torch.uint8, device="cuda")
out = torch.empty(B, 2, 32, device="cuda")
_lib.check(lib.ncde_forward(ctypes.byref(p), out.data_ptr(), ws.data_ptr(), ws.numel(), None), "fwd")
torch.cuda.synchronize()
f = ws[ws.numel() - fb:].view(torch.int32)[:nwg].cpu().numpy()
print("forward: faulted workgroups %d of %d" % (int((f != 0).sum()), nwg), f[:8])
ws = torch.zeros(int(lib.ncde_workspace_bytes(ctypes.byref(p), 1)), dtype=torch.uint8, device="cuda")
torch.manual_seed(0)
gout = torch.randn(B, 2, 32, device="cuda") / B
uniq = spec.unique_params(); gbuf = {id(q): torch.empty_like(q) for q in uniq}
g = _lib.NcdeGrads(); gz0 = torch.empty_like(z0); g.grad_z0 = gz0.data_ptr()
for i, (w, b) in enumerate(spec.layers):
    g.grad_layer_W[i], g.grad_layer_b[i] = gbuf[id(w)].data_ptr(), gbuf[id(b)].data_ptr()
g.grad_Wo, g.grad_bo = gbuf[id(spec.Wo)].data_ptr(), gbuf[id(spec.bo)].data_ptr()
_lib.check(lib.ncde_adjoint(ctypes.byref(p), out.data_ptr(), gout.data_ptr(), ctypes.byref(g), ws.data_ptr(), ws.numel(), None), "adj")
torch.cuda.synchronize()
f = ws[ws.numel() - fb:].view(torch.int32)[:nwg].cpu().numpy()
print("adjoint: faulted workgroups %d of %d" % (int((f != 0).sum()), nwg), f[:8])
