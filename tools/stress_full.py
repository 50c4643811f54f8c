"""Run-to-run reproducibility at the benchmarked size (cfg2: B = 4096, T = 399; `stress_full.py N cfg5 [L]`: cfg5 -- the XCD-cooperative
forward and sweep -- at B = 4096 and L observations): forward + adjoint through the C-ABI, N times, bit-identical and finite."""
import ctypes, hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, ncde_amd
from ncde_amd import _lib, solver
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
CFG = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
c = dict(bench.CONFIGS[CFG]); B = 4096
if len(sys.argv) > 3:
    c["L"] = int(sys.argv[3])
HID = c["H"]
coeffs = bench.make_inputs(c, B, 0, torch.device("cuda", 0))
model, fw, rw = bench.make_model(c, "cuda")
spec = model.func.fused_spec()
with torch.no_grad():
    z0 = model.initial_linear(coeffs[:, 0]).contiguous()
lib = _lib.lib()
p = solver.build_problem(coeffs, "linear", z0, spec, "rk4", _lib.OUT_INTERVAL, 0)
torch.manual_seed(0)
gout = torch.randn(B, 2, HID, device="cuda") / B
hashes = []
for i in range(N):
    ws0 = torch.zeros(int(lib.ncde_workspace_bytes(ctypes.byref(p), 0)), dtype=torch.uint8, device="cuda")
    out = torch.empty(B, 2, HID, device="cuda")
    _lib.check(lib.ncde_forward(ctypes.byref(p), out.data_ptr(), ws0.data_ptr(), ws0.numel(), None), "fwd")
    ws = torch.zeros(int(lib.ncde_workspace_bytes(ctypes.byref(p), 1)), dtype=torch.uint8, device="cuda")
    uniq = spec.unique_params(); gbuf = {id(q): torch.empty_like(q) for q in uniq}
    g = _lib.NcdeGrads(); gz0 = torch.empty_like(z0); g.grad_z0 = gz0.data_ptr()
    for k, (w, b) in enumerate(spec.layers):
        g.grad_layer_W[k], g.grad_layer_b[k] = gbuf[id(w)].data_ptr(), gbuf[id(b)].data_ptr()
    g.grad_Wo, g.grad_bo = gbuf[id(spec.Wo)].data_ptr(), gbuf[id(spec.bo)].data_ptr()
    _lib.check(lib.ncde_adjoint(ctypes.byref(p), out.data_ptr(), gout.data_ptr(), ctypes.byref(g), ws.data_ptr(), ws.numel(), None), "adj")
    torch.cuda.synchronize()
    h = hashlib.md5(out.cpu().numpy().tobytes() + gz0.cpu().numpy().tobytes() + b"".join(gbuf[id(q)].cpu().numpy().tobytes() for q in uniq)).hexdigest()[:8]
    finite = bool(torch.isfinite(out).all()) and bool(torch.isfinite(gz0).all()) and all(bool(torch.isfinite(gbuf[id(q)]).all()) for q in uniq)
    hashes.append(h if finite else "NONFINITE")
print("kernels", [(lib.ncde_kernel_name(ctypes.byref(p), k) or b"?").decode() for k in (0, 1)])
print("hashes", hashes, "-> all identical and finite:", len(set(hashes)) == 1 and "NONFINITE" not in hashes)
