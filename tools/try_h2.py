"""Split-fp16 vs split-bf16 forward (cfg2 shapes): time per launch and error against an fp64 torch re-statement of the same RK4 solve."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ncde_amd
from ncde_amd import _lib, solver
import bench
c = dict(bench.CONFIGS["cfg2"])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
coeffs = bench.make_inputs(c, B, 0, torch.device("cuda", 0))
model, fw, rw = bench.make_model(c, "cuda")
if scale != 1.0:
    with torch.no_grad():
        for q in model.func.parameters():
            q.mul_(scale)
spec = model.func.fused_spec()
with torch.no_grad():
    z0 = model.initial_linear(coeffs[:, 0]).contiguous()
lib = _lib.lib()


def ref64():
    W = [(w.double(), b.double()) for w, b in spec.layers]
    Wo, bo = spec.Wo.double(), spec.bo.double()
    X = coeffs.double()

    def f(z):
        h = z
        h = torch.relu(h @ W[0][0].T + W[0][1])
        for _ in range(c["nl"] - 1):      # the inner layer is ONE shared module applied nl - 1 times
            h = torch.relu(h @ W[1][0].T + W[1][1])
        return torch.tanh(h @ Wo.T + bo).view(z.shape[0], 32, 20)
    z = z0.double()
    T = X.shape[1]
    for n in range(T - 1):
        dX = (X[:, n + 1] - X[:, n]).unsqueeze(-1)
        dXl = (X[:, n] - X[:, n - 1]).unsqueeze(-1) if n else dX      # an exact knot belongs to the piece on its left
        k1 = (f(z) @ dXl).squeeze(-1)
        k2 = (f(z + k1 / 3) @ dX).squeeze(-1)
        k3 = (f(z + (k2 - k1 / 3)) @ dX).squeeze(-1)
        k4 = (f(z + (k1 - k2 + k3)) @ dX).squeeze(-1)
        z = z + (k1 + 3 * (k2 + k3) + k4) / 8
    return z


with torch.no_grad():
    zr = ref64()
print("max |z_T| %.3f" % float(zr.abs().max()))
for flags, label in ((4, "fp32-mfma"), (64, "split-bf16"), (0, "split-fp16")):      # 0 = default, 64 = NCDE_FLAG_SPLIT_BF16
    p = solver.build_problem(coeffs, "linear", z0, spec, "rk4", _lib.OUT_INTERVAL, flags)
    ws = torch.zeros(int(lib.ncde_workspace_bytes(ctypes.byref(p), 0)), dtype=torch.uint8, device="cuda")
    out = torch.empty(B, 2, 32, device="cuda")
    ms = ctypes.c_float()
    _lib.check(lib.ncde_time_kernel(ctypes.byref(p), 0, out.data_ptr(), None, None, ws.data_ptr(), ws.numel(), None, 5, ctypes.byref(ms)), "time")
    err = (out[:, 1].double() - zr).abs()
    if flags == 4:
        o32 = out.clone()
    d32 = float((out - o32).abs().max())
    print("%-11s %.3f ms/launch   |z_T - fp64| max %.2e  mean %.2e   |z_T - fp32 kernel| max %.2e" % (label, ms.value, float(err.max()), float(err.mean()), d32))
