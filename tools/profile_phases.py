"""Phase-cycle breakdown of the fast forward kernel (s_memtime instrumented variant): run on the GPU box."""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ncde_amd
from ncde_amd import _lib, solver
sys.path.insert(0, ROOT)
import bench
c = dict(bench.CONFIGS["cfg2"])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
coeffs = bench.make_inputs(c, B, 0, torch.device("cuda", 0))
model, fw, rw = bench.make_model(c, "cuda")
spec = model.func.fused_spec()
with torch.no_grad():
    z0 = model.initial_linear(coeffs[:, 0]).contiguous()
lib = _lib.lib()
for flags, label in ((4, "fp32-mfma plain"), (0x104, "fp32-mfma instrumented"), (64, "split-bf16 plain"), (0x140, "split-bf16 instrumented"),
                     (0, "split-fp16 plain (default)"), (0x100, "split-fp16 instrumented")):
    p = solver.build_problem(coeffs, "linear", z0, spec, "rk4", _lib.OUT_INTERVAL, flags)
    ws = torch.zeros(int(lib.ncde_workspace_bytes(ctypes.byref(p), 0)), dtype=torch.uint8, device="cuda")
    out = torch.empty(B, 2, 32, device="cuda")
    ms = ctypes.c_float()
    _lib.check(lib.ncde_time_kernel(ctypes.byref(p), 0, out.data_ptr(), None, None, ws.data_ptr(), ws.numel(), None, 3, ctypes.byref(ms)), "time")
    print(label, "ms/launch", ms.value)
    if flags & 0x100:
        nwv = 4
        cyc = ws[: (B // 16) * nwv * 4 * 8].view(torch.int64).view(-1, nwv, 4).cpu().numpy().astype(np.float64)
        stages = 398 * 4
        per = cyc.mean(axis=0) / stages      # [wave][phase]
        print("cycles per stage by wave x phase (hidden | out+tanh | rk+exchange | -):")
        print(np.array2string(per, precision=0))
        tot = per[:, :3].sum(axis=1)
        print("total cycles/stage per wave", tot, "=> effective clock %.2f GHz" % (tot.mean() * stages / (ms.value * 1e-3) / 1e9))

# ---- adjoint ----
theta = 32*32+32+32*32+32+640*32+640
for flags, label in ((64, "adj v3 all split-bf16 plain"), (0x140, "adj v3 all split-bf16 instrumented"), (0, "adj v3 default plain"), (0x100, "adj v3 default instrumented")):
    p = solver.build_problem(coeffs, "linear", z0, spec, "rk4", _lib.OUT_INTERVAL, flags)
    ws = torch.zeros(int(lib.ncde_workspace_bytes(ctypes.byref(p), 1)), dtype=torch.uint8, device="cuda")
    out = torch.randn(B, 2, 32, device="cuda"); gout = torch.randn(B, 2, 32, device="cuda")
    uniq = spec.unique_params(); gbuf = {id(q): torch.empty_like(q) for q in uniq}
    g = _lib.NcdeGrads(); gz0 = torch.empty_like(z0); g.grad_z0 = gz0.data_ptr()
    for i, (w, b) in enumerate(spec.layers):
        g.grad_layer_W[i], g.grad_layer_b[i] = gbuf[id(w)].data_ptr(), gbuf[id(b)].data_ptr()
    g.grad_Wo, g.grad_bo = gbuf[id(spec.Wo)].data_ptr(), gbuf[id(spec.bo)].data_ptr()
    ms = ctypes.c_float()
    _lib.check(lib.ncde_time_kernel(ctypes.byref(p), 1, out.data_ptr(), gout.data_ptr(), ctypes.byref(g), ws.data_ptr(), ws.numel(), None, 3, ctypes.byref(ms)), "time")
    print(label, "ms/launch", ms.value)
    if flags & 0x100:
        nwg = B // 16
        off = (nwg * theta + 64) * 4
        cyc = ws[off: off + nwg * 8 * 6 * 8].view(torch.int64).view(-1, 8, 6).cpu().numpy().astype(np.float64)
        per = cyc.mean(axis=0) / (398 * 4)
        if not flags & 32:
            print("chain rows 0-3: recompute | out tiles | wait barrier A | reduce+hidden bwd+vy | rk+exchange(+barrier B)")
            print("grad  rows 4-7: dw_hidden(prev) | wait group0 | dxl0+dWo blocks | wait group1 | dxl1+red | A+dWo rest+B")
        else:
            print("Y rows 0-3: hidden+masks+images | out tiles + t blocks | rk + stores | barrier E | - | -")
            print("A rows 4-7: dP, Wo^T dP, dWo blocks | reduction hand-off | sum + hidden bwd + dW + vy | rk | barrier E | -")
        print(np.array2string(per[:, :6], precision=0))
        print("total", per[:, :6].sum(axis=1))
