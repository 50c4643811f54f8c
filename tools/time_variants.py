"""Time forward/adjoint kernel variants (run on the GPU box):
    python tools/time_variants.py <cfg> <pass> [flags...]     e.g.  cfg5 0 0x1 0x1000 0x2000 0x4000
Prints ms/launch (HIP events inside the C-ABI) and the max difference of the outputs vs the first variant."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ncde_amd
from ncde_amd import _lib, solver
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
pass_ = int(sys.argv[2]) if len(sys.argv) > 2 else 0
flag_list = [int(x, 0) for x in (sys.argv[3:] or ["0"])]
iters = int(os.environ.get("ITERS", "2"))
c = dict(bench.CONFIGS[cfg])
B = int(os.environ.get("B", c["B"]))
if "L" in os.environ:
    c["L"] = int(os.environ["L"])
coeffs = torch.from_numpy(bench.make_inputs(c, B, 0)).cuda()
model, fw, rw = bench.make_model(c, "cuda")
spec = model.func.fused_spec()
interp = "cubic" if c["interpolation"] == "cubic" else "linear"
with torch.no_grad():
    z0 = model.initial_linear(coeffs[:, 0, :c["C"]]).contiguous()
lib = _lib.lib()
ref = None
for flags in flag_list:
    p = solver.build_problem(coeffs, interp, z0, spec, c["solver"], _lib.OUT_INTERVAL, flags)
    name = (lib.ncde_kernel_name(ctypes.byref(p), pass_) or b"?").decode()
    ws = torch.zeros(int(_lib.check(lib.ncde_workspace_bytes(ctypes.byref(p), pass_), "ws")), dtype=torch.uint8, device="cuda")
    ms = ctypes.c_float()
    if pass_ == 0:
        out = torch.empty(B, 2, c["H"], device="cuda")
        _lib.check(lib.ncde_time_kernel(ctypes.byref(p), 0, out.data_ptr(), None, None, ws.data_ptr(), ws.numel(), None, iters, ctypes.byref(ms)), "time")
        res = [out.cpu().numpy()]
    else:
        p0 = solver.build_problem(coeffs, interp, z0, spec, c["solver"], _lib.OUT_INTERVAL, 0)
        ws0 = torch.zeros(int(lib.ncde_workspace_bytes(ctypes.byref(p0), 0)), dtype=torch.uint8, device="cuda")
        zout = torch.empty(B, 2, c["H"], device="cuda")
        src = zout
        if pass_ == 2:
            rec = torch.empty(int(lib.ncde_stage_record_bytes(ctypes.byref(p0))) // 4, device="cuda")
            _lib.check(lib.ncde_forward_record(ctypes.byref(p0), zout.data_ptr(), rec.data_ptr(), ws0.data_ptr(), ws0.numel(), None), "fwd")
            src = rec
        else:
            _lib.check(lib.ncde_forward(ctypes.byref(p0), zout.data_ptr(), ws0.data_ptr(), ws0.numel(), None), "fwd")
        torch.manual_seed(0)
        gout = torch.randn(B, 2, c["H"], device="cuda")
        uniq = spec.unique_params(); gbuf = {id(q): torch.zeros_like(q) for q in uniq}
        g = _lib.NcdeGrads(); gz0 = torch.zeros_like(z0); g.grad_z0 = gz0.data_ptr()
        for i, (w, b) in enumerate(spec.layers):
            g.grad_layer_W[i], g.grad_layer_b[i] = gbuf[id(w)].data_ptr(), gbuf[id(b)].data_ptr()
        g.grad_Wo, g.grad_bo = gbuf[id(spec.Wo)].data_ptr(), gbuf[id(spec.bo)].data_ptr()
        _lib.check(lib.ncde_time_kernel(ctypes.byref(p), pass_, src.data_ptr(), gout.data_ptr(), ctypes.byref(g), ws.data_ptr(), ws.numel(), None, iters, ctypes.byref(ms)), "time")
        fn = lib.ncde_backward if pass_ == 2 else lib.ncde_adjoint        # full call (with the partial reduction) for the values
        _lib.check(fn(ctypes.byref(p), src.data_ptr(), gout.data_ptr(), ctypes.byref(g), ws.data_ptr(), ws.numel(), None), "adj")
        torch.cuda.synchronize()
        res = [gz0.cpu().numpy()] + [gbuf[id(q)].cpu().numpy() for q in uniq]
    if ref is None:
        ref = res
    errs = [float(np.abs(a_ - b_).max() / max(np.abs(b_).max(), 1e-30)) for a_, b_ in zip(res, ref)]
    print("flags 0x%x  %-44s %10.3f ms/launch   max rel diff vs first: %.2e" % (flags, name, ms.value, max(errs)), flush=True)
