"""development: time forward + adjoint of cfg2 for each variants/*.so (flags 0 and 64)."""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    from ncde_amd import _lib
    _lib.LIB_PATH = sys.argv[1]
    import numpy as np, torch, ncde_amd
    from ncde_amd import solver
    import bench
    c = dict(bench.CONFIGS["cfg2"]); B = 4096
    coeffs = bench.make_inputs(c, B, 0, torch.device("cuda", 0))
    model, fw, rw = bench.make_model(c, "cuda")
    spec = model.func.fused_spec()
    with torch.no_grad():
        z0 = model.initial_linear(coeffs[:, 0]).contiguous()
    lib = _lib.lib()
    for flags in [int(x) for x in sys.argv[2].split(",")]:
        p = solver.build_problem(coeffs, "linear", z0, spec, "rk4", _lib.OUT_INTERVAL, flags)
        ws = torch.zeros(int(lib.ncde_workspace_bytes(ctypes.byref(p), 0)), dtype=torch.uint8, device="cuda")
        out = torch.empty(B, 2, 32, device="cuda")
        ms = ctypes.c_float()
        _lib.check(lib.ncde_time_kernel(ctypes.byref(p), 0, out.data_ptr(), None, None, ws.data_ptr(), ws.numel(), None, 5, ctypes.byref(ms)), "time")
        tf = ms.value
        ws = torch.zeros(int(lib.ncde_workspace_bytes(ctypes.byref(p), 1)), dtype=torch.uint8, device="cuda")
        torch.manual_seed(0)
        gout = torch.randn(B, 2, 32, device="cuda") / B
        uniq = spec.unique_params(); gbuf = {id(q): torch.empty_like(q) for q in uniq}
        g = _lib.NcdeGrads(); gz0 = torch.empty_like(z0); g.grad_z0 = gz0.data_ptr()
        for i, (w, b) in enumerate(spec.layers):
            g.grad_layer_W[i], g.grad_layer_b[i] = gbuf[id(w)].data_ptr(), gbuf[id(b)].data_ptr()
        g.grad_Wo, g.grad_bo = gbuf[id(spec.Wo)].data_ptr(), gbuf[id(spec.bo)].data_ptr()
        _lib.check(lib.ncde_time_kernel(ctypes.byref(p), 1, out.data_ptr(), gout.data_ptr(), ctypes.byref(g), ws.data_ptr(), ws.numel(), None, 5, ctypes.byref(ms)), "time")
        print("%-28s flags %3d: forward %.3f ms, adjoint %.3f ms   |gz0| %.6e" % (os.path.basename(sys.argv[1]), flags, tf, ms.value, float(gz0.abs().sum())), flush=True)
else:
    for so in sorted(glob.glob(os.path.join(ROOT, "variants", "*.so"))):
        subprocess.run([sys.executable, __file__, so, "0"])
