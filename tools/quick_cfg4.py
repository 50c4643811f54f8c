"""Quick check on the GPU box: cfg4 adjoint kernel time for the dispatch variants (in-sweep H = 64 kernel with one / two sample tiles
per workgroup, split-fp16 / fp32 forward side; the batch-tiled sweep + pass B it replaced) and their mutual agreement.
usage: python tools/quick_cfg4.py [B] [cfg]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import ncde_amd, bench
from ncde_amd import _lib, solver
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg4"
c = dict(bench.CONFIGS[cfg])
if len(sys.argv) > 3:
    c["solver"] = sys.argv[3]
B = int(sys.argv[1]) if len(sys.argv) > 1 else c["B"]
dev = torch.device("cuda", 0)
coeffs = bench.make_inputs(c, B, 0, dev)
model, fw, rw = bench.make_model(c, "cuda")
spec = model.func.fused_spec()
interp = "cubic" if c["interpolation"] == "cubic" else "linear"
with torch.no_grad():
    z0 = model.initial_linear(coeffs[:, 0, :c["C"]]).contiguous()
lib = _lib.lib()
H = c["H"]
torch.manual_seed(0)
gout = torch.randn(B, 2, H, device=dev) / B
ref = None
for flags, label in ((0, "auto"), (_lib.FLAG_TILED_NS1, "NS1"), (_lib.FLAG_TILED_NS2, "NS2"), (_lib.FLAG_TILED_NS2 | _lib.FLAG_FP32_MFMA, "NS2 fp32"),
                     (_lib.FLAG_FORCE_TILED, "tiled sweep + pass B")):
    p = solver.build_problem(coeffs, interp, z0, spec, c["solver"], _lib.OUT_INTERVAL, flags)
    ws0 = solver._workspace(p, 0, dev)
    out = torch.empty(B, 2, H, device=dev)
    _lib.check(lib.ncde_forward(ctypes.byref(p), out.data_ptr(), ws0.data_ptr(), ws0.numel(), None), "fwd")
    ws = solver._workspace(p, 1, dev)
    uniq = spec.unique_params(); gbuf = {id(q): torch.zeros_like(q) for q in uniq}
    g = _lib.NcdeGrads(); gz0 = torch.zeros_like(z0); g.grad_z0 = gz0.data_ptr()
    for i, (w, b) in enumerate(spec.layers):
        g.grad_layer_W[i], g.grad_layer_b[i] = gbuf[id(w)].data_ptr(), gbuf[id(b)].data_ptr()
    g.grad_Wo, g.grad_bo = gbuf[id(spec.Wo)].data_ptr(), gbuf[id(spec.bo)].data_ptr()
    ms = ctypes.c_float()
    _lib.check(lib.ncde_time_kernel(ctypes.byref(p), 1, out.data_ptr(), gout.data_ptr(), ctypes.byref(g), ws.data_ptr(), ws.numel(), None, 5, ctypes.byref(ms)), "time")
    _lib.check(lib.ncde_adjoint(ctypes.byref(p), out.data_ptr(), gout.data_ptr(), ctypes.byref(g), ws.data_ptr(), ws.numel(), None), "adj")
    torch.cuda.synchronize()
    res = {"dz0": gz0.cpu().numpy()}
    res.update({"p%d" % i: gbuf[id(q)].cpu().numpy() for i, q in enumerate(uniq)})
    name = (lib.ncde_kernel_name(ctypes.byref(p), 1) or b"?").decode()
    line = "%-22s %-70s adjoint %.3f ms" % (label, name, ms.value)
    if ref is None:
        ref = res
    else:
        line += "   max rel diff vs auto: " + " ".join("%s %.1e" % (k, np.abs(res[k] - ref[k]).max() / (np.abs(ref[k]).max() + 1e-30)) for k in res)
    print(line, flush=True)
