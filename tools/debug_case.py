import os, sys, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import golden_util as gu, gpu_util, ncde_amd
from ncde_amd import _lib, solver
import ncde_oracle as orc

name = sys.argv[1] if len(sys.argv) > 1 else "g3_cubic_rk4_seq"
case = gu.load_case(name); m = case["meta"]; ex = case["expect"]
r1 = gpu_util.run_case(case, flags=1); r2 = gpu_util.run_case(case, flags=1)
print("deterministic:", all(np.array_equal(r1["grads"][k], r2["grads"][k]) for k in r1["grads"] if r1["grads"][k] is not None), np.array_equal(r1["dz0"], r2["dz0"]))
# adjoint fed with the reference's z_out
dev = "cuda"
coeffs = torch.from_numpy(case["coeffs"]).to(dev)
func = gpu_util.CaseField(case["params"], case["layers"], dev)
spec = func.fused_spec()
def run_adj(z_out_np):
    z_out = torch.from_numpy(z_out_np).to(dev).contiguous()
    gout = torch.from_numpy(ex["grad_out"]).to(dev).contiguous()
    z0 = z_out[:, 0].contiguous()
    p = solver.build_problem(coeffs, m["kind"], z0, spec, m["method"], 1 if m["sequence"] else 0, 1)
    uniq = spec.unique_params(); gbuf = {id(q): torch.zeros_like(q) for q in uniq}
    g = _lib.NcdeGrads(); gz0 = torch.zeros_like(z0); g.grad_z0 = gz0.data_ptr()
    for i, (w, b) in enumerate(spec.layers):
        g.grad_layer_W[i], g.grad_layer_b[i] = gbuf[id(w)].data_ptr(), gbuf[id(b)].data_ptr()
    g.grad_Wo, g.grad_bo = gbuf[id(spec.Wo)].data_ptr(), gbuf[id(spec.bo)].data_ptr()
    ws = solver._workspace(p, 1, dev)
    _lib.check(_lib.lib().ncde_adjoint(ctypes.byref(p), z_out.data_ptr(), gout.data_ptr(), ctypes.byref(g), ws.data_ptr(), ws.numel(), None), "adj")
    torch.cuda.synchronize()
    return gz0.cpu().numpy(), {k: gbuf[id(v)].cpu().numpy() for k, v in func.p.items() if id(v) in gbuf}
gz0, gp = run_adj(ex["z_out"])
print("adjoint with REFERENCE z_out: dz0 err", gu.relerr(gz0, ex["dz0"]), {k: "%.1e" % gu.relerr(gp[k], ex["d"+k]) for k in gp if "d"+k in ex})
# oracle with GPU z_out
field = gu.oracle_field(case); ctl = orc.Control(case["coeffs"], m["kind"])
dz0o, gpo = orc.solve_adjoint(ctl, field, torch.from_numpy(r1["z_out"]), ex["grad_out"], m["method"], m["sequence"])
print("oracle adjoint fed with GPU z_out vs reference: dz0", gu.relerr(dz0o, ex["dz0"]), [ "%.1e" % gu.relerr(a, ex["d"+n]) for n, a in zip(m["param_names"], gpo)])
print("GPU vs oracle(fed GPU z_out): dz0", gu.relerr(r1["dz0"], dz0o), ["%.1e" % gu.relerr(r1["grads"][n], a) for n, a in zip(m["param_names"], gpo)])
gz0c, gpc = run_adj(r1["z_out"])
print("(c) direct adjoint with GPU z_out: dz0 err", gu.relerr(gz0c, ex["dz0"]), {k: "%.1e" % gu.relerr(gpc[k], ex["d"+k]) for k in gpc if "d"+k in ex})
d = np.abs(r1["dz0"] - ex["dz0"]).max(axis=1) / np.abs(ex["dz0"]).max()
print("per-sample dz0 err (autograd path):", np.array2string(d, precision=1))
d = np.abs(gz0c - ex["dz0"]).max(axis=1) / np.abs(ex["dz0"]).max()
print("per-sample dz0 err (direct, GPU z_out):", np.array2string(d, precision=1))
dz = np.abs(r1["z_out"] - ex["z_out"]).max(axis=(1,2))
print("per-sample z_out abs diff:", np.array2string(dz, precision=1))
