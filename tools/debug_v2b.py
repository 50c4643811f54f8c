import os, sys, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import golden_util as gu, gpu_util
import test_gpu_parity as T
from ncde_amd import _lib, solver
case = T._seeded_case("cubic", "midpoint", False, B=21, L=9, C=20, H=32, HH=32, nl=3, seed=120)
ex = case["expect"]; m = case["meta"]; dev = "cuda"
coeffs = torch.from_numpy(case["coeffs"]).to(dev)
func = gpu_util.CaseField(case["params"], case["layers"], dev); spec = func.fused_spec()
def run(flags):
    z_out = torch.from_numpy(np.ascontiguousarray(ex["z_out"])).to(dev); gout = torch.from_numpy(ex["grad_out"]).to(dev).contiguous()
    z0 = z_out[:, 0].contiguous()
    p = solver.build_problem(coeffs, "cubic", z0, spec, "midpoint", _lib.OUT_INTERVAL, flags)
    uniq = spec.unique_params(); gbuf = {id(q): torch.zeros_like(q) for q in uniq}
    g = _lib.NcdeGrads(); gz0 = torch.zeros_like(z0); g.grad_z0 = gz0.data_ptr()
    for i, (w, b) in enumerate(spec.layers):
        g.grad_layer_W[i], g.grad_layer_b[i] = gbuf[id(w)].data_ptr(), gbuf[id(b)].data_ptr()
    g.grad_Wo, g.grad_bo = gbuf[id(spec.Wo)].data_ptr(), gbuf[id(spec.bo)].data_ptr()
    nb = int(_lib.lib().ncde_workspace_bytes(ctypes.byref(p), 1))
    ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib().ncde_adjoint(ctypes.byref(p), z_out.data_ptr(), gout.data_ptr(), ctypes.byref(g), ws.data_ptr(), ws.numel(), None), "adj")
    torch.cuda.synchronize()
    theta = 32*32+32+32*32+32+640*32+640
    off = (2 * theta + 64) * 4
    nst = 8 * 2
    return ws[off: off + nst * 5 * 64 * 4].view(torch.float32).view(nst, 5, 64).cpu().numpy(), gz0.cpu().numpy()
d1, z1 = run(0x200)
d2, z2 = run(0x208)
names = ["kout", "vy", "gpre0", "ys", "as"]
for st in range(16):
    for q in range(5):
        if not np.array_equal(d1[st, q], d2[st, q]):
            bad = np.argwhere(d1[st, q] != d2[st, q]).ravel()
            print("stage", st, names[q], "differs in lanes", bad[:16].tolist(), "v1", d1[st, q][bad[:4]], "v2", d2[st, q][bad[:4]])
print("dz0 equal:", np.array_equal(z1, z2), np.abs(z1 - z2).max())
