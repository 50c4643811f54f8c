import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/oracle"]
import numpy as np, torch
import golden_util as gu, gpu_util
import ncde_oracle as orc
from ncde_amd import _lib
B, L, C, H, HH, nl, interp, method = 8192, 182, 4, 64, 64, 3, "cubic", "midpoint"
coeffs = gu.data.make_cubic_coeffs(B, L, C - 1, seed=1234)
p = gu.data.make_field_weights(H, HH, C, seed=0)
rw = gu.data.make_readin_weights(H, C, 1, seed=0)
z0 = (coeffs[:, 0, :C] @ rw["Wi"].T + rw["bi"]).astype(np.float32)
names = ["W0", "b0", "W1", "b1", "Wo", "bo"]
meta = {"kind": interp, "method": method, "sequence": False, "param_names": names, "field": "original", "dims": {"C": C, "H": H, "HH": HH, "nl": nl}}
gout = (gu.data.normal(3, B * 2 * H, stream=1).reshape(B, 2, H) / np.sqrt(2.0)).astype(np.float32)
big = {"meta": meta, "coeffs": coeffs, "z0": z0, "params": p, "layers": [("W0", "b0")] + [("W1", "b1")] * (nl - 1), "H": H, "C": C, "expect": {"grad_out": gout}}
torch.set_num_threads(16)
field, ctl = gu.oracle_field(big), orc.Control(coeffs, interp)
z = orc.solve_forward(ctl, field, z0, method, False)
dz0, gp = orc.solve_adjoint(ctl, field, z, gout, method, False)
# fp64 oracle
c64 = dict(big, params={k: v.astype(np.float64) for k, v in p.items()})
f64 = gu.oracle_field(c64); ctl64 = orc.Control(coeffs.astype(np.float64), interp)
z64 = orc.solve_forward(ctl64, f64, z0.astype(np.float64), method, False)
d64, g64 = orc.solve_adjoint(ctl64, f64, z64, gout.astype(np.float64), method, False)
print("fp32 oracle vs fp64:", {n: "%.1e" % gu.relerr(a.numpy(), b.numpy()) for n, a, b in zip(["dz0"] + names, [dz0] + list(gp), [d64] + list(g64))})
for fl, lab in ((0, "h64 fp16x2"), (_lib.FLAG_FP32_MFMA, "h64 fp32"), (_lib.FLAG_FORCE_TILED, "tiled"), (1, "generic")):
    iso = gpu_util.run_adjoint_direct(big, z.numpy(), flags=fl)
    e32 = {n: "%.1e" % gu.relerr(iso["grads"][n], g.numpy()) for n, g in zip(names, gp)}
    e64 = {n: "%.1e" % gu.relerr(iso["grads"][n], g.numpy()) for n, g in zip(names, g64)}
    per = np.abs(iso["dz0"] - dz0.numpy()).max(1) / np.abs(dz0.numpy()).max()
    print(lab, "vs fp32 oracle", e32, "| vs fp64", e64, "| dz0 rows > 1e-5:", int((per > 1e-5).sum()), "max %.1e" % per.max())
