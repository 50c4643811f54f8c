"""cfg5 full-length adjoint vs the oracle on a 32-sample subset: fp32-MFMA path against the split-bf16 path (run on the GPU box)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import golden_util as gu, gpu_util, ncde_oracle as orc
from ncde_amd import _lib
B, L, C, H, HH, nl = 64, 400, 80, 128, 128, 3
coeffs = gu.data.make_rectilinear_coeffs(B, L, C - 1, missing=0.6, seed=1234)
p = gu.data.make_field_weights(H, HH, C, seed=0)
rw = gu.data.make_readin_weights(H, C, 1, seed=0)
z0 = (coeffs[:, 0] @ rw["Wi"].T + rw["bi"]).astype(np.float32)
names = ["W0", "b0", "W1", "b1", "Wo", "bo"]
meta = {"kind": "linear", "method": "rk4", "sequence": False, "param_names": names, "field": "original", "dims": {"C": C, "H": H, "HH": HH, "nl": nl}}
gout = (gu.data.normal(3, B * 2 * H, stream=1).reshape(B, 2, H) / np.sqrt(2.0)).astype(np.float32)
case = {"meta": meta, "coeffs": coeffs, "z0": z0, "params": p, "layers": [("W0", "b0")] + [("W1", "b1")] * (nl - 1), "H": H, "C": C, "expect": {"grad_out": gout}}
field = gu.oracle_field(case)
ctl = orc.Control(coeffs, "linear")
torch.set_num_threads(16)
z = orc.solve_forward(ctl, field, z0, "rk4", False)
dz0, gp = orc.solve_adjoint(ctl, field, z, gout, "rk4", False)
# fp64 run of the same discrete scheme, to size the fp32 drift itself
c64 = dict(case, params={k: v.astype(np.float64) for k, v in p.items()})
f64 = gu.oracle_field(c64)
z64 = orc.solve_forward(orc.Control(coeffs.astype(np.float64), "linear"), f64, z0.astype(np.float64), "rk4", False)
dz64, gp64 = orc.solve_adjoint(orc.Control(coeffs.astype(np.float64), "linear"), f64, z64, gout.astype(np.float64), "rk4", False)
print("fp32 oracle vs fp64: z %.2e dz0 %.2e" % (gu.relerr(z.numpy(), z64.numpy()), gu.relerr(dz0.numpy(), dz64.numpy())),
      {n: "%.2e" % gu.relerr(g.numpy(), g64.numpy()) for n, g, g64 in zip(names, gp, gp64)})
for flags, label in ((_lib.FLAG_FORCE_TILED if hasattr(_lib, "FLAG_FORCE_TILED") else 0x8000, "default (split-bf16)"), (0x8000 | 4, "fp32-input MFMA")):
    iso = gpu_util.run_adjoint_direct(case, z.numpy(), flags=flags)
    print(label, "vs fp32 oracle: dz0 %.2e" % gu.relerr(iso["dz0"], dz0), {n: "%.2e" % gu.relerr(iso["grads"][n], g) for n, g in zip(names, gp)})
    print(label, "vs fp64       : dz0 %.2e" % gu.relerr(iso["dz0"], dz64.numpy()), {n: "%.2e" % gu.relerr(iso["grads"][n], g.numpy()) for n, g in zip(names, gp64)})
