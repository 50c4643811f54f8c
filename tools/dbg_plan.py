import sys
sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/oracle"]
import numpy as np, torch
import golden_util as gu, gpu_util
import ncde_oracle as orc
from ncde_amd import _lib
C, H, HH, nl = [int(v) for v in sys.argv[1].split(",")]
interp, method, step = sys.argv[2], sys.argv[3], float(sys.argv[4])
B, L = 37, 9
rng = np.random.RandomState(7)
x = (gu.data.normal(61, B * L * C, stream=3).reshape(B, L, C) * 0.5).astype(np.float32)
if interp == "linear":
    kn = np.cumsum(np.concatenate([[0.0], 0.6 + 0.8 * rng.rand(L - 1)])).astype(np.float32); x[:, :, 0] = kn[None, :]; coeffs = x
else:
    kn = np.arange(L, dtype=np.float32); x[:, :, 0] = kn[None, :]; coeffs = gu.data.natural_cubic_coeffs(x)
p = gu.data.make_field_weights(H, HH, C, seed=29)
z0 = (gu.data.normal(63, B * H, stream=2).reshape(B, H) * 0.5).astype(np.float32)
tout = np.array([kn[0], 0.5 * (kn[1] + kn[2]), kn[4], kn[6] + 0.05, kn[-1] - 0.125], np.float32)
meta = {"kind": interp, "method": method, "step_size": step, "dims": {"nl": nl}}
field = orc.Field.variant(p, H, C, nl, "original", "matmul")
ctl = orc.Control(coeffs, interp, t=kn if interp == "linear" else None)
z = orc.solve_forward_times(ctl, field, z0, tout, method, step)
gout = (gu.data.normal(25, z.numel(), stream=1).reshape(z.shape) / 2.0).astype(np.float32)
dz0, gp = orc.solve_adjoint_times(ctl, field, tout, z, gout, method, step)
g = {"coeffs": coeffs, "z0": z0, "t_out": tout, "grad_out": gout}
if interp == "linear": g["knots"] = kn
names = [n for n in ("W0", "b0", "W1", "b1", "Wo", "bo") if n in p]
for flags, lab in ((0, "auto"), (64, "split-bf16"), (0x8000, "tiled"), (1, "generic")):
    res = gpu_util.run_times_case(g, meta, adjoint=True, params=p, flags=flags)
    per = np.abs(res["dz0"] - dz0.numpy()).max(1) / np.abs(dz0.numpy()).max()
    print(lab, "z %.1e" % gu.relerr(res["z_out"], z), "dz0 %.1e" % gu.relerr(res["dz0"], dz0), "rows>2e-4:", np.where(per > 2e-4)[0], np.sort(per)[-3:], {n: "%.1e" % gu.relerr(res["grads"][n], g_) for n, g_ in zip(names, gp)})
z0t = z0
bdz0, bgp = orc.solve_discrete_backward_times(ctl, field, z0t, tout, gout, method, step)
for flags, lab in ((0, "auto"), (64, "split-bf16"), (0x8000, "tiled"), (1, "generic")):
    try:
        res = gpu_util.run_times_case(g, meta, adjoint=False, params=p, flags=flags)
    except Exception as e:
        print("discrete", lab, "->", str(e)[:100]); continue
    per = np.abs(res["dz0"] - bdz0.numpy()).max(1) / np.abs(bdz0.numpy()).max()
    print("discrete", lab, "z %.1e" % gu.relerr(res["z_out"], z), "dz0 %.1e" % gu.relerr(res["dz0"], bdz0), "rows>2e-4:", np.where(per > 2e-4)[0], np.sort(per)[-3:], {n: "%.1e" % gu.relerr(res["grads"][n], g_) for n, g_ in zip(names, bgp)})
