import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import golden_util as gu, gpu_util, ncde_oracle as orc
kind, interp, method, step = "original", sys.argv[2], sys.argv[3], float(sys.argv[1])
B, L, C, H, HH, nl = [int(v) for v in sys.argv[4].split(",")]
x = (gu.data.normal(41, B * L * C, stream=3).reshape(B, L, C) * 0.5).astype(np.float32)
kn = np.arange(L, dtype=np.float32); x[:, :, 0] = kn[None, :]
coeffs = gu.data.natural_cubic_coeffs(x) if interp == "cubic" else x
p = gu.data.make_field_weights(H, HH, C, seed=19)
z0 = (gu.data.normal(43, B * H, stream=2).reshape(B, H) * 0.5).astype(np.float32)
for tout in (np.array([kn[0], kn[4], kn[-1]], np.float32),):
    meta = {"kind": interp, "method": method, "step_size": step, "dims": {"nl": nl}}
    field = orc.Field.variant(p, H, C, nl, kind, "matmul")
    ctl = orc.Control(coeffs, interp, t=None)
    z = orc.solve_forward_times(ctl, field, z0, tout, method, step)
    gout = (gu.data.normal(23, z.numel(), stream=1).reshape(z.shape) / 2.0).astype(np.float32)
    dz0, gp = orc.solve_adjoint_times(ctl, field, tout, z, gout, method, step)
    bdz0, bgp = orc.solve_discrete_backward_times(ctl, field, z0, tout, gout, method, step)
    g = {"coeffs": coeffs, "z0": z0, "t_out": tout, "grad_out": gout}
    names = ["W0", "b0", "W1", "b1", "Wo", "bo"]
    for flags, label in ((1, "generic"),):
        r = gpu_util.run_times_case(g, meta, adjoint=True, flags=flags, params=p)
        rd = gpu_util.run_times_case(g, meta, adjoint=False, flags=flags, params=p)
        print(label, "tout", tout.tolist(), "z %.1e adj dz0 %.1e" % (gu.relerr(r["z_out"], z), gu.relerr(r["dz0"], dz0)), {n: "%.1e" % gu.relerr(r["grads"][n], gg) for n, gg in zip(names, gp)},
              "| disc dz0 %.1e" % gu.relerr(rd["dz0"], bdz0), {n: "%.1e" % gu.relerr(rd["grads"][n], gg) for n, gg in zip(names, bgp)})
    ps = np.abs(r["dz0"] - dz0.numpy()).max(axis=1) / np.abs(dz0.numpy()).max()
    print("per-sample dz0 error:", np.array2string(ps, precision=1, max_line_width=200))
    print("oracle adjoint vs oracle discrete: dz0 %.1e" % gu.relerr(dz0.numpy(), bdz0.numpy()))
