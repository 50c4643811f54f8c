"""dopri5 on a g10 / g12 golden: fused attempt kernels (default) vs the per-launch kernels (kernel_flags=1): step sequences side by side.
usage: python tools/dbg_dp5.py [golden name]"""
import json, os, sys
sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/oracle"]
import numpy as np, torch
import golden_util as gu, gpu_util
import ncde_amd
name = sys.argv[1] if len(sys.argv) > 1 else "g10_ncde_dopri5_rect_final"
f = dict(np.load(os.path.join(gu.GOLD, name + ".npz")))
m = json.loads(str(f["meta"]))
coeffs = torch.from_numpy(f["coeffs"]).cuda()
X = (ncde_amd.LinearInterpolation if m["kind"] == "linear" else ncde_amd.NaturalCubicSpline)(coeffs)
params = {k[2:]: f[k] for k in f if k.startswith("p_")}
layers = [("W0", "b0"), ("W1", "b1")] if m["field"] == "toy" else [("W0", "b0")] + [("W1", "b1")] * (m["dims"]["nl"] - 1)
t = None
res = {}
for flags in (0, 1):
    func = gpu_util.CaseField(params, layers, "cuda")
    z0 = torch.from_numpy(f["z0"]).cuda().requires_grad_(True)
    t = X.grid_points if m["sequence"] else X.interval
    out = ncde_amd.cdeint(X, func, z0, t, adjoint=True, method="dopri5", rtol=m["rtol"], atol=m["atol"], options=dict(m["options"], _trace=4096), kernel_flags=flags)
    tr = func.dopri5_trace
    (out * torch.from_numpy(f["grad_out"]).cuda()).sum().backward()
    res[flags] = (out.detach().cpu().numpy(), tr, z0.grad.cpu().numpy(), func.dopri5_trace)
    print("flags", flags, "z err vs ref %.2e" % gu.relerr(res[flags][0], f["z_out"]), "dz0 err %.2e" % gu.relerr(res[flags][2], f["dz0"]), "attempts fwd", len(tr), "bwd", len(func.dopri5_trace), "dims", m.get("dims"))
a, b = res[0][1], res[1][1]
n = min(len(a), len(b))
for i in range(n):
    if not np.allclose(a[i], b[i], rtol=1e-3, atol=0):
        print("first forward attempt that differs:", i)
        for k in range(max(0, i - 2), min(n, i + 3)):
            print(k, "fused", a[k], " per-launch", b[k])
        break
else:
    print("forward step sequences agree to 1e-3 over", n, "attempts; max rel diff of ratio %.2e" % np.max(np.abs(a[:n, 3] - b[:n, 3]) / (np.abs(b[:n, 3]) + 1e-30)))
print("fused vs per-launch z diff %.2e" % gu.relerr(res[0][0], res[1][0]))
d = np.abs(res[0][0] - res[1][0])
print("per output row max |fused - per-launch|:", np.array2string(d.max(axis=(0, 2)) if d.ndim == 3 else d.max(axis=0), precision=2))
print("per sample:", np.array2string(d.max(axis=(1, 2)), precision=2))
print("per hidden unit:", np.array2string(d.max(axis=(0, 1)), precision=2))
dr = np.abs(a[:n] - b[:n]) / (np.abs(b[:n]) + 1e-12)
print("max rel diff per trace column over attempts with ratio > 1e-3:", dr[b[:n, 3] > 1e-3].max(axis=0))
for i in range(n):
    if abs(a[i, 1] - b[i, 1]) > 1e-4 * abs(b[i, 1]) or a[i, 2] != b[i, 2]:
        print("first attempt whose dt / decision differs:", i)
        for k in range(max(0, i - 4), min(n, i + 3)):
            print(k, "fused", a[k], " per-launch", b[k])
        break
ref = np.array(m["trace_fwd"], dtype=np.float64)
nn = min(n, len(ref))
print("attempt: reference dt | fused rel diff | per-launch rel diff   (accepted r/f/p)")
for i in range(min(nn, 28)):
    print("%3d  %.8f  %9.2e  %9.2e   %d/%d/%d   ratio fused %.4e per-launch %.4e" % (i, ref[i, 1], (a[i, 1] - ref[i, 1]) / ref[i, 1], (b[i, 1] - ref[i, 1]) / ref[i, 1], ref[i, 2], a[i, 2], b[i, 2], a[i, 3], b[i, 3]))
