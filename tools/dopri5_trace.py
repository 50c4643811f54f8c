import sys, os, json
sys.path[:0]=['/root/repo','/root/repo/oracle','/root/repo/tests']
import numpy as np, torch
import ncde_amd, gpu_util, golden_util as gu
for name in sys.argv[1:]:
    f = dict(np.load(os.path.join(gu.GOLD, name + ".npz"))); m = json.loads(str(f["meta"]))
    coeffs = torch.from_numpy(f["coeffs"]).cuda()
    X = (ncde_amd.LinearInterpolation if m["kind"] == "linear" else ncde_amd.NaturalCubicSpline)(coeffs)
    params = {k[2:]: f[k] for k in f if k.startswith("p_")}
    layers = [("W0", "b0"), ("W1", "b1")] if m["field"] == "toy" else [("W0", "b0")] + [("W1", "b1")] * (m["dims"]["nl"] - 1)
    func = gpu_util.CaseField(params, layers, "cuda")
    z0 = torch.from_numpy(f["z0"]).cuda()
    t = X.grid_points if m["sequence"] else X.interval
    opts = dict(m["options"]); opts["_trace"] = 4096
    with torch.no_grad():
        out = ncde_amd.cdeint(X, func, z0, t, adjoint=True, method="dopri5", rtol=m["rtol"], atol=m["atol"], options=opts)
    tr = func.dopri5_trace; ref = m["trace_fwd"]
    print(name, len(tr), len(ref), 'z err', gu.relerr(out.cpu().numpy(), f["z_out"]))
    for i in range(min(len(tr), len(ref))):
        a, b = tr[i], ref[i]
        bad = abs(a[0]-b[0]) > 1e-9*max(1,abs(b[0])) or abs(a[1]-b[1]) > 1e-6*abs(b[1]) or int(a[2]) != b[2]
        if i < 6 or bad:
            print(i, a, b, '<<<' if bad else '')
        if bad: break
