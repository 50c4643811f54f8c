"""fused vs per-launch dopri5 (replayed step sequences) at cfg2 dims for several batch sizes / lengths: where do they part?"""
import sys
sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/oracle"]
import numpy as np, torch
import golden_util as gu
import ncde_amd, bench
c = dict(bench.CONFIGS["cfg2"])
for B, L in ((64, 20), (256, 20), (4096, 20), (256, 200), (4096, 200)):
    cc = dict(c, L=L)
    coeffs = bench.make_inputs(cc, B, 0, torch.device("cuda", 0))
    torch.manual_seed(0)
    m = ncde_amd.NeuralCDE(c["C"], c["H"], 1, hidden_hidden_dim=c["HH"], num_layers=c["nl"], interpolation="rectilinear", solver="dopri5").cuda()
    X = ncde_amd.LinearInterpolation(coeffs)
    func = m.func
    with torch.no_grad():
        z0v = m.initial_linear(coeffs[:, 0, :c["C"]]).contiguous()
    gout = torch.randn(B, 2, c["H"], device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) / B
    kw = dict(method="dopri5", rtol=1e-3, atol=1e-5)
    def run(flags, options, aopt=None):
        for q in func.parameters(): q.grad = None
        z0 = z0v.clone().requires_grad_(True)
        out = ncde_amd.cdeint(X, func, z0, X.interval, adjoint=True, options=dict(options), adjoint_options=aopt, kernel_flags=flags, **kw)
        trf = func.dopri5_trace.copy() if getattr(func, "dopri5_trace", None) is not None else None
        (out * gout).sum().backward()
        trb = func.dopri5_trace_backward.copy() if getattr(func, "dopri5_trace_backward", None) is not None else None
        return out.detach().clone(), z0.grad.clone(), [q.grad.clone() for q in func.parameters()], trf, trb
    ref = run(1, {"min_step": 0.5, "_trace": 4096})
    got = run(0, {"min_step": 0.5, "_replay": ref[3][:, 1:3], "_trace": 4096}, {"min_step": 0.5, "_replay": ref[4][:, 1:3], "_trace": 4096})
    e = lambda a, b: float((a - b).abs().max() / b.abs().max())
    nb = min(len(ref[4]), len(got[4]))
    dtr = np.abs(ref[4][:nb, :3] - got[4][:nb, :3]).max()
    print("B %5d L %3d: attempts fwd %d bwd %d/%d (trace diff %.1e; max ratio rel diff %.1e)  z %.1e dz0 %.1e params %s" % (
        B, L, len(ref[3]), len(ref[4]), len(got[4]), dtr, np.max(np.abs(ref[4][:nb, 3] - got[4][:nb, 3]) / (np.abs(ref[4][:nb, 3]) + 1e-12)),
        e(got[0], ref[0]), e(got[1], ref[1]), ["%.1e" % e(a, b) for a, b in zip(got[2], ref[2])]), flush=True)
