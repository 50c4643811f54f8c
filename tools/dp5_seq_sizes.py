"""dopri5 at cfg2 dims, fused attempt kernels vs the per-launch ones: final-time and sequence outputs (one adaptive reverse solve per output
interval: INIT0 / INIT1 / STEP / FIN launches), batch sizes 4096 / 1 / 17; forward and backward wall time.  Run on the GPU box."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench, ncde_amd
c = dict(bench.CONFIGS["cfg2"])
for B in (4096, 1, 17):
    coeffs = bench.make_inputs(c, B, 0, torch.device("cuda", 0))
    X = ncde_amd.LinearInterpolation(coeffs)
    torch.manual_seed(0)
    m = ncde_amd.NeuralCDE(c["C"], c["H"], 1, hidden_hidden_dim=c["HH"], num_layers=c["nl"], interpolation="rectilinear", solver="dopri5").cuda()
    z0v = m.initial_linear(coeffs[:, 0, :c["C"]]).detach().contiguous()
    for flags, lab in ((0, "fused"), (1, "per-launch")):
        for seq in (False, True):
            ts = []
            for it in range(2):
                z0 = z0v.clone().requires_grad_(True)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                out = ncde_amd.cdeint(X, m.func, z0, X.grid_points if seq else X.interval, adjoint=True, method="dopri5", rtol=1e-3, atol=1e-5, options={"min_step": 0.5}, kernel_flags=flags)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                out.square().sum().backward()
                torch.cuda.synchronize(); t2 = time.perf_counter()
                ts = [(t1 - t0) * 1e3, (t2 - t1) * 1e3]
            print("B %5d %-10s %-8s forward %7.1f ms  backward %7.1f ms  finite %s" % (B, lab, "seq" if seq else "final", ts[0], ts[1], bool(torch.isfinite(z0.grad).all())), flush=True)
