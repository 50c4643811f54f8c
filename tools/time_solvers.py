"""Training-step time of the non-default solver settings at cfg2 dims (run on the GPU box):
rk4 on the default grid (specialised kernels), rk4 with step_size 0.5 (time plan, generic family), dopri5 (adaptive kernels)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, ncde_amd
c = dict(bench.CONFIGS["cfg2"])
B = int(os.environ.get("B", 4096))
coeffs = bench.make_inputs(c, B, 0, torch.device("cuda", 0))
torch.manual_seed(1)      # (labels seeded: an adaptive solve's number of attempts depends on the data, and so does its time)
y = (torch.rand(B, 1, device="cuda") > 0.5).float()
for label, kw, opts in (("rk4 default grid", dict(solver="rk4"), None), ("rk4 step_size 0.5", dict(solver="rk4"), {"step_size": 0.5}),
                        ("dopri5 (min_step 0.5)", dict(solver="dopri5"), None),
                        ("dopri5, adjoint=False", dict(solver="dopri5", adjoint=False), None)):
    torch.manual_seed(0)
    try:
        m = ncde_amd.NeuralCDE(c["C"], c["H"], 1, hidden_hidden_dim=c["HH"], num_layers=c["nl"], interpolation="rectilinear",
                               **dict(dict(adjoint=True), **kw)).cuda()
    except TypeError as e:
        print(label, "not constructible:", e); continue
    if opts is not None:
        m.cdeint_options = opts
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    lf = torch.nn.BCEWithLogitsLoss()
    def step():
        opt.zero_grad(set_to_none=True); l = lf(m(coeffs), y); l.backward(); opt.step(); return l
    step(); torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 2
    for _ in range(n): l = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    nfe = getattr(m.func, "nfe", None)
    print("%-24s %9.1f ms/step   loss %.4f   nfe %s" % (label, dt * 1e3, float(l.detach()), nfe), flush=True)
