"""Training-step time of the vector-field variants at cfg2 dims (run on the GPU box)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, ncde_amd
c = dict(bench.CONFIGS["cfg2"])
B = int(os.environ.get("B", 4096))
coeffs = bench.make_inputs(c, B, 0, torch.device("cuda", 0))
y = (torch.rand(B, 1, device="cuda") > 0.5).float()
ALL = [(a, b) for a in ("original", "minimal", "gru") for b in ("matmul", "evaluate", "derivative")]
for vf, vft in (ALL if os.environ.get("ALL") else (("original", "matmul"), ("minimal", "matmul"), ("gru", "matmul"), ("original", "evaluate"), ("gru", "derivative"))):
    torch.manual_seed(0)
    m = ncde_amd.NeuralCDE(c["C"], c["H"], 1, hidden_hidden_dim=c["HH"], num_layers=c["nl"], interpolation="rectilinear",
                           vector_field=vf, vector_field_type=vft, adjoint=True, solver="rk4").cuda()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    lf = torch.nn.BCEWithLogitsLoss()
    def step():
        opt.zero_grad(set_to_none=True); l = lf(m(coeffs), y); l.backward(); opt.step(); return l
    step(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): l = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print("%-9s %-10s %8.1f ms/step  %.3e sample-steps/s  loss %.4f" % (vf, vft, dt * 1e3, B * 398 / dt, float(l)), flush=True)
