"""One dopri5 training step at cfg2 dims under rocprofv3 --kernel-trace --stats (run: rocprofv3 ... -- python3 tools/prof_dopri5.py [adjoint|taped])."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, ncde_amd
c = dict(bench.CONFIGS["cfg2"])
B = int(os.environ.get("B", 4096))
coeffs = bench.make_inputs(c, B, 0, torch.device("cuda", 0))
y = (torch.rand(B, 1, device="cuda") > 0.5).float()
torch.manual_seed(0)
m = ncde_amd.NeuralCDE(c["C"], c["H"], 1, hidden_hidden_dim=c["HH"], num_layers=c["nl"], interpolation="rectilinear", solver="dopri5",
                       adjoint=(len(sys.argv) < 2 or sys.argv[1] != "taped")).cuda()
lf = torch.nn.BCEWithLogitsLoss()
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = m(coeffs); torch.cuda.synchronize(); t1 = time.perf_counter()
    lf(out, y).backward(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("forward %.1f ms  backward %.1f ms  nfe %s" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, getattr(m.func, "nfe", None)), flush=True)
