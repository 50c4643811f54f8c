"""Phase breakdown of one fused dopri5 attempt (workgroup 0, wall clock, 10 ns ticks) from the -DNCDE_DPF_PROF build.
usage: python tools/prof_dpf.py [variants/dpfprof.so]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ncde_amd, bench
from ncde_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "variants/dpfprof.so")
c = dict(bench.CONFIGS["cfg2"])
coeffs = bench.make_inputs(c, 4096, 0, torch.device("cuda", 0))
torch.manual_seed(0)
m = ncde_amd.NeuralCDE(c["C"], c["H"], 1, hidden_hidden_dim=c["HH"], num_layers=c["nl"], interpolation="rectilinear", solver="dopri5").cuda()
X = ncde_amd.LinearInterpolation(coeffs)
with torch.no_grad():
    z0 = m.initial_linear(coeffs[:, 0, :c["C"]]).contiguous()
    for _ in range(2):
        out = ncde_amd.cdeint(X, m.func, z0, X.interval, adjoint=False, method="dopri5", rtol=1e-3, atol=1e-5, options={"min_step": 0.5})
torch.cuda.synchronize()
st = out[0, 0, :8].cpu().numpy() * 0.01      # us
names = ["weights -> registers", "dX staging + state load", "6 stages", "error / outputs / stores", "partial sums", "__threadfence", "ticket (+ controller if last)"]
for k in range(7):
    print("%-42s %7.2f us" % (names[k], st[k + 1] - st[k]))
print("%-42s %7.2f us" % ("attempt, workgroup 0", st[7]))
