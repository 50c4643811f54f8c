"""Phase breakdown of one fused dopri5 ADJOINT attempt (workgroup 0, wall clock) from the -DNCDE_DPF_PROF build: the seven stages of the last
launch of the solve (the dense-output pass, same work as an attempt with one gradient sum).   usage: python tools/prof_dpa.py [variants/dpfprof.so]"""
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
z0 = m.initial_linear(coeffs[:, 0, :c["C"]]).detach().contiguous().requires_grad_(True)
out = ncde_amd.cdeint(X, m.func, z0, X.interval, adjoint=True, method="dopri5", rtol=1e-3, atol=1e-5, options={"min_step": 0.5})
out[:, -1].sum().backward()
torch.cuda.synchronize()
st = z0.grad[0, :7].cpu().numpy() * 0.01      # us
names = ["prologue (weights, state, first dX)", "forward recompute + images (7 stages)", "output tiles: P, tanh, dP, Wo^T dP, dWo", "cross-wave sum of dL/dx_L", "hidden backward + dW1 / dW0",
         "a^T df/dy, stage bookkeeping, exchange", "epilogue (state, partial sums, 2 x 95 KB of partials)"]
for k in range(7):
    print("%-62s %8.2f us" % (names[k], st[k]))
print("%-62s %8.2f us" % ("launch, workgroup 0", st.sum()))
