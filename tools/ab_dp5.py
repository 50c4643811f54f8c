"""A/B of dopri5 training-step time between library builds on ONE box: python tools/ab_dp5.py variants/a.so [variants/b.so ...]
(the shipped library first).  Same seeds, same data: the step sequences are identical wherever the builds compute the same numbers."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, torch
sys.path.insert(0, %r)
from ncde_amd import _lib
if sys.argv[1] != "-": _lib.LIB_PATH = os.path.join(%r, sys.argv[1])
import bench, ncde_amd
c = dict(bench.CONFIGS["cfg2"])
coeffs = bench.make_inputs(c, 4096, 0, torch.device("cuda", 0))
torch.manual_seed(1)
y = (torch.rand(4096, 1, device="cuda") > 0.5).float()
for adjoint in (True, False):
    torch.manual_seed(0)
    m = ncde_amd.NeuralCDE(c["C"], c["H"], 1, hidden_hidden_dim=c["HH"], num_layers=c["nl"], interpolation="rectilinear", solver="dopri5", adjoint=adjoint).cuda()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    lf = torch.nn.BCEWithLogitsLoss()
    def step():
        opt.zero_grad(set_to_none=True); l = lf(m(coeffs), y); l.backward(); opt.step(); return l
    step(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): step()
    torch.cuda.synchronize()
    print("%%-22s adjoint=%%-5s %%7.1f ms/step  nfe %%d" %% (sys.argv[1], adjoint, (time.perf_counter() - t0) / 4 * 1e3, m.func.nfe), flush=True)
''' % (ROOT, ROOT)
for lib in ["-"] + sys.argv[1:]:
    subprocess.run([sys.executable, "-c", CHILD, lib], check=False)
