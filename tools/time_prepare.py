"""HBM throughput of the coefficient builders at BASELINE sizes (run on the GPU box)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ncde_amd
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n, out
x = torch.from_numpy(ncde_amd.data.synthetic_series(4096, 200, 19, missing=0.3, seed=1234)).cuda()
dt, out = timeit(lambda: ncde_amd.linear_interpolation_coeffs(x, rectilinear=0))
print("rectilinear cfg2: %.1f us, %.0f GB/s (in %.0f MB + out %.0f MB)" % (dt * 1e6, (x.numel() + out.numel()) * 4 / dt / 1e9, x.numel() * 4 / 1e6, out.numel() * 4 / 1e6))
dt, out = timeit(lambda: ncde_amd.linear_interpolation_coeffs(x))
print("linear NaN fill cfg2: %.1f us, %.0f GB/s" % (dt * 1e6, (x.numel() + out.numel()) * 4 / dt / 1e9))
xc = torch.from_numpy(ncde_amd.data.synthetic_series(8192, 182, 3, missing=0.0, seed=1234)).cuda()
dt, out = timeit(lambda: ncde_amd.natural_cubic_coeffs(xc))
print("natural cubic cfg4: %.1f us, %.0f GB/s (in %.0f MB + out %.0f MB)" % (dt * 1e6, (xc.numel() + out.numel()) * 4 / dt / 1e9, xc.numel() * 4 / 1e6, out.numel() * 4 / 1e6))
xm = torch.from_numpy(ncde_amd.data.synthetic_series(8192, 182, 3, missing=0.3, seed=1234)).cuda()
dt, out = timeit(lambda: ncde_amd.natural_cubic_coeffs(xm))
print("natural cubic with 30%% missing cfg4: %.1f us, %.0f GB/s" % (dt * 1e6, (xm.numel() + out.numel()) * 4 / dt / 1e9))
x5 = torch.from_numpy(ncde_amd.data.synthetic_series(4096, 400, 79, missing=0.6, seed=1234)).cuda()
dt, out = timeit(lambda: ncde_amd.linear_interpolation_coeffs(x5, rectilinear=0), 5)
print("rectilinear cfg5: %.1f us, %.0f GB/s (in %.0f MB + out %.0f MB)" % (dt * 1e6, (x5.numel() + out.numel()) * 4 / dt / 1e9, x5.numel() * 4 / 1e6, out.numel() * 4 / 1e6))
