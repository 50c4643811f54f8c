"""Shape sweep of the kernel dispatch: for H x HH x C x nl over the reference's hyper-parameter ranges
(experiments/configurations/configurations.json5:34-36, src/ncde/ncde.py:42-61) time the forward and the adjoint kernels at B = 4096,
T = 99 (rectilinear, RK4) and compare each shape's time PER ALGORITHMIC FLOP with that of the aligned shape it is padded to.
Round 5: `frac_fwd` / `frac_adj` = the shape's own ALGORITHMIC fp32 flops (forward; 3 x forward for the adjoint)
per second over the 157.3 TFLOP/s fp32 MFMA / vector peak -- the absolute column beside the self-relative ratios -- and a wider grid
(H up to 256, C up to 100: the batch-tiled backward of round 5).
Run on the GPU box:  python tools/shape_sweep_perf.py [--quick] > profiles/r06_shape_sweep_perf.txt"""
import ctypes, itertools, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import ncde_amd, bench
from ncde_amd import _lib, solver

quick = "--quick" in sys.argv
Hs = [32, 47, 64, 96, 128, 160, 256]
HHs = [15, 32, 93, 128]
Cs = [4, 5, 10, 20, 21, 100]
NLs = [1, 2, 3, 4]
if quick:
    Hs, HHs, Cs, NLs = [32, 47], [15, 32], [5, 20], [1, 3]
B, L = 4096, 50
dev = torch.device("cuda", 0)
lib = _lib.lib()
ru = lambda x, m: (x + m - 1) // m * m
padw = lambda w: 16 if w <= 16 else 32 if w <= 32 else 64 if w <= 64 else 128 if w <= 128 else ru(w, 16)
cache = {}


def flops(C, H, HH, nl):
    return 4 * 2 * (H * HH + (nl - 1) * HH * HH + HH * H * C + H * C)


def measure(C, H, HH, nl):
    key = (C, H, HH, nl)
    if key in cache:
        return cache[key]
    c = dict(B=B, L=L, C=C, H=H, HH=HH, nl=nl, interpolation="rectilinear", solver="rk4", missing=0.3)
    coeffs = bench.make_inputs(c, B, 0, dev)
    model, fw, rw = bench.make_model(c, "cuda")
    ms_f, ms_a, names = bench.time_kernels(model, c, coeffs, iters=2)
    cache[key] = (ms_f, ms_a, names)
    del coeffs, model
    return cache[key]


print("B = %d, T = %d, rectilinear, RK4; ms per launch of the forward / adjoint kernels (HIP events); ratio = (time per algorithmic flop) / "
      "(time per flop of the aligned shape the library pads to)" % (B, 2 * L - 1))
print("frac_* = algorithmic fp32 TFLOP/s of the shape itself / 157.3")
print("%-22s %-22s %9s %9s %7s %7s %8s %8s  %s" % ("C,H,HH,nl", "padded to", "fwd ms", "adj ms", "r_fwd", "r_adj", "frac_fwd", "frac_adj", "kernels"))
lowest = (1.0, None)
worst = 0.0
for H, HH, C, nl in itertools.product(Hs, HHs, Cs, NLs):
    Cp, Hp, HHp = ru(C, 4), ru(H, 16), padw(HH)
    try:
        f, a, names = measure(C, H, HH, nl)
        fa, aa, _ = measure(Cp, Hp, HHp, nl)
    except Exception as e:      # noqa: BLE001
        print("%-22s FAILED: %s" % ("%d,%d,%d,%d" % (C, H, HH, nl), str(e)[:120]))
        continue
    fl, fla = flops(C, H, HH, nl), flops(Cp, Hp, HHp, nl)
    rf, ra = (f / fl) / (fa / fla), (a / fl) / (aa / fla)
    aligned = (Cp, Hp, HHp) == (C, H, HH)
    if not aligned:
        worst = max(worst, rf, ra)
    steps = B * (2 * L - 2)
    ff, fa_ = fl * steps / (f * 1e-3) / 157.3e12, 3 * fl * steps / (a * 1e-3) / 157.3e12
    if max(H, HH) <= 64 and min(ff, fa_) < lowest[0]:
        lowest = (min(ff, fa_), (C, H, HH, nl))
    print("%-22s %-22s %9.3f %9.3f %7.2f %7.2f %8.3f %8.3f  %s | %s" % ("%d,%d,%d,%d" % (C, H, HH, nl), "-" if aligned else "%d,%d,%d" % (Cp, Hp, HHp),
                                                                          f, a, rf, ra, ff, fa_, names[0], names[1]), flush=True)
    assert "generic" not in names[0] and "generic" not in names[1], names
print("worst per-flop ratio of a padded shape: %.2f" % worst)
print("lowest absolute fraction among shapes with H, HH <= 64: %.3f at (C, H, HH, nl) = %s" % lowest)
