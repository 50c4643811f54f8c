"""Golden vectors for the UNFUSED path of cdeint (arbitrary func, decreasing t, gradients wrt the control path / the output times):
the imported reference (torchcde.cdeint + torchdiffeq, this container only) on small seeded problems with the reference tests' own
inline field (modules/torchcde/test/test_tricks.py:6-18: z.sigmoid().unsqueeze(-1) + variable).
    python oracle/gen_golden_unfused.py   ->  tests/golden/g13_unfused.npz
Test infrastructure: nothing in the product path imports this."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = ["/root/reference/modules/torchdiffeq", "/root/reference/modules/torchcde"]
import torchcde  # noqa: E402


class RefFunc(torch.nn.Module):      # test_tricks.py:6-18 (without its batch-size-1 assertions)
    def __init__(self, variable):
        super().__init__()
        self.variable = torch.nn.Parameter(variable)

    def forward(self, t, z):
        return z.sigmoid().unsqueeze(-1) + self.variable


CASES = [  # name, interp, knots on a user grid?, output times, method, step_size, adjoint
    ("tricks_rk4_adj", "cubic", True, [0.0, 9.0], "rk4", None, True),            # test_grad_paths: one step over the whole interval
    ("tricks_rk4_tape", "cubic", True, [0.0, 9.0], "rk4", None, False),
    ("detach_rk4_half", "cubic", False, [0.0, 9.0], "rk4", 0.5, True),            # test_detach_trick
    ("decreasing_midpoint", "linear", False, [9.0, 6.5, 2.25, 0.0], "midpoint", 0.75, True),
    ("decreasing_euler_tape", "linear", False, [8.0, 3.0, 1.0], "euler", 0.4, False),
    ("interior_rk4_adj", "linear", False, [0.5, 2.0, 7.3, 9.0], "rk4", 1.0, True),
    # round 5: the adaptive solver's continuous adjoint also returns dL/dt (adjoint.py:112-136 computes time_vjps for
    # every solver).  Forced step sequence (first_step = min_step = max_step) so that both sides walk the same steps.
    ("dopri5_adj_tgrad", "cubic", False, [0.0, 2.5, 9.0], "dopri5", None, True),
]
DOPRI5_OPTS = {"first_step": 0.5, "min_step": 0.5, "max_step": 0.5}

out = {}
g = torch.Generator().manual_seed(20261002)
for name, interp, user_grid, tt, method, step, adjoint in CASES:
    B, L, C, H = 3, 10, 3, 4
    x = torch.rand(B, L, C, generator=g)
    knots = torch.linspace(0, 9, L)
    if user_grid:
        knots = knots + 0.3 * torch.rand(L, generator=g) * (knots > 0) * (knots < 9)
    build = torchcde.natural_cubic_coeffs if interp == "cubic" else torchcde.linear_interpolation_coeffs
    coeffs = build(x, knots).detach().clone().requires_grad_(True)
    kn = knots.clone().requires_grad_(True)
    X = (torchcde.NaturalCubicSpline if interp == "cubic" else torchcde.LinearInterpolation)(coeffs, kn)
    f = RefFunc(torch.rand(1, 1, C, generator=g))
    z0 = torch.rand(B, H, generator=g).requires_grad_(True)
    t = torch.tensor(tt, requires_grad=True)
    kw = {"adjoint_params": tuple(f.parameters()) + (coeffs, kn)} if adjoint else {}
    opts = {} if step is None else {"step_size": step}
    if method == "dopri5":
        opts = dict(DOPRI5_OPTS)
        kw.update(rtol=1e-3, atol=1e-5)
    z = torchcde.cdeint(X, f, z0, t, adjoint=adjoint, method=method, options=opts, **kw)
    w = torch.rand(z.shape, generator=g) - 0.5
    (z * w).sum().backward()
    for k, v in (("x", x), ("knots", knots), ("coeffs", coeffs), ("variable", f.variable), ("z0", z0), ("t", t), ("w", w), ("z", z),
                 ("d_z0", z0.grad), ("d_variable", f.variable.grad), ("d_coeffs", coeffs.grad), ("d_t", t.grad), ("d_knots", kn.grad)):
        out[name + "__" + k] = v.detach().numpy().copy()
    out[name + "__meta"] = np.array([interp, method, "" if step is None else repr(step), str(int(adjoint))])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g13_unfused.npz"), **out)
print("wrote g13_unfused.npz:", len(CASES), "cases")
