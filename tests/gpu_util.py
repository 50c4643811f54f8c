"""Helpers for the GPU parity tests: run a golden/oracle case through the product path (cdeint -> C-ABI)."""
import numpy as np
import torch

import ncde_amd
from ncde_amd import _lib


class CaseField(torch.nn.Module):
    """Vector field built from a case's parameter dict; repeated (W, b) names share one Parameter."""

    def __init__(self, params, layers, device):
        super().__init__()
        self.p = torch.nn.ParameterDict({k: torch.nn.Parameter(torch.from_numpy(np.ascontiguousarray(v)).to(device))
                                         for k, v in params.items()})
        self.layer_names = layers
        self.nfe = 0

    def fused_spec(self):
        return ncde_amd.FieldSpec([(self.p[w], self.p[b]) for w, b in self.layer_names], self.p["Wo"], self.p["bo"])


def run_case(case, flags=_lib.FLAG_AUTO, device="cuda", need_grads=True):
    """-> dict(z_out, dz0, grads{name: array}) computed by the HIP path."""
    m = case["meta"]
    coeffs = torch.from_numpy(case["coeffs"]).to(device)
    X = (ncde_amd.LinearInterpolation if m["kind"] == "linear" else ncde_amd.NaturalCubicSpline)(coeffs)
    func = CaseField(case["params"], case["layers"], device)
    z0 = torch.from_numpy(case["z0"]).to(device).requires_grad_(True)
    t = X.grid_points if m["sequence"] else X.interval
    out = ncde_amd.cdeint(X, func, z0, t, adjoint=True, method=m["method"], options={"step_size": 1},
                          kernel_flags=flags)
    res = {"z_out": out.detach().cpu().numpy(), "nfe_fwd": func.nfe}
    if need_grads:
        gout = torch.from_numpy(case["expect"]["grad_out"]).to(device)
        (out * gout).sum().backward()
        res["dz0"] = z0.grad.cpu().numpy()
        res["grads"] = {k: v.grad.cpu().numpy() for k, v in func.p.items()}
        res["nfe"] = func.nfe
    torch.cuda.synchronize()
    return res
