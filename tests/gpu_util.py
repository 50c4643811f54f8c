"""Helpers for the GPU parity tests: run a golden/oracle case through the product path (cdeint -> C-ABI)."""
import numpy as np
import torch

import ncde_amd
from ncde_amd import _lib


class CaseField(torch.nn.Module):
    """Vector field built from a case's parameter dict; repeated (W, b) names share one Parameter."""

    def __init__(self, params, layers, device, kind="original", mode="matmul"):
        super().__init__()
        self.p = torch.nn.ParameterDict({k: torch.nn.Parameter(torch.from_numpy(np.ascontiguousarray(v)).to(device))
                                         for k, v in params.items()})
        self.layer_names = layers
        self.kind, self.mode = kind, mode
        self.nfe = 0

    def fused_spec(self):
        g = lambda k: self.p[k] if k in self.p else None   # noqa: E731
        return ncde_amd.FieldSpec([(self.p[w], self.p[b]) for w, b in self.layer_names], self.p["Wo"], self.p["bo"],
                                  self.kind, self.mode, g("Wg"), g("bg"), g("Wr"), g("br"))


def case_field(case, device):
    m = case["meta"]
    return CaseField(case["params"], case["layers"], device, m.get("field_kind", "original"), m.get("field_mode", "matmul"))


def run_case(case, flags=_lib.FLAG_AUTO, device="cuda", need_grads=True, adjoint=True):
    """-> dict(z_out, dz0, grads{name: array}) computed by the HIP path (adjoint=False: exact discrete backward)."""
    m = case["meta"]
    coeffs = torch.from_numpy(case["coeffs"]).to(device)
    X = (ncde_amd.LinearInterpolation if m["kind"] == "linear" else ncde_amd.NaturalCubicSpline)(coeffs)
    func = case_field(case, device)
    z0 = torch.from_numpy(case["z0"]).to(device).requires_grad_(True)
    t = X.grid_points if m["sequence"] else X.interval
    out = ncde_amd.cdeint(X, func, z0, t, adjoint=adjoint, vector_field_type=func.mode, method=m["method"],
                          options={"step_size": 1}, kernel_flags=flags)
    res = {"z_out": out.detach().cpu().numpy(), "nfe_fwd": func.nfe, "kernels": kernel_names(case, flags, device)}
    if need_grads:
        gout = torch.from_numpy(case["expect"]["grad_out"]).to(device)
        (out * gout).sum().backward()
        res["dz0"] = z0.grad.cpu().numpy()
        res["grads"] = {k: (v.grad.cpu().numpy() if v.grad is not None else None) for k, v in func.p.items()}
        res["nfe"] = func.nfe
    torch.cuda.synchronize()
    return res


def run_case_async(case, flags=_lib.FLAG_AUTO, device="cuda"):
    """The forward solve of `case` enqueued on the CURRENT stream; returns the device tensor without synchronising."""
    m = case["meta"]
    coeffs = torch.from_numpy(case["coeffs"]).to(device)
    X = (ncde_amd.LinearInterpolation if m["kind"] == "linear" else ncde_amd.NaturalCubicSpline)(coeffs)
    func = case_field(case, device)
    z0 = torch.from_numpy(case["z0"]).to(device)
    with torch.no_grad():
        return ncde_amd.cdeint(X, func, z0, X.grid_points if m["sequence"] else X.interval, vector_field_type=func.mode,
                               method=m["method"], options={"step_size": 1}, kernel_flags=flags)


def run_adjoint_direct(case, z_out, flags=_lib.FLAG_AUTO, device="cuda", stages=None):
    """Call ncde_adjoint through the C-ABI on a GIVEN forward solution (e.g. the reference's own z_out):
    isolates the adjoint kernel from forward round-off (a last-bit change of z can flip a ReLU mask).
    With `stages` (a stage record [(T-1)*S, B, H]) it calls ncde_backward (exact discrete backward) instead."""
    import ctypes
    from ncde_amd import solver
    m = case["meta"]
    coeffs = torch.from_numpy(case["coeffs"]).to(device)
    func = case_field(case, device)
    spec = func.fused_spec()
    z_out = torch.from_numpy(np.ascontiguousarray(z_out)).to(device)
    gout = torch.from_numpy(case["expect"]["grad_out"]).to(device).contiguous()
    z0 = z_out[:, 0].contiguous()
    p = solver.build_problem(coeffs, m["kind"], z0, spec, m["method"],
                             _lib.OUT_KNOTS if m["sequence"] else _lib.OUT_INTERVAL, flags)
    uniq = spec.unique_params()
    gbuf = {id(q): torch.full_like(q, float("nan")) for q in uniq}
    g = _lib.NcdeGrads()
    gz0 = torch.full_like(z0, float("nan"))
    g.grad_z0 = gz0.data_ptr()
    for i, (w, b) in enumerate(spec.layers):
        g.grad_layer_W[i], g.grad_layer_b[i] = gbuf[id(w)].data_ptr(), gbuf[id(b)].data_ptr()
    g.grad_Wo, g.grad_bo = gbuf[id(spec.Wo)].data_ptr(), gbuf[id(spec.bo)].data_ptr()
    if spec.kind != "original":
        g.grad_Wg, g.grad_bg = gbuf[id(spec.Wg)].data_ptr(), gbuf[id(spec.bg)].data_ptr()
    if spec.kind == "gru":
        g.grad_Wr, g.grad_br = gbuf[id(spec.Wr)].data_ptr(), gbuf[id(spec.br)].data_ptr()
    if stages is not None:
        rec = torch.from_numpy(np.ascontiguousarray(stages)).to(device)
        assert rec.numel() * 4 == _lib.lib().ncde_stage_record_bytes(ctypes.byref(p))
        ws = solver._workspace(p, 2, device)
        rc = _lib.lib().ncde_backward(ctypes.byref(p), rec.data_ptr(), gout.data_ptr(), ctypes.byref(g), ws.data_ptr(),
                                      ws.numel(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "ncde_backward")
    else:
        ws = solver._workspace(p, 1, device)
        rc = _lib.lib().ncde_adjoint(ctypes.byref(p), z_out.data_ptr(), gout.data_ptr(), ctypes.byref(g), ws.data_ptr(),
                                     ws.numel(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "ncde_adjoint")
    torch.cuda.synchronize()
    name = (_lib.lib().ncde_kernel_name(ctypes.byref(p), 2 if stages is not None else 1) or b"?").decode()
    return {"dz0": gz0.cpu().numpy(), "kernel": name, "grads": {k: gbuf[id(v)].cpu().numpy() for k, v in func.p.items() if id(v) in gbuf}}


def coop_status_word(case, pass_, flags=_lib.FLAG_AUTO, device="cuda"):
    """The cooperative status word (include/ncde_hip.h: ncde_coop_status_offset) the LAST call of this case / pass / flags left in the
    workspace arena of the current stream (no other call in between); None if the pass launches nothing cooperative."""
    import ctypes
    from ncde_amd import solver
    m = case["meta"]
    coeffs = torch.from_numpy(case["coeffs"]).to(device)
    func = case_field(case, device)
    z0 = torch.from_numpy(case["z0"]).to(device)
    p = solver.build_problem(coeffs, m["kind"], z0, func.fused_spec(), m["method"],
                             _lib.OUT_KNOTS if m["sequence"] else _lib.OUT_INTERVAL, flags)
    off = _lib.lib().ncde_coop_status_offset(ctypes.byref(p), pass_)
    if off < 0:
        return None
    torch.cuda.synchronize()
    ws = solver._workspace(p, pass_, device)
    return int(ws[off:off + 4].view(torch.int32).cpu()[0])


def kernel_names(case, flags=_lib.FLAG_AUTO, device="cuda"):
    """(forward, adjoint, discrete backward) kernel family names the C-ABI would dispatch this case to."""
    import ctypes
    from ncde_amd import solver
    m = case["meta"]
    coeffs = torch.from_numpy(case["coeffs"]).to(device)
    func = case_field(case, device)
    z0 = torch.from_numpy(case["z0"]).to(device)
    p = solver.build_problem(coeffs, m["kind"], z0, func.fused_spec(), m["method"],
                             _lib.OUT_KNOTS if m["sequence"] else _lib.OUT_INTERVAL, flags)
    lib = _lib.lib()
    return tuple((lib.ncde_kernel_name(ctypes.byref(p), k) or b"?").decode() for k in (0, 1, 2))


def run_times_case(f, meta, adjoint=True, flags=_lib.FLAG_AUTO, device="cuda", kind="original", mode="matmul", params=None,
                   tagged=None):
    """A general-time-axis case (golden g11 layout: coeffs, [knots], t_out, z0, p_*, grad_out) through cdeint."""
    coeffs = torch.from_numpy(f["coeffs"]).to(device)
    kn = torch.from_numpy(f["knots"]).to(device) if "knots" in f else None
    X = (ncde_amd.LinearInterpolation if meta["kind"] == "linear" else ncde_amd.NaturalCubicSpline)(coeffs, t=kn)
    params = params if params is not None else {k[2:]: f[k] for k in f if k.startswith("p_")}
    nl = meta["dims"]["nl"]
    func = CaseField(params, [("W0", "b0")] + [("W1", "b1")] * (nl - 1), device, kind, mode)
    z0 = torch.from_numpy(f["z0"]).to(device).requires_grad_(True)
    t = torch.from_numpy(f["t_out"]).to(device)
    if tagged is not None:          # the control's own (tagged) tensors instead of a plain tensor with the same values
        t = X.interval if tagged == "interval" else X.grid_points
    out = ncde_amd.cdeint(X, func, z0, t, adjoint=adjoint, vector_field_type=mode, method=meta["method"],
                          options={"step_size": meta["step_size"]}, kernel_flags=flags)
    nfe_fwd = func.nfe
    (out * torch.from_numpy(f["grad_out"]).to(device)).sum().backward()
    torch.cuda.synchronize()
    return {"z_out": out.detach().cpu().numpy(), "dz0": z0.grad.cpu().numpy(), "nfe": func.nfe, "nfe_fwd": nfe_fwd,
            "grads": {k: v.grad.cpu().numpy() for k, v in func.p.items() if v.grad is not None}}
