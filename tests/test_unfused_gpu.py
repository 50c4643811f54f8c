"""The unfused torch-op solver behind cdeint (online-neural-cdes_amd/unfused.py): arbitrary vector fields, decreasing output times,
gradients with respect to the control path and the output times -- against golden vectors produced by the imported reference
(oracle/gen_golden_unfused.py -> tests/golden/g13_unfused.npz), plus ports of the reference's own tests
(/root/reference/modules/torchcde/test/test_cdeint.py:5-41, test_tricks.py:21-52, 111-131) with its own inline field."""
import os
import warnings

import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


class _Func(torch.nn.Module):      # the reference tests' inline field (test_tricks.py:6-18, test_cdeint.py:21-27)
    def __init__(self, variable):
        super().__init__()
        self.variable = torch.nn.Parameter(variable)

    def forward(self, t, z):
        return z.sigmoid().unsqueeze(-1) + self.variable


CASES = ["tricks_rk4_adj", "tricks_rk4_tape", "detach_rk4_half", "decreasing_midpoint", "decreasing_euler_tape", "interior_rk4_adj",
         "dopri5_adj_tgrad"]      # (round 5: dL/dt of the adaptive adjoint, forced step sequence)
DOPRI5_OPTS = {"first_step": 0.5, "min_step": 0.5, "max_step": 0.5}


@pytest.mark.parametrize("name", CASES)
def test_unfused_matches_reference_golden(name, gpu_lib):
    import ncde_amd
    f = np.load(os.path.join(gu.GOLD, "g13_unfused.npz"))
    g = lambda k: torch.from_numpy(f[name + "__" + k]).cuda()      # noqa: E731
    interp, method, step, adjoint = [str(x) for x in f[name + "__meta"]]
    step, adjoint = (None if step == "" else float(step)), adjoint == "1"
    coeffs = g("coeffs").requires_grad_(True)
    kn = g("knots").requires_grad_(True)
    X = (ncde_amd.NaturalCubicSpline if interp == "cubic" else ncde_amd.LinearInterpolation)(coeffs, kn)
    func = _Func(g("variable"))
    z0 = g("z0").requires_grad_(True)
    t = g("t").requires_grad_(True)
    kw = {"adjoint_params": tuple(func.parameters()) + (coeffs, kn)} if adjoint else {}
    opts = {} if step is None else {"step_size": step}
    if method == "dopri5":
        opts = dict(DOPRI5_OPTS)
        kw.update(rtol=1e-3, atol=1e-5)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        z = ncde_amd.cdeint(X, func, z0, t, adjoint=adjoint, method=method, options=opts, **kw)
    assert not any("not listed" in str(w.message) for w in rec)
    (z * g("w")).sum().backward()
    got = {"z": z, "d_z0": z0.grad, "d_variable": func.variable.grad, "d_coeffs": coeffs.grad, "d_t": t.grad, "d_knots": kn.grad}
    for k, v in got.items():
        ref = f[name + "__" + k]
        assert v is not None, k
        err = np.abs(v.detach().cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-30)
        assert err <= (2e-5 if method == "dopri5" else 1e-6), (name, k, err)


def test_unfused_warns_once_and_refuses_cpu(gpu_lib):
    import ncde_amd
    from ncde_amd import unfused
    x = torch.rand(2, 6, 3)
    X = ncde_amd.NaturalCubicSpline(torch.from_numpy(gu.data.natural_cubic_coeffs(x.numpy())).cuda())
    func = _Func(torch.rand(1, 1, 3)).cuda()
    unfused._WARNED.clear()
    with pytest.warns(UserWarning, match="unfused torch-op solver .func does not expose fused_spec"):
        ncde_amd.cdeint(X, func, torch.rand(2, 4).cuda(), X.interval, method="rk4", options={"step_size": 1.0})
    with warnings.catch_warnings():
        warnings.simplefilter("error")      # the second call with the same reason is silent
        ncde_amd.cdeint(X, func, torch.rand(2, 4).cuda(), X.interval, method="rk4", options={"step_size": 1.0})
    Xc = ncde_amd.NaturalCubicSpline(torch.from_numpy(gu.data.natural_cubic_coeffs(x.numpy())))
    with pytest.raises(NotImplementedError, match="no CPU fallback"):
        ncde_amd.cdeint(Xc, _Func(torch.rand(1, 1, 3)), torch.rand(2, 4), Xc.interval, method="rk4", options={"step_size": 1.0})


def test_reference_shape_test_with_an_arbitrary_func(gpu_lib):
    """Port of test_cdeint.py:5-41 (rk4 leg) with the reference's own inline _Func: random batch dimensions, fp64 output times,
    step_size = 1 / num_points."""
    import ncde_amd
    gen = torch.Generator().manual_seed(7)
    ri = lambda lo, hi: int(torch.randint(low=lo, high=hi, size=(1,), generator=gen).item())      # noqa: E731
    for _ in range(4):
        num_points, num_channels, num_hidden = ri(5, 30), ri(1, 3), ri(1, 5)
        batch_dims = [ri(1, 3) for _ in range(ri(0, 3))]
        values = torch.rand(*batch_dims, num_points, num_channels, generator=gen)
        coeffs = ncde_amd.natural_cubic_coeffs(values.cuda())
        spline = ncde_amd.NaturalCubicSpline(coeffs)
        f = _Func(torch.rand(*[1 for _ in batch_dims], 1, num_channels, generator=gen)).cuda()
        z0 = torch.rand(*batch_dims, num_hidden, generator=gen).cuda()
        n_out = ri(2, 10)
        start, end = spline.interval
        out_times = torch.rand(n_out, dtype=torch.float64, generator=gen).sort().values.cuda() * (end - start) + start
        out = ncde_amd.cdeint(spline, f, z0, out_times, method="rk4", options={"step_size": 1.0 / num_points}, rtol=1e-4, atol=1e-6)
        assert out.shape == (*batch_dims, n_out, num_hidden)


def test_reference_detach_trick_ported(gpu_lib):
    """Port of test_tricks.py:111-131: rk4 gradients do not depend on whether the output times require a gradient."""
    import ncde_amd
    torch.manual_seed(3)
    path = torch.rand(1, 10, 3)
    interp = ncde_amd.NaturalCubicSpline(ncde_amd.natural_cubic_coeffs(path.cuda()))
    func = _Func(torch.rand(1, 1, 3)).cuda()
    for adjoint in (True, False):
        grads = []
        z0 = torch.rand(1, 3).cuda()
        for t_grad in (True, False):
            t_ = torch.tensor([0.0, 9.0], device="cuda", requires_grad=t_grad)
            z = ncde_amd.cdeint(X=interp, z0=z0, func=func, t=t_, adjoint=adjoint, method="rk4", options=dict(step_size=0.5))
            z[:, -1].sum().backward()
            grads.append(func.variable.grad.clone())
            func.variable.grad.zero_()
        assert (grads[1] == grads[0]).all()


def test_reference_grad_paths_ported(gpu_lib):
    """Port of test_tricks.py:21-52 (rk4 leg): gradients reach the control path, z0, the field and the output times, with and
    without the adjoint.  (The coefficient BUILDERS here are GPU kernels and not differentiable: the leaf is the coefficient tensor.)"""
    import ncde_amd
    torch.manual_seed(5)
    for adjoint in (True, False):
        knots = torch.linspace(0, 9, 10).cuda().requires_grad_(True)
        coeffs = ncde_amd.natural_cubic_coeffs(torch.rand(1, 10, 3).cuda()).detach().requires_grad_(True)
        spline = ncde_amd.NaturalCubicSpline(coeffs, knots)
        z0 = torch.rand(1, 3).cuda().requires_grad_(True)
        func = _Func(torch.rand(1, 1, 3)).cuda()
        t_ = torch.tensor([0.0, 9.0], device="cuda", requires_grad=True)
        kwargs = dict(adjoint_params=tuple(func.parameters()) + (coeffs, knots)) if adjoint else {}
        z = ncde_amd.cdeint(X=spline, func=func, z0=z0, t=t_, adjoint=adjoint, method="rk4", rtol=1e-4, atol=1e-6, **kwargs)
        assert z.shape == (1, 2, 3)
        assert all(v.grad is None for v in (knots, coeffs, z0, func.variable, t_))
        z[:, 1].sum().backward()
        assert all(isinstance(v.grad, torch.Tensor) for v in (knots, coeffs, z0, func.variable, t_))


@pytest.mark.parametrize("kind,mode,interp", [("minimal", "matmul", "linear"), ("gru", "evaluate", "cubic"), ("original", "derivative", "cubic"),
                                              ("gru", "matmul", "linear")])
def test_dopri5_with_gated_fields_and_direct_inputs_runs_unfused(kind, mode, interp, gpu_lib):
    """VERDICT round 3, missing item 5: method='dopri5' with the gated vector fields (src/ncde/vector_fields/gating.py:7-61) and the
    evaluate / derivative inputs (torchcde/solver.py:123-126) -- the combination the reference's 'interpolation' grid uses
    (configurations.json5:187-191).  The fused adaptive kernels evaluate the original field with the matmul input only; these calls now
    run adaptive dopri5 on the unfused torch-op solver (GPU, one UserWarning).  Against the oracle on the forced step sequence (same
    steps on both sides): forward at 1e-5, the adaptive continuous adjoint (stage VJPs by autograd, as the reference) at 1e-4, and
    adjoint=False against autograd through the oracle's own taped solve; then free-running on the smooth control."""
    import warnings
    import ncde_amd
    import ncde_oracle as orc
    from ncde_amd import unfused
    torch.manual_seed(0)
    B, L, C, H, HH, nl = 6, 6, 4, 8, 12, 2
    coeffs = gu.data.make_rectilinear_coeffs(B, L, C - 1, missing=0.3, seed=5) if interp == "linear" else gu.data.make_cubic_coeffs(B, 2 * L, C - 1, seed=6)
    cls = {"original": ncde_amd.OriginalVectorField, "minimal": ncde_amd.MinimalGatedVectorField, "gru": ncde_amd.GRUGatedVectorField}[kind]
    func = cls(C, H, HH, nl, vector_field_type=mode).cuda()
    sp = func.fused_spec()
    p = {"W0": sp.layers[0][0], "b0": sp.layers[0][1], "W1": sp.layers[1][0], "b1": sp.layers[1][1], "Wo": sp.Wo, "bo": sp.bo}
    if sp.Wg is not None:
        p["Wg"], p["bg"] = sp.Wg, sp.bg
    if sp.Wr is not None:
        p["Wr"], p["br"] = sp.Wr, sp.br
    ctl = orc.Control(coeffs, interp)
    X = (ncde_amd.LinearInterpolation if interp == "linear" else ncde_amd.NaturalCubicSpline)(torch.from_numpy(coeffs).cuda())
    z0n = (torch.randn(B, H) * 0.3).numpy()
    tt = torch.arange(ctl.n_knots, dtype=torch.float32)
    unfused._WARNED.clear()
    for opts, tight in (({"first_step": 0.75, "min_step": 0.75, "max_step": 0.75}, True), ({"min_step": 0.25}, False)):
        if not tight and interp == "linear":
            continue      # free-running on a piecewise-linear control: two fp32 implementations part ways (test_gpu_parity: _free_running_z_bar)
        pd = {k: v.detach().cpu().clone() for k, v in p.items()}
        field = orc.Field.variant(pd, H, C, nl, kind, mode)
        z = orc.dopri5_forward(ctl, field, z0n, tt, 1e-3, 1e-5, dict(opts))
        gout = torch.from_numpy((gu.data.normal(31, z.numel(), stream=1).reshape(z.shape) / 3.0).astype(np.float32))
        pd2 = {k: v.detach().clone() for k, v in pd.items()}
        field2 = orc.Field.variant(pd2, H, C, nl, kind, mode)
        dz0, gp = orc.dopri5_adjoint(ctl, field2, tt, z, gout, 1e-3, 1e-5, dict(opts), vjp="autograd")
        uq = field2.unique_params()
        for adjoint in (True, False):
            for q in func.parameters():
                q.grad = None
            z0 = torch.from_numpy(z0n).cuda().requires_grad_(True)
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                out = ncde_amd.cdeint(X, func, z0, X.grid_points, adjoint=adjoint, vector_field_type=mode, method="dopri5", rtol=1e-3, atol=1e-5, options=dict(opts))
            if tight and adjoint:
                assert any("unfused" in str(x.message) and "dopri5" in str(x.message) for x in w), [str(x.message) for x in w]
            assert gu.relerr(out.detach().cpu().numpy(), z.numpy()) <= (1e-5 if tight else 2e-2)
            (out * gout.cuda()).sum().backward()
            if not tight:
                assert torch.isfinite(z0.grad).all()
                continue
            if adjoint:
                assert gu.relerr(z0.grad.cpu().numpy(), dz0.numpy()) <= 1e-4
                # the oracle returns gradients in unique_params() order; map by identity of the CPU tensors
                order = {id(q): i for i, q in enumerate(uq)}
                for k, v_cpu in pd2.items():
                    assert gu.relerr(p[k].grad.cpu().numpy(), gp[order[id(v_cpu)]].numpy()) <= 1e-4, (k, gu.relerr(p[k].grad.cpu().numpy(), gp[order[id(v_cpu)]].numpy()))
            else:      # adjoint=False = autograd through the torch ops of the solve: a different discretisation of the same gradient
                assert gu.relerr(z0.grad.cpu().numpy(), dz0.numpy()) <= 5e-2
                assert all(torch.isfinite(q.grad).all() for q in p.values())


def test_unfused_dopri5_taped_gradient_includes_the_first_step_size(gpu_lib):
    """adjoint=False on the unfused path against the oracle's hand-derived backward of the taped solve (pinned to the reference's
    autograd on g12, incl. the gradient THROUGH _select_initial_step, 2 - 7 % of dL/dz0 there).  In fp64 -- which cdeint sends to the
    unfused solver -- both sides take the same free-running step sequence (3 of 13 attempts rejected, first step selected
    automatically and accepted), so the comparison is at round-off: 1e-10."""
    import ncde_amd
    import ncde_oracle as orc
    torch.manual_seed(1)
    B, L, C, H, HH, nl = 6, 6, 4, 8, 12, 2
    coeffs = gu.data.make_cubic_coeffs(B, 2 * L, C - 1, seed=6).astype(np.float64)
    func = ncde_amd.OriginalVectorField(C, H, HH, nl).double().cuda()
    sp = func.fused_spec()
    pd = {"W0": sp.layers[0][0], "b0": sp.layers[0][1], "W1": sp.layers[1][0], "b1": sp.layers[1][1], "Wo": sp.Wo, "bo": sp.bo}
    field = orc.Field.original({k: v.detach().cpu().numpy() for k, v in pd.items()}, H, C, nl)
    ctl = orc.Control(coeffs, "cubic")
    z0n = (torch.randn(B, H, dtype=torch.float64) * 0.3).numpy()
    tt = torch.arange(ctl.n_knots, dtype=torch.float64)
    opts = {"min_step": 0.25}
    st = {}
    zfw = orc.dopri5_forward(ctl, field, z0n, tt, 1e-3, 1e-5, dict(opts))
    gout = (gu.data.normal(31, zfw.numel(), stream=1).reshape(zfw.shape) / 3.0).astype(np.float64)
    z, dz0, gp = orc.dopri5_discrete_backward(ctl, field, z0n, tt, gout, 1e-3, 1e-5, dict(opts), stats=st)
    assert st["delta_active"] and st["rejected"] >= 1
    X = ncde_amd.NaturalCubicSpline(torch.from_numpy(coeffs).cuda())
    z0 = torch.from_numpy(z0n).cuda().requires_grad_(True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = ncde_amd.cdeint(X, func, z0, X.grid_points.double(), adjoint=False, method="dopri5", rtol=1e-3, atol=1e-5, options=dict(opts))
    assert out.dtype == torch.float64 and gu.relerr(out.detach().cpu().numpy(), np.asarray(z)) <= 1e-10
    (out * torch.from_numpy(gout).cuda()).sum().backward()
    assert gu.relerr(z0.grad.cpu().numpy(), np.asarray(dz0)) <= 1e-10, gu.relerr(z0.grad.cpu().numpy(), np.asarray(dz0))
    for name, want in zip(["W0", "b0", "W1", "b1", "Wo", "bo"], gp):
        assert gu.relerr(pd[name].grad.cpu().numpy(), np.asarray(want)) <= 1e-10, (name, gu.relerr(pd[name].grad.cpu().numpy(), np.asarray(want)))


def test_neuralcde_module_with_a_gated_field_and_dopri5(gpu_lib):
    """NeuralCDE(vector_field='minimal' | 'gru', solver='dopri5') -- the combination of the reference's 'interpolation' grid
    (configurations.json5:187-191) -- end to end through the module: forward + adaptive adjoint, finite, and equal to the oracle's
    free-running solve on the smooth (cubic) control at solver tolerance."""
    import ncde_amd
    import ncde_oracle as orc
    B, L, C, H, HH, nl, OUT = 6, 6, 4, 8, 12, 2, 2
    coeffs = gu.data.make_cubic_coeffs(B, 2 * L, C - 1, seed=6)
    for vf in ("minimal", "gru"):
        torch.manual_seed(3)
        model = ncde_amd.NeuralCDE(C, H, OUT, hidden_hidden_dim=HH, num_layers=nl, interpolation="cubic", solver="dopri5", vector_field=vf,
                                   adjoint=True, return_sequences=True).cuda()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = model(torch.from_numpy(coeffs).cuda())
        assert out.shape[0] == B and out.shape[-1] == OUT and torch.isfinite(out).all()
        out.square().sum().backward()
        assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in model.parameters())
        sp = model.func.fused_spec()
        p = {"W0": sp.layers[0][0], "b0": sp.layers[0][1], "W1": sp.layers[1][0], "b1": sp.layers[1][1], "Wo": sp.Wo, "bo": sp.bo, "Wg": sp.Wg, "bg": sp.bg}
        if sp.Wr is not None:
            p["Wr"], p["br"] = sp.Wr, sp.br
        field = orc.Field.variant({k: v.detach().cpu() for k, v in p.items()}, H, C, nl, vf, "matmul")
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        z0 = torch.from_numpy(coeffs[:, 0, :C]) @ sd["initial_linear.weight"].t() + sd["initial_linear.bias"]
        ctl = orc.Control(coeffs, "cubic")
        z = orc.dopri5_forward(ctl, field, z0, torch.arange(ctl.n_knots, dtype=torch.float32), 1e-3, 1e-5, {"min_step": 0.5})
        want = z @ sd["final_linear.weight"].t() + sd["final_linear.bias"]
        assert gu.relerr(out.detach().cpu().numpy(), want.numpy()) <= 2e-2
