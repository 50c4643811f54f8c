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


CASES = ["tricks_rk4_adj", "tricks_rk4_tape", "detach_rk4_half", "decreasing_midpoint", "decreasing_euler_tape", "interior_rk4_adj"]


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
        assert err <= 1e-6, (name, k, err)


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
