"""CPU: the oracle (oracle/ncde_oracle.py) against the golden vectors produced by the imported
reference (oracle/gen_golden.py).  This is what pins the oracle; GPU parity tests then compare the
HIP path with both the oracle and these same fixtures."""
import json
import os

import numpy as np
import pytest
import torch

import golden_util as gu
import ncde_oracle as orc

# forward: the restatement is bit-exact vs the reference on this image; allow 1 ulp-ish slack for
# other hosts' MKL kernels.  adjoint: hand VJPs vs autograd differ in summation order only.
TOL_Z, TOL_G = 2e-6, 2e-5


@pytest.mark.parametrize("name", gu.SOLVE_CASES)
def test_oracle_matches_reference_golden(name):
    case = gu.load_case(name)
    m = case["meta"]
    field = gu.oracle_field(case)
    ctl = orc.Control(case["coeffs"], m["kind"])
    nfe = [0]
    z = orc.solve_forward(ctl, field, case["z0"], m["method"], m["sequence"], nfe=nfe)
    ex = case["expect"]
    assert z.shape == ex["z_out"].shape
    assert gu.relerr(z, ex["z_out"]) <= TOL_Z
    dz0, gp = orc.solve_adjoint(ctl, field, z, ex["grad_out"], m["method"], m["sequence"], nfe=nfe)
    assert gu.relerr(dz0, ex["dz0"]) <= TOL_G
    for pname, g in zip(m["param_names"], gp):
        if "d" + pname in ex:
            assert gu.relerr(g, ex["d" + pname]) <= TOL_G, pname
        else:
            assert gu.relerr(g.numpy()[::16], ex["d" + pname + "__rows16"]) <= TOL_G, pname
            assert gu.relerr(g.double().sum(0).numpy(), ex["d" + pname + "__colsum"]) <= TOL_G, pname
    stages = {"rk4": 4, "midpoint": 2, "euler": 1}[m["method"]]
    assert nfe[0] == 2 * stages * (ctl.n_knots - 1)      # same nfe accounting as base.py:90 (fwd + adjoint)


def _check_grads(m, ex, dz0, gp, prefix, tol):
    assert gu.relerr(dz0, ex[prefix + "dz0"]) <= tol
    for pname, g in zip(m["param_names"], gp):
        key = prefix + "d" + pname
        if key in ex:
            assert gu.relerr(g, ex[key]) <= tol, pname
        else:
            assert gu.relerr(np.asarray(g)[::16], ex[key + "__rows16"]) <= tol, pname
            assert gu.relerr(np.asarray(g, dtype=np.float64).sum(0), ex[key + "__colsum"]) <= tol, pname


@pytest.mark.parametrize("name", gu.SOLVE_CASES)
def test_oracle_discrete_backward_matches_reference_golden(name):
    """adjoint=False: the reference backpropagates through the solver with autograd; the oracle's hand-written
    reverse sweep must give the same gradients (fixtures bp_*)."""
    case = gu.load_case(name)
    m = case["meta"]
    ctl = orc.Control(case["coeffs"], m["kind"])
    ex = case["expect"]
    dz0, gp = orc.solve_discrete_backward(ctl, gu.oracle_field(case), case["z0"], ex["grad_out"], m["method"], m["sequence"])
    _check_grads(m, ex, dz0.numpy(), [g.numpy() for g in gp], "bp_", TOL_G)


@pytest.mark.parametrize("name", gu.VARIANT_CASES)
def test_oracle_field_variants_match_reference_golden(name):
    """Gated vector fields and the evaluate / derivative input modes: forward, continuous
    adjoint and exact discrete backward of the oracle against the reference's own outputs."""
    case = gu.load_case(name)
    m, ex = case["meta"], case["expect"]
    field = gu.oracle_field(case)
    ctl = orc.Control(case["coeffs"], m["kind"])
    z = orc.solve_forward(ctl, field, case["z0"], m["method"], m["sequence"])
    assert gu.relerr(z, ex["z_out"]) <= TOL_Z
    dz0, gp = orc.solve_adjoint(ctl, field, z, ex["grad_out"], m["method"], m["sequence"])
    _check_grads(m, ex, dz0.numpy(), [g.numpy() for g in gp], "", TOL_G)
    dz0, gp = orc.solve_discrete_backward(ctl, field, case["z0"], ex["grad_out"], m["method"], m["sequence"])
    _check_grads(m, ex, dz0.numpy(), [g.numpy() for g in gp], "bp_", TOL_G)


def test_oracle_full_size_cfg2_forward():
    """BASELINE config 2 at full size (B=4096, T=399): z_T from the reference, inputs regenerated."""
    f = np.load(os.path.join(gu.GOLD, "g5_cfg2_full.npz"))
    torch.set_num_threads(min(8, len(os.sched_getaffinity(0))))
    coeffs = gu.data.make_rectilinear_coeffs(4096, 200, 19, missing=0.3, seed=1234)
    p = gu.data.make_field_weights(32, 32, 20, seed=0)
    rw = gu.data.make_readin_weights(32, 20, 1, seed=0)
    z0 = (coeffs[:, 0] @ rw["Wi"].T + rw["bi"]).astype(np.float32)
    field = orc.Field.original(p, 32, 20, 3)
    zT = orc.solve_forward(orc.Control(coeffs, "linear"), field, z0, "rk4", False)[:, -1]
    assert gu.relerr(zT, f["zT"]) <= TOL_Z


def test_knot_index_rule():
    """Left piece at an exact knot (bucketize right=False), clamped -- SURVEY.md §3.1."""
    ctl = orc.Control(np.zeros((1, 6, 2), np.float32), "linear")
    third = torch.tensor(1.0) * (1 / 3)
    got = [ctl.piece(torch.tensor(float(n))) for n in range(6)]
    assert got == [0, 0, 1, 2, 3, 4]
    assert ctl.piece(torch.tensor(2.0) + third) == 2
    assert ctl.piece(torch.tensor(7.0)) == 4 and ctl.piece(torch.tensor(-1.0)) == 0


def test_manifest_records_oracle_pin():
    with open(os.path.join(gu.GOLD, "MANIFEST.json")) as fh:
        man = json.load(fh)
    for rec in man:
        if "oracle_vs_ref" in rec and isinstance(rec["oracle_vs_ref"], dict):
            assert rec["oracle_vs_ref"]["z"] <= TOL_Z and rec["oracle_vs_ref"]["dtheta"] <= TOL_G


DOPRI5_CASES = ["g10_toy_dopri5_seq", "g10_ncde_dopri5_rect_final", "g10_ncde_dopri5_rect_seq", "g10_ncde_dopri5_cubic_final",
                "g10_ncde_dopri5_cubic_seq", "g10_adaptive_cubic_final"]


def load_dopri5_case(name):
    f = dict(np.load(os.path.join(gu.GOLD, name + ".npz")))
    m = json.loads(str(f["meta"]))
    p = {k[2:]: torch.from_numpy(v) for k, v in f.items() if k.startswith("p_")}
    d = m["dims"]
    if m["field"] == "toy":
        field = orc.Field([(p["W0"], p["b0"]), (p["W1"], p["b1"])], p["Wo"], p["bo"], d["H"], d["C"])
    else:
        field = orc.Field.original(p, d["H"], d["C"], d["nl"])
    ctl = orc.Control(f["coeffs"], m["kind"])
    T = ctl.n_knots
    t = torch.arange(T, dtype=torch.float32) if m["sequence"] else torch.tensor([0.0, T - 1.0])
    return f, m, field, ctl, t


@pytest.mark.parametrize("name", DOPRI5_CASES)
def test_oracle_dopri5_matches_reference_golden(name):
    """Adaptive dopri5 (goldens g10 = outputs of the imported reference).  Forward: the reference's step sequence (same nfe,
    same accepted / rejected counts) and z bit-level.  Adjoint: with the reference's own stage VJP (autograd) again the same
    step sequence and gradients; with the hand VJPs the solve is a rounding-level-different run of the same algorithm --
    same sequence => 1e-4, otherwise the spread two such runs show (documented in gen_golden.py, <= 5e-2)."""
    f, m, field, ctl, t = load_dopri5_case(name)
    opts = m["options"] or None
    sf, sa, sb = {}, {}, {}
    z = orc.dopri5_forward(ctl, field, f["z0"], t, m["rtol"], m["atol"], opts, stats=sf)
    assert sf["nfe"] == m["nfe_fwd"] and [sf["accepted"], sf["rejected"]] == m["steps_fwd"]
    assert gu.relerr(z, f["z_out"]) <= TOL_Z
    dz0, gp = orc.dopri5_adjoint(ctl, field, t, z, f["grad_out"], m["rtol"], m["atol"], opts, stats=sa, vjp="autograd")
    assert sa["nfe"] == m["nfe_bwd"]
    assert gu.relerr(dz0, f["dz0"]) <= TOL_G
    for pname, g in zip(m["param_names"], gp):
        assert gu.relerr(g, f["d" + pname]) <= TOL_G, pname
    dz0, gp = orc.dopri5_adjoint(ctl, field, t, z, f["grad_out"], m["rtol"], m["atol"], opts, stats=sb)
    tol = 1e-4 if sb["nfe"] == m["nfe_bwd"] else 5e-2
    assert gu.relerr(dz0, f["dz0"]) <= tol
    for pname, g in zip(m["param_names"], gp):
        assert gu.relerr(g, f["d" + pname]) <= tol, pname


DOPRI5_TAPED_CASES = ["g12_ncde_dopri5_rect_final", "g12_ncde_dopri5_rect_seq", "g12_ncde_dopri5_linear_final", "g12_ncde_dopri5_cubic_final",
                      "g12_ncde_dopri5_cubic_seq", "g12_adaptive_cubic_final", "g12_first_step_given_rect_seq"]


@pytest.mark.parametrize("name", DOPRI5_TAPED_CASES)
def test_oracle_dopri5_taped_backward_matches_reference_golden(name):
    """dopri5 with adjoint=False (goldens g12 = the imported reference's autograd through its taped adaptive solve -- the setting of
    the shipped "interpolation" experiment grid, configurations.json5:187-191): the oracle's hand-written reverse sweep over the
    accepted steps, dense output and first-step-size gradient included, reproduces z bit-level and every gradient to fp32 round-off."""
    f, m, field, ctl, t = load_dopri5_case(name)
    st = {}
    z, dz0, gp = orc.dopri5_discrete_backward(ctl, field, f["z0"], t, f["grad_out"], m["rtol"], m["atol"], m["options"] or None, stats=st)
    assert st["nfe"] == m["nfe_fwd"] and [st["accepted"], st["rejected"]] == m["steps_fwd"]
    assert st["delta_active"] == m["first_step_differentiable"]
    assert gu.relerr(z, f["z_out"]) <= TOL_Z
    assert gu.relerr(dz0, f["bp_dz0"]) <= TOL_G
    for pname, g in zip(m["param_names"], gp):
        assert gu.relerr(g, f["bp_d" + pname]) <= TOL_G, pname


@pytest.mark.parametrize("name", gu.SOLVE_CASES)
def test_cpu_cabi_restatement_matches_reference_golden(name):
    """oracle/ncde_cpu.cpp -- the scalar C++ / OpenMP restatement behind the SAME C-ABI as the HIP library --
    against the reference's goldens: forward, continuous adjoint, recording forward + exact discrete backward, through the
    identical NcdeProblem / NcdeGrads plumbing the GPU tests use (host pointers instead of device pointers)."""
    import cpu_lib_util as cu
    case = gu.load_case(name)
    m, ex = case["meta"], case["expect"]
    cc = cu.CpuCase(case["coeffs"], m["kind"], case["z0"], case["params"], case["layers"], m["method"], m["sequence"])
    z = cc.forward()
    assert gu.relerr(z, ex["z_out"]) <= TOL_Z
    dz0, g = cc.backward(ex["z_out"], ex["grad_out"])
    zr, rec = cc.forward(record=True)
    assert np.array_equal(zr, z)
    bdz0, bg = cc.backward(rec, ex["grad_out"], discrete=True)
    for prefix, dz, gg in (("", dz0, g), ("bp_", bdz0, bg)):
        assert gu.relerr(dz, ex[prefix + "dz0"]) <= TOL_G, prefix
        for pname in m["param_names"]:
            key = prefix + "d" + pname
            if key in ex:
                assert gu.relerr(gg[pname], ex[key]) <= TOL_G, (prefix, pname)
            else:
                assert gu.relerr(gg[pname][::16], ex[key + "__rows16"]) <= TOL_G, (prefix, pname)


@pytest.mark.parametrize("name", ["g11_times_rk4_half", "g11_times_midpoint_third", "g11_knots_rk4", "g11_knots_cubic_euler",
                                  "g11_knots_interval_rk4", "g11_times_f64_rk4", "g11_times_cubic_rk4_ragged"])
def test_oracle_general_time_axis_matches_reference_golden(name):
    """The oracle's general-time functions (any output times, step size, user knot grid) against the reference's own outputs
    (goldens g11, written by oracle/gen_golden.py from the imported torchcde / torchdiffeq): forward, continuous adjoint
    (one reverse solve per output interval) and the autograd-through-the-solver gradients."""
    import json
    import os
    f = dict(np.load(os.path.join(gu.GOLD, name + ".npz")))
    m = json.loads(str(f["meta"]))
    p = {k[2:]: f[k] for k in f if k.startswith("p_")}
    d = m["dims"]
    field = orc.Field.original(p, d["H"], d["C"], d["nl"])
    ctl = orc.Control(f["coeffs"], m["kind"], t=f["knots"] if "knots" in f else None)
    z = orc.solve_forward_times(ctl, field, f["z0"], f["t_out"], m["method"], m["step_size"])
    assert gu.relerr(z.numpy(), f["z_out"]) <= 2e-6
    dz0, gp = orc.solve_adjoint_times(ctl, field, f["t_out"], z, f["grad_out"], m["method"], m["step_size"])
    assert gu.relerr(dz0.numpy(), f["dz0"]) <= 2e-5
    for n, g in zip(m["param_names"], gp):
        assert gu.relerr(g.numpy(), f["d" + n]) <= 2e-5, n
    bdz0, bgp = orc.solve_discrete_backward_times(ctl, field, f["z0"], f["t_out"], f["grad_out"], m["method"], m["step_size"])
    assert gu.relerr(bdz0.numpy(), f["bp_dz0"]) <= 2e-5
    for n, g in zip(m["param_names"], bgp):
        assert gu.relerr(g.numpy(), f["bp_d" + n]) <= 2e-5, n
