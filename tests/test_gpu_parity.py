"""GPU parity tests proper: the HIP path (through the C-ABI) against
  (1) the golden vectors produced by the imported reference (tests/golden), and
  (2) the oracle (oracle/ncde_oracle.py) on fresh seeded inputs,
plus size-independent properties at BASELINE.json's full sizes.

Tolerances (fp32, stated per BASELINE.json north_star / SURVEY.md §8c):
  forward z: 1e-4 relative (max-abs-diff / max-abs) -- the north-star bar; typically ~1e-6
  adjoint dL/dtheta: 1e-3, dL/dz0: 2e-3 -- the reference's own fp32-vs-fp64 spread is 3e-5..4e-4
"""
import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu

TOL_Z, TOL_DTHETA, TOL_DZ0 = 1e-4, 1e-3, 2e-3
# what we actually expect from an exact-fp32 MFMA implementation; tightened guard against regressions
TIGHT_Z, TIGHT_G = 2e-5, 2e-5
E2E_G = 2e-4   # end-to-end gradient guard on the fixed seeded cases below (no ReLU-mask flips occur on them)

FLAGS = {"generic": 1, "auto": 0}


def _grad_errors(case, res, prefix=""):
    """prefix "" = continuous-adjoint fixtures, "bp_" = adjoint=False (backprop through the solver) fixtures."""
    ex, m = case["expect"], case["meta"]
    errs = {"dz0": gu.relerr(res["dz0"], ex[prefix + "dz0"])}
    for pname in m["param_names"]:
        g = res["grads"][pname]
        key = prefix + "d" + pname
        if key in ex:
            errs[pname] = gu.relerr(g, ex[key])
        else:
            errs[pname] = max(gu.relerr(g[::16], ex[key + "__rows16"]),
                              gu.relerr(g.astype(np.float64).sum(0), ex[key + "__colsum"]))
    return errs


def _check_case(name, flags):
    """End to end (forward kernel -> adjoint kernel) against the reference, at the documented tolerances;
    then the adjoint kernel in isolation (fed the reference's own forward solution) at the tight ones --
    a last-bit difference in z can flip a ReLU mask and move one sample's gradient by ~1e-4, which is
    fp32 behaviour of the model, not a kernel error (seen on g3_cubic_rk4_seq, sample 0)."""
    import gpu_util
    case = gu.load_case(name)
    res = gpu_util.run_case(case, flags=flags)
    ex, m = case["expect"], case["meta"]
    assert res["z_out"].shape == ex["z_out"].shape
    ez = gu.relerr(res["z_out"], ex["z_out"])
    assert ez <= TOL_Z and ez <= TIGHT_Z, ("z", ez)
    for k, e in _grad_errors(case, res).items():
        assert e <= (TOL_DZ0 if k == "dz0" else TOL_DTHETA), ("end-to-end", k, e)
    iso = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=flags)
    for k, e in _grad_errors(case, iso).items():
        assert e <= TIGHT_G, ("adjoint kernel on reference z_out", k, e)
    stages = {"rk4": 4, "midpoint": 2, "euler": 1}[m["method"]]
    n_knots = ex["z_out"].shape[1] if m["sequence"] else (case["coeffs"].shape[1] + (m["kind"] == "cubic"))
    assert res["nfe"] == 2 * stages * (n_knots - 1)


@pytest.mark.parametrize("name", gu.SOLVE_CASES)
def test_generic_kernels_match_reference_golden(name, gpu_lib):
    _check_case(name, FLAGS["generic"])


@pytest.mark.parametrize("name", gu.SOLVE_CASES)
def test_auto_dispatch_matches_reference_golden(name, gpu_lib):
    _check_case(name, FLAGS["auto"])


@pytest.mark.parametrize("name", gu.VARIANT_CASES)
def test_field_variants_match_reference_golden(name, gpu_lib):
    """Gated vector fields (minimal, GRU) and the evaluate / derivative input modes (goldens g9):
    forward, continuous adjoint and exact discrete backward through ncde_fwd_variant / ncde_adj_variant."""
    import gpu_util
    import ncde_oracle as orc
    case = gu.load_case(name)
    m, ex = case["meta"], case["expect"]
    vnames = ("ncde_fwd_variant", "ncde_adj_variant", "ncde_adj_variant<discrete>")
    res = gpu_util.run_case(case)
    if m.get("field_kind") == "minimal" and m.get("field_mode", "matmul") == "matmul":
        # the library zero-pads the golden's odd widths into the batch-tiled family (its gated kernels); the variant kernels are
        # what NCDE_FLAG_FORCE_GENERIC selects, and they are held to the same golden below
        assert all(k.startswith(("ncde_fwd_tiled<", "ncde_adj_tiled<")) and "gated" in k for k in res["kernels"]), res["kernels"]
        resv = gpu_util.run_case(case, flags=1)
        assert resv["kernels"] == vnames, resv["kernels"]
        assert gu.relerr(resv["z_out"], ex["z_out"]) <= TIGHT_Z
        for k, e in _grad_errors(case, resv).items():
            assert e <= (TOL_DZ0 if k == "dz0" else TOL_DTHETA), ("variant kernels end-to-end", k, e)
    else:
        assert res["kernels"] == vnames, res["kernels"]
    assert gu.relerr(res["z_out"], ex["z_out"]) <= TIGHT_Z
    for k, e in _grad_errors(case, res).items():
        assert e <= (TOL_DZ0 if k == "dz0" else TOL_DTHETA), ("end-to-end", k, e)
    iso = gpu_util.run_adjoint_direct(case, ex["z_out"])
    for k, e in _grad_errors(case, iso).items():
        assert e <= TIGHT_G, ("adjoint kernel on reference z_out", k, e)
    resd = gpu_util.run_case(case, adjoint=False)
    for k, e in _grad_errors(case, resd, "bp_").items():
        assert e <= (TOL_DZ0 if k == "dz0" else TOL_DTHETA), ("discrete end-to-end", k, e)
    rec = orc.stage_record(orc.Control(case["coeffs"], m["kind"]), gu.oracle_field(case), case["z0"], m["method"]).numpy()
    isod = gpu_util.run_adjoint_direct(case, ex["z_out"], stages=rec)
    for k, e in _grad_errors(case, isod, "bp_").items():
        assert e <= TIGHT_G, ("backward kernel on the oracle's stage record", k, e)


@pytest.mark.parametrize("vf,vft", [("gru", "evaluate"), ("minimal", "matmul"), ("original", "derivative")])
def test_neuralcde_module_with_field_variants(vf, vft, gpu_lib):
    """NeuralCDE(vector_field=..., vector_field_type=...) end to end (same constructor arguments as the reference,
    src/ncde/ncde.py:42-61) against the oracle evaluated with the module's own parameters."""
    import ncde_amd
    import ncde_oracle as orc
    B, L, C, H, HH, nl, OUT = 9, 6, 5, 16, 24, 3, 2
    coeffs = gu.data.make_rectilinear_coeffs(B, L, C - 1, missing=0.3, seed=51)
    torch.manual_seed(3)
    model = ncde_amd.NeuralCDE(C, H, OUT, hidden_hidden_dim=HH, num_layers=nl, interpolation="rectilinear", vector_field=vf,
                               vector_field_type=vft, adjoint=False, solver="rk4", return_sequences=True).cuda()
    x = torch.from_numpy(coeffs).cuda()
    out = model(x)
    w = torch.from_numpy(gu.data.normal(5, out.numel(), stream=2).reshape(out.shape).astype(np.float32)).cuda()
    (out * w).sum().backward()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    p = {"W0": sd["func.net_to_hh.0.weight"], "b0": sd["func.net_to_hh.0.bias"], "W1": sd["func.net_to_hh.2.weight"], "b1": sd["func.net_to_hh.2.bias"]}
    head = "func.tanh_output_layer.0" if vf == "original" else "func.tanh_net.0"
    p["Wo"], p["bo"] = sd[head + ".weight"], sd[head + ".bias"]
    if vf != "original":
        p["Wg"], p["bg"] = sd["func.sigmoid_net.0.weight"], sd["func.sigmoid_net.0.bias"]
    if vf == "gru":
        p["Wr"], p["br"] = sd["func.reset_net.0.weight"], sd["func.reset_net.0.bias"]
    field = orc.Field.variant(p, H, C, nl, vf, vft)
    ctl = orc.Control(coeffs, "linear")
    z0 = torch.from_numpy(coeffs[:, 0]) @ sd["initial_linear.weight"].t() + sd["initial_linear.bias"]
    z = orc.solve_forward(ctl, field, z0, "rk4", True)
    ref = (z @ sd["final_linear.weight"].t() + sd["final_linear.bias"])[:, ::2]
    assert out.shape == ref.shape and gu.relerr(out.detach().cpu(), ref) <= TIGHT_Z
    gz = torch.zeros_like(z)
    gz[:, ::2] = w.cpu() @ sd["final_linear.weight"]
    dz0, gp = orc.solve_discrete_backward(ctl, field, z0, gz, "rk4", True)
    got = {id(q): q.grad.cpu() for q in model.func.parameters()}
    for q, g in zip(model.func.fused_spec().unique_params(), gp):
        assert gu.relerr(got[id(q)], g) <= E2E_G
    assert gu.relerr(model.initial_linear.weight.grad.cpu(), dz0.t() @ torch.from_numpy(coeffs[:, 0])) <= E2E_G
    assert model.nfe == 4 * (2 * L - 2)


@pytest.mark.parametrize("flags", ["generic", "auto"])
@pytest.mark.parametrize("name", gu.SOLVE_CASES)
def test_discrete_backward_matches_reference_golden(name, flags, gpu_lib):
    """adjoint=False: ncde_forward_record + ncde_backward against the gradients the reference's
    autograd produces by taping the solver (fixtures bp_*).  End to end at the documented tolerances, then the
    backward kernel alone, fed the oracle's stage record, at the tight ones."""
    import gpu_util
    import ncde_oracle as orc
    case = gu.load_case(name)
    m, ex = case["meta"], case["expect"]
    res = gpu_util.run_case(case, flags=FLAGS[flags], adjoint=False)
    assert gu.relerr(res["z_out"], ex["z_out"]) <= TIGHT_Z
    for k, e in _grad_errors(case, res, "bp_").items():
        assert e <= (TOL_DZ0 if k == "dz0" else TOL_DTHETA), ("end-to-end", k, e)
    rec = orc.stage_record(orc.Control(case["coeffs"], m["kind"]), gu.oracle_field(case), case["z0"], m["method"]).numpy()
    iso = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=FLAGS[flags], stages=rec)
    for k, e in _grad_errors(case, iso, "bp_").items():
        assert e <= TIGHT_G, ("backward kernel on the oracle's stage record", k, e)
    stages = {"rk4": 4, "midpoint": 2, "euler": 1}[m["method"]]
    n_knots = ex["z_out"].shape[1] if m["sequence"] else (case["coeffs"].shape[1] + (m["kind"] == "cubic"))
    assert res["nfe"] == stages * (n_knots - 1)        # autograd does not re-evaluate f (base.py:90 counts forward calls)


def test_full_size_cfg2_forward_vs_reference(gpu_lib):
    """BASELINE config 2 at full size against the reference's own z_T (golden g5)."""
    import os
    import ncde_amd
    f = np.load(os.path.join(gu.GOLD, "g5_cfg2_full.npz"))
    coeffs = gu.data.make_rectilinear_coeffs(4096, 200, 19, missing=0.3, seed=1234)
    p = gu.data.make_field_weights(32, 32, 20, seed=0)
    rw = gu.data.make_readin_weights(32, 20, 1, seed=0)
    z0 = (coeffs[:, 0] @ rw["Wi"].T + rw["bi"]).astype(np.float32)
    import gpu_util
    func = gpu_util.CaseField(p, [("W0", "b0"), ("W1", "b1"), ("W1", "b1")], "cuda")
    X = ncde_amd.LinearInterpolation(torch.from_numpy(coeffs).cuda())
    for flags in (0, 1):
        with torch.no_grad():
            zT = ncde_amd.cdeint(X, func, torch.from_numpy(z0).cuda(), X.interval, method="rk4",
                                 options={"step_size": 1}, kernel_flags=flags)[:, -1]
        e = gu.relerr(zT.cpu().numpy(), f["zT"])
        assert e <= TOL_Z and e <= TIGHT_Z, (flags, e)


def _cfg2_case(B):
    coeffs = gu.data.make_rectilinear_coeffs(B, 200, 19, missing=0.3, seed=1234)
    p = gu.data.make_field_weights(32, 32, 20, seed=0)
    rw = gu.data.make_readin_weights(32, 20, 1, seed=0)
    z0 = (coeffs[:, 0] @ rw["Wi"].T + rw["bi"]).astype(np.float32)
    names = ["W0", "b0", "W1", "b1", "Wo", "bo"]
    meta = {"kind": "linear", "method": "rk4", "sequence": False, "param_names": names, "field": "original",
            "dims": {"C": 20, "H": 32, "HH": 32, "nl": 3}}
    gout = (gu.data.normal(3, B * 2 * 32, stream=1).reshape(B, 2, 32) / np.sqrt(2.0)).astype(np.float32)
    return {"meta": meta, "coeffs": coeffs, "z0": z0, "params": p, "layers": [("W0", "b0"), ("W1", "b1"), ("W1", "b1")],
            "H": 32, "C": 20, "expect": {"grad_out": gout}}, names


# Full-length (T = 399, 398 steps = 1592 stages) tolerances for the BENCHMARKED cfg2/cfg3 kernels.  Measured on MI355X
# (this test prints them): against the oracle on the 32-sample sub-batch every gradient -- continuous adjoint (y re-integrated
# backwards over 398 steps) and exact discrete backward, kernel in isolation and end to end -- is within 1.1e-5
# (dz0 3e-7, W0/W1 5e-6, Wo 1.1e-5, biases 2e-6): the fp32 round-off class of the split-bf16 chain vs the oracle's addmm
# order, no drift worth a looser bound.  Guard: 5e-5.  Over all 4096 samples a last-bit difference in z flips a ReLU mask
# for a handful of samples (fp32 behaviour of the model, see _check_case), which moves THAT sample's dL/dz0 at the 1e-3
# level (measured: 25 of 4096 samples beyond 5e-5, median 7e-8, p99 7e-6, worst 1.3e-3): the full-batch check is therefore
# per sample (99 % within 5e-5, none beyond 2e-2), and the parameter gradients -- sums over the batch that include those
# samples -- are held to the documented 1e-3 (measured: W0 6.0e-4, b0 5.2e-4, W1 1.8e-4, Wo 7.9e-5).
CFG2_ISO_G = CFG2_E2E_G = 5e-5
CFG2_FULL_BATCH_G = TOL_DTHETA


def test_full_size_cfg2_adjoint_and_discrete_backward_vs_oracle(gpu_lib):
    """BASELINE configs 2/3 at the benchmarked size (B = 4096 launched, T = 399): `ncde_adj_fast3` and
    `ncde_adj_fast3<discrete>` (adjoint.py:37-145 / autograd through solvers.py:94-119) against the oracle on a 32-sample
    sub-batch that straddles tile boundaries, bit-exact sample independence of z and dL/dz0 between the big batch and the
    sub-batch, and the full-batch parameter gradients of the continuous adjoint against the oracle on all 4096 samples."""
    import gpu_util
    import ncde_oracle as orc
    B = 4096
    big, names = _cfg2_case(B)
    torch.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    rb = gpu_util.run_case(big)                               # forward + continuous adjoint, the benchmarked kernels
    assert rb["kernels"][0].startswith("ncde_fwd_fast_bf3") and rb["kernels"][1].startswith("ncde_adj_fast3"), rb["kernels"]
    assert "discrete" in rb["kernels"][2] and rb["kernels"][2].startswith("ncde_adj_fast3"), rb["kernels"]
    rbd = gpu_util.run_case(big, adjoint=False)               # recording forward + exact discrete backward
    sel = slice(2039, 2071)                                   # tiles 127..129
    sub = dict(big, coeffs=big["coeffs"][sel].copy(), z0=big["z0"][sel].copy(), expect={"grad_out": big["expect"]["grad_out"][sel].copy()})
    rs = gpu_util.run_case(sub)
    rsd = gpu_util.run_case(sub, adjoint=False)
    # (a) samples never interact: bit for bit
    assert np.array_equal(rs["z_out"], rb["z_out"][sel]) and np.array_equal(rsd["z_out"], rb["z_out"][sel])
    assert np.array_equal(rs["dz0"], rb["dz0"][sel])
    assert np.array_equal(rsd["dz0"], rbd["dz0"][sel])
    # (b) sub-batch vs the oracle
    field = gu.oracle_field(sub)
    ctl = orc.Control(sub["coeffs"], "linear")
    z = orc.solve_forward(ctl, field, sub["z0"], "rk4", False)
    assert gu.relerr(rs["z_out"], z) <= TIGHT_Z
    dz0, gp = orc.solve_adjoint(ctl, field, z, sub["expect"]["grad_out"], "rk4", False)
    iso = gpu_util.run_adjoint_direct(sub, z.numpy())
    report = {"iso_dz0": gu.relerr(iso["dz0"], dz0), "e2e_dz0": gu.relerr(rs["dz0"], dz0)}
    for pname, g in zip(names, gp):
        report["iso_" + pname] = gu.relerr(iso["grads"][pname], g)
        report["e2e_" + pname] = gu.relerr(rs["grads"][pname], g)
    bdz0, bgp = orc.solve_discrete_backward(ctl, field, sub["z0"], sub["expect"]["grad_out"], "rk4", False)
    rec = orc.stage_record(ctl, field, sub["z0"], "rk4").numpy()
    isod = gpu_util.run_adjoint_direct(sub, z.numpy(), stages=rec)
    report["bp_iso_dz0"], report["bp_e2e_dz0"] = gu.relerr(isod["dz0"], bdz0), gu.relerr(rsd["dz0"], bdz0)
    for pname, g in zip(names, bgp):
        report["bp_iso_" + pname] = gu.relerr(isod["grads"][pname], g)
        report["bp_e2e_" + pname] = gu.relerr(rsd["grads"][pname], g)
    print("cfg2 full-length parity:", {k: "%.2e" % v for k, v in report.items()})
    for k, v in report.items():
        assert v <= (CFG2_ISO_G if "iso" in k else CFG2_E2E_G), (k, v, report)
    # (c) all 4096 samples: parameter gradients of the benchmarked step against the oracle
    fieldb = gu.oracle_field(big)
    ctlb = orc.Control(big["coeffs"], "linear")
    zb = orc.solve_forward(ctlb, fieldb, big["z0"], "rk4", False)
    assert gu.relerr(rb["z_out"], zb) <= TIGHT_Z
    dz0b, gpb = orc.solve_adjoint(ctlb, fieldb, zb, big["expect"]["grad_out"], "rk4", False)
    per = np.abs(rb["dz0"] - dz0b.numpy()).max(1) / np.abs(dz0b.numpy()).max()
    full = {pname: gu.relerr(rb["grads"][pname], g) for pname, g in zip(names, gpb)}
    print("cfg2 full batch: dz0 per-sample rel err median %.2e, p99 %.2e, max %.2e (%d of %d samples > 5e-5);" %
          (np.median(per), np.quantile(per, 0.99), per.max(), int((per > 5e-5).sum()), B), {k: "%.2e" % v for k, v in full.items()})
    assert np.quantile(per, 0.99) <= 5e-5 and per.max() <= 2e-2, (np.quantile(per, 0.99), per.max())
    for pname, e in full.items():
        assert e <= CFG2_FULL_BATCH_G, (pname, e)


def test_full_size_cfg4_all_samples_vs_oracle(gpu_lib):
    """BASELINE config 4 at the benchmarked size (B = 8192, cubic path of 182 observations, midpoint, H = HH = 64) -- every one of the
    512 workgroups of `ncde_adj_h64`, continuous adjoint AND exact discrete backward, against the oracle on ALL samples: z_T, the
    per-sample dL/dz0 rows (on the oracle's z, so that forward round-off does not enter) and the batch-summed parameter gradients.
    A sum over 8192 samples carries the handful of samples whose ReLU masks flip under ANY fp32 rounding (24 rows here, for every
    kernel family alike -- generic, batch-tiled, this one with fp32-input MFMA throughout: 4e-4 .. 9e-4, tools/cfg4_errors.py): the bar
    for the sums is therefore relative to the fp64 run of the same discrete scheme -- as close to it as the bit-pinned fp32 oracle is,
    within 2 x, with the documented gradient tolerance TOL_DTHETA = 1e-3 as the floor -- for the default (forward side split-fp16) and
    the all-fp32-MFMA instance; the per-sample rows are held to 5e-5 at the 99th percentile."""
    import gpu_util
    import ncde_oracle as orc
    from ncde_amd import _lib
    B, L, C, H, HH, nl, interp, method = 8192, 182, 4, 64, 64, 3, "cubic", "midpoint"
    coeffs = gu.data.make_cubic_coeffs(B, L, C - 1, seed=1234)
    p = gu.data.make_field_weights(H, HH, C, seed=0)
    rw = gu.data.make_readin_weights(H, C, 1, seed=0)
    z0 = (coeffs[:, 0, :C] @ rw["Wi"].T + rw["bi"]).astype(np.float32)
    names = ["W0", "b0", "W1", "b1", "Wo", "bo"]
    meta = {"kind": interp, "method": method, "sequence": False, "param_names": names, "field": "original",
            "dims": {"C": C, "H": H, "HH": HH, "nl": nl}}
    gout = (gu.data.normal(3, B * 2 * H, stream=1).reshape(B, 2, H) / np.sqrt(2.0)).astype(np.float32)
    big = {"meta": meta, "coeffs": coeffs, "z0": z0, "params": p, "layers": [("W0", "b0")] + [("W1", "b1")] * (nl - 1),
           "H": H, "C": C, "expect": {"grad_out": gout}}
    torch.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    rb = gpu_util.run_case(big, need_grads=False)
    assert rb["kernels"][1].startswith("ncde_adj_h64") and rb["kernels"][2].startswith("ncde_adj_h64"), rb["kernels"]
    field, ctl = gu.oracle_field(big), orc.Control(coeffs, interp)
    z = orc.solve_forward(ctl, field, z0, method, False)
    assert gu.relerr(rb["z_out"], z) <= TIGHT_Z
    c64 = dict(big, params={k: v.astype(np.float64) for k, v in p.items()})
    f64, ctl64 = gu.oracle_field(c64), orc.Control(coeffs.astype(np.float64), interp)
    z64 = orc.solve_forward(ctl64, f64, z0.astype(np.float64), method, False)
    rec = orc.stage_record(ctl, field, z0, method).numpy()
    for disc in (False, True):
        if disc:
            d32, g32 = orc.solve_discrete_backward(ctl, field, z0, gout, method, False)
            d64, g64 = orc.solve_discrete_backward(ctl64, f64, z0.astype(np.float64), gout.astype(np.float64), method, False)
        else:
            d32, g32 = orc.solve_adjoint(ctl, field, z, gout, method, False)
            d64, g64 = orc.solve_adjoint(ctl64, f64, z64, gout.astype(np.float64), method, False)
        ref = {n: gu.relerr(a.numpy(), b.numpy()) for n, a, b in zip(names, g32, g64)}
        for flags in (_lib.FLAG_AUTO, _lib.FLAG_FP32_MFMA):
            iso = gpu_util.run_adjoint_direct(big, z.numpy(), flags=flags, stages=rec if disc else None)
            per = np.abs(iso["dz0"] - d32.numpy()).max(1) / np.abs(d32.numpy()).max()
            err = {n: gu.relerr(iso["grads"][n], g.numpy()) for n, g in zip(names, g64)}
            print("cfg4 full batch, %s, flags %d: dz0 rows p99 %.2e max %.2e (%d rows > 1e-5);" % ("discrete backward" if disc else "continuous adjoint", flags,
                  np.quantile(per, 0.99), per.max(), int((per > 1e-5).sum())), {k: "%.1e (fp32 oracle %.1e)" % (err[k], ref[k]) for k in err})
            assert np.quantile(per, 0.99) <= 5e-5 and per.max() <= 2e-2, (np.quantile(per, 0.99), per.max())
            for k in err:
                assert err[k] <= max(2.0 * ref[k], TOL_DTHETA), (disc, flags, k, err[k], ref[k])


_CFG5_CACHE = {}


def _cfg5_1024_case():
    """BASELINE config 5 at its full length on 1024 samples (the seeds of both full-size cfg5 tests below)."""
    B, L, C, H, HH, nl, interp, method = 1024, 400, 80, 128, 128, 3, "linear", "rk4"
    coeffs = gu.data.make_rectilinear_coeffs(B, L, C - 1, missing=0.6, seed=1234)
    p = gu.data.make_field_weights(H, HH, C, seed=0)
    rw = gu.data.make_readin_weights(H, C, 1, seed=0)
    z0 = (coeffs[:, 0] @ rw["Wi"].T + rw["bi"]).astype(np.float32)
    names = ["W0", "b0", "W1", "b1", "Wo", "bo"]
    meta = {"kind": interp, "method": method, "sequence": False, "param_names": names, "field": "original",
            "dims": {"C": C, "H": H, "HH": HH, "nl": nl}}
    gout = (gu.data.normal(3, B * 2 * H, stream=1).reshape(B, 2, H) / np.sqrt(2.0)).astype(np.float32)
    return {"meta": meta, "coeffs": coeffs, "z0": z0, "params": p, "layers": [("W0", "b0")] + [("W1", "b1")] * (nl - 1),
            "H": H, "C": C, "expect": {"grad_out": gout}}


def _cfg5_subset_oracle():
    """fp32 and fp64 oracle (forward, continuous adjoint) on rows 505 .. 536 of that case: the yardstick of both tests -- how far the
    bit-pinned fp32 oracle itself is from the exact result of the same discrete scheme.  Computed once per session."""
    import ncde_oracle as orc
    if "sub" not in _CFG5_CACHE:
        big = _cfg5_1024_case()
        sel = slice(505, 537)                                   # straddles 16-sample tile boundaries
        sub = dict(big, coeffs=big["coeffs"][sel].copy(), z0=big["z0"][sel].copy(), expect={"grad_out": big["expect"]["grad_out"][sel].copy()})
        torch.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
        field, ctl = gu.oracle_field(sub), orc.Control(sub["coeffs"], "linear")
        z = orc.solve_forward(ctl, field, sub["z0"], "rk4", False)
        dz0, gp = orc.solve_adjoint(ctl, field, z, sub["expect"]["grad_out"], "rk4", False)
        c64 = dict(sub, params={k: v.astype(np.float64) for k, v in sub["params"].items()})
        f64, ctl64 = gu.oracle_field(c64), orc.Control(sub["coeffs"].astype(np.float64), "linear")
        z64 = orc.solve_forward(ctl64, f64, sub["z0"].astype(np.float64), "rk4", False)
        dz64, gp64 = orc.solve_adjoint(ctl64, f64, z64, sub["expect"]["grad_out"].astype(np.float64), "rk4", False)
        _CFG5_CACHE["sub"] = {"sel": sel, "sub": sub, "z": z, "dz0": dz0, "gp": gp, "z64": z64, "dz64": dz64, "gp64": gp64}
    return _CFG5_CACHE["sub"]


def test_full_size_cfg5_1024_samples_vs_oracle(gpu_lib):
    """BASELINE config 5 at its full length (T = 799, RK4: 3192 stages; C = 80, H = HH = 128) on 1024 samples -- the XCD-cooperative sweep
    (two groups of 32 workgroups) + the paired gradient pass, against the oracle on ALL samples: z_T, the
    per-sample dL/dz0 rows (on the oracle's z) and the batch-summed parameter gradients.  The continuous adjoint re-integrates y backwards
    over 798 steps, so ANY two fp32 implementations of it differ by what each differs from the exact (fp64) result of the same scheme: a
    few ReLU masks flip, 1e-4 .. 1.5e-3 in the max norm of a gradient.  That band is measured here on the first 32 samples (fp32 oracle vs
    the oracle run in fp64 -- the fp64 run on all 1024 would take the test to ten minutes) and is the yardstick: the rows within 2 x of
    the fp32 oracle's own distance from fp64 (median and 99th percentile), no parameter gradient further from the fp32 oracle's than
    twice the LARGEST distance of that oracle's gradients from their fp64 values (the sums of a 32-sample subset are a noisy yardstick
    tensor by tensor: 1e-4 .. 1.5e-3).  Measured with a 64-sample yardstick: rows median 2.2e-7 (oracle vs fp64: 5.2e-7), p99 5.7e-4
    (3.2e-4), 116 of 1024 rows above 5e-5; sums 2.2e-4 (Wo) .. 1.8e-3 (W0) against 1.1e-4 .. 1.5e-3."""
    import gpu_util
    import ncde_oracle as orc
    big = _cfg5_1024_case()
    coeffs, z0, gout, names, interp, method = big["coeffs"], big["z0"], big["expect"]["grad_out"], big["meta"]["param_names"], "linear", "rk4"
    torch.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    rb = gpu_util.run_case(big, need_grads=False)
    field, ctl = gu.oracle_field(big), orc.Control(coeffs, interp)
    z = orc.solve_forward(ctl, field, z0, method, False)
    assert gu.relerr(rb["z_out"], z) <= TIGHT_Z
    d32, g32 = orc.solve_adjoint(ctl, field, z, gout, method, False)
    d32 = d32.numpy()
    scale = np.abs(d32).max()
    # the yardstick: fp32 oracle vs fp64 oracle on rows 505 .. 536 (shared with test_full_size_cfg4_cfg5_sample_subset_vs_oracle)
    ys = _cfg5_subset_oracle()
    n64 = 32
    ref_rows = np.abs(ys["dz0"].numpy() - ys["dz64"].numpy()).max(1) / scale
    ref = {n: gu.relerr(a.numpy(), b.numpy()) for n, a, b in zip(names, ys["gp"], ys["gp64"])}
    iso = gpu_util.run_adjoint_direct(big, z.numpy())
    assert "coop" in iso["kernel"], iso["kernel"]
    rows = np.abs(iso["dz0"] - d32).max(1) / scale
    err = {n: gu.relerr(iso["grads"][n], g.numpy()) for n, g in zip(names, g32)}
    print("cfg5, 1024 samples, %s: dz0 rows vs the fp32 oracle median %.2e p99 %.2e max %.2e (%d rows > 5e-5); fp32 oracle vs fp64 on %d samples: "
          "median %.2e p99 %.2e max %.2e;" % (iso["kernel"], np.median(rows), np.quantile(rows, 0.99), rows.max(), int((rows > 5e-5).sum()), n64,
                                              np.median(ref_rows), np.quantile(ref_rows, 0.99), ref_rows.max()),
          {k: "%.1e (fp32 vs fp64 oracle, %d samples: %.1e)" % (err[k], n64, ref[k]) for k in err})
    assert np.median(rows) <= max(2.0 * np.median(ref_rows), 1e-6)
    # (the 99th percentile of 1024 rows sits in the mask-flip tail -- 116 rows above 5e-5 --, which 32 yardstick rows sample thinly: floor 1e-3)
    assert np.quantile(rows, 0.99) <= max(2.0 * np.quantile(ref_rows, 0.99), 1e-3) and rows.max() <= 2e-2, (np.quantile(rows, 0.99), rows.max())
    for k in err:
        assert err[k] <= max(2.0 * max(ref.values()), 5e-4), (k, err[k], ref)


@pytest.mark.parametrize("cfg", ["cfg4", "cfg5"])
def test_full_batch_cfg4_cfg5_sample_independence(cfg, gpu_lib):
    """cfg4 at B = 8192 and cfg5 at B = 4096 (their full batch AND length, through forward and backward, i.e. the full
    workspace of the tiled backward): every workgroup / part runs; a 32-sample sub-batch reproduces its rows of the big
    batch bit for bit (z and dL/dz0), and the big batch's parameter gradients are finite and reproducible run to run."""
    import gpu_util
    if cfg == "cfg4":
        B, L, C, H, HH, nl, interp, method = 8192, 182, 4, 64, 64, 3, "cubic", "midpoint"
        coeffs = gu.data.make_cubic_coeffs(B, L, C - 1, seed=1234)
        x0 = coeffs[:, 0, :C]
    else:
        B, L, C, H, HH, nl, interp, method = 4096, 400, 80, 128, 128, 3, "linear", "rk4"
        coeffs = gu.data.make_rectilinear_coeffs(B, L, C - 1, missing=0.6, seed=1234)
        x0 = coeffs[:, 0]
    p = gu.data.make_field_weights(H, HH, C, seed=0)
    rw = gu.data.make_readin_weights(H, C, 1, seed=0)
    z0 = (x0 @ rw["Wi"].T + rw["bi"]).astype(np.float32)
    names = ["W0", "b0", "W1", "b1", "Wo", "bo"]
    meta = {"kind": interp, "method": method, "sequence": False, "param_names": names, "field": "original",
            "dims": {"C": C, "H": H, "HH": HH, "nl": nl}}
    gout = (gu.data.normal(3, B * 2 * H, stream=1).reshape(B, 2, H) / np.sqrt(2.0)).astype(np.float32)
    big = {"meta": meta, "coeffs": coeffs, "z0": z0, "params": p, "layers": [("W0", "b0")] + [("W1", "b1")] * (nl - 1),
           "H": H, "C": C, "expect": {"grad_out": gout}}
    rb = gpu_util.run_case(big)
    sel = slice(B - 1000 - 9, B - 1000 + 23)
    sub = dict(big, coeffs=coeffs[sel].copy(), z0=z0[sel].copy(), expect={"grad_out": gout[sel].copy()})
    rs = gpu_util.run_case(sub)
    if "coop" in rb["kernels"][0]:      # (round 5: the full cfg5 batch also runs the cooperative FORWARD -- exact per-sample scaling, a different
        # rounding than the per-workgroup kernel's -- so its rows are compared at the forward tolerance, the per-workgroup kernel's bit for bit)
        assert gu.relerr(rs["z_out"], rb["z_out"][sel]) <= TIGHT_Z
    else:
        assert np.array_equal(rs["z_out"], rb["z_out"][sel])
    if "coop" in rb["kernels"][1]:
        # round 5: the full cfg5 batch runs the XCD-cooperative sweep (fp16-split transposed product), a 32-sample batch the
        # per-workgroup sweep (fp32 transposed product): two kernels, fp32 round-off apart (the per-workgroup sweep keeps the bitwise property),
        # and over cfg5's 3192 re-integrated stages that is the 1e-4 .. 1.5e-3 band any two fp32 implementations of this sweep sit in
        # (measured against the fp64 oracle in test_full_size_cfg4_cfg5_sample_subset_vs_oracle below)
        from ncde_amd import _lib
        per = np.abs(rs["dz0"] - rb["dz0"][sel]).max(1) / np.abs(rs["dz0"]).max()
        print("cfg5 full batch, cooperative vs per-workgroup sweep: dz0 rows median %.2e max %.2e" % (np.median(per), per.max()))
        assert per.max() <= 1.5e-3 and np.median(per) <= 3e-4, (np.median(per), per.max())
        rbo = gpu_util.run_case(big, flags=_lib.FLAG_NO_COOP)
        assert not any("coop" in k for k in rbo["kernels"]) and np.array_equal(rs["z_out"], rbo["z_out"][sel]) and np.array_equal(rs["dz0"], rbo["dz0"][sel])
        for k in names:
            assert gu.relerr(rb["grads"][k], rbo["grads"][k]) <= 1.5e-3, k      # (batch-summed over 4096 samples x 3192 stages; the same band)
    else:
        assert np.array_equal(rs["dz0"], rb["dz0"][sel])
    assert all(np.isfinite(g).all() for g in rb["grads"].values())
    rb2 = gpu_util.run_case(big)
    assert all(np.array_equal(rb2["grads"][k], rb["grads"][k]) for k in names)


@pytest.mark.parametrize("cfg", ["cfg4", "cfg5"])
def test_full_size_cfg4_cfg5_sample_subset_vs_oracle(cfg, gpu_lib):
    """BASELINE configs 4 and 5 at their full length T (and 1024-sample batches, so every workgroup / part of the
    dispatched kernels is exercised): (a) samples do not interact -- a 32-sample sub-batch reproduces the corresponding
    rows of the big batch bit for bit in the forward; (b) the sub-batch forward and backward match the oracle."""
    import gpu_util
    import ncde_oracle as orc
    if cfg == "cfg4":
        B, L, C, H, HH, nl, interp, method = 1024, 182, 4, 64, 64, 3, "cubic", "midpoint"
        coeffs = gu.data.make_cubic_coeffs(B, L, C - 1, seed=1234)
        x0 = coeffs[:, 0, :C]
    else:
        B, L, C, H, HH, nl, interp, method = 1024, 400, 80, 128, 128, 3, "linear", "rk4"
        coeffs = gu.data.make_rectilinear_coeffs(B, L, C - 1, missing=0.6, seed=1234)
        x0 = coeffs[:, 0]
    p = gu.data.make_field_weights(H, HH, C, seed=0)
    rw = gu.data.make_readin_weights(H, C, 1, seed=0)
    z0 = (x0 @ rw["Wi"].T + rw["bi"]).astype(np.float32)
    layers = [("W0", "b0")] + [("W1", "b1")] * (nl - 1)
    names = ["W0", "b0", "W1", "b1", "Wo", "bo"]
    meta = {"kind": interp, "method": method, "sequence": False, "param_names": names, "field": "original",
            "dims": {"C": C, "H": H, "HH": HH, "nl": nl}}
    gout = (gu.data.normal(3, B * 2 * H, stream=1).reshape(B, 2, H) / np.sqrt(2.0)).astype(np.float32)
    big = {"meta": meta, "coeffs": coeffs, "z0": z0, "params": p, "layers": layers, "H": H, "C": C, "expect": {"grad_out": gout}}
    rb = gpu_util.run_case(big, need_grads=False)
    sel = slice(505, 537)                                   # straddles 16-sample tile boundaries
    sub = dict(big, coeffs=coeffs[sel].copy(), z0=z0[sel].copy(), expect={"grad_out": gout[sel].copy()})
    rs = gpu_util.run_case(sub)
    if "coop" in rb["kernels"][0]:      # (cfg5, round 5: the 1024-sample batch runs the cooperative forward; see the test above)
        assert gu.relerr(rs["z_out"], rb["z_out"][sel]) <= TIGHT_Z
        rbo = gpu_util.run_case(big, need_grads=False, flags=gpu_util._lib.FLAG_NO_COOP)
        assert rs["kernels"][0] == rbo["kernels"][0] and np.array_equal(rs["z_out"], rbo["z_out"][sel])
    else:
        assert rs["kernels"][0] == rb["kernels"][0]
        assert np.array_equal(rs["z_out"], rb["z_out"][sel])
    torch.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    if cfg == "cfg5":      # (fp32 and fp64 oracle of these rows: computed once per session, shared with the 1024-sample test above)
        ys = _cfg5_subset_oracle()
        assert ys["sel"] == sel and np.array_equal(ys["sub"]["coeffs"], sub["coeffs"]) and np.array_equal(ys["sub"]["z0"], sub["z0"])
        z, dz0, gp = ys["z"], ys["dz0"], ys["gp"]
    else:
        field = gu.oracle_field(sub)
        ctl = orc.Control(sub["coeffs"], interp)
        z = orc.solve_forward(ctl, field, sub["z0"], method, False)
        dz0, gp = orc.solve_adjoint(ctl, field, z, sub["expect"]["grad_out"], method, False)
    assert gu.relerr(rs["z_out"], z) <= TIGHT_Z
    iso = gpu_util.run_adjoint_direct(sub, z.numpy())
    if cfg == "cfg4":      # the tight guard
        assert gu.relerr(iso["dz0"], dz0) <= E2E_G
        for pname, g in zip(names, gp):
            assert gu.relerr(iso["grads"][pname], g) <= E2E_G, pname
        return
    # cfg5 re-integrates y over 798 steps (3192 stages) and the max-norm gradient error of ANY fp32 implementation of that sweep is
    # 1e-4 .. 1.5e-3 (a few ReLU masks flip): measured here by running the oracle itself in fp64 (the exact-arithmetic version of
    # the same discrete scheme; the fp32 oracle is bit-pinned to the reference).  The bar for the kernel: as close to the fp64
    # result as the fp32 reference arithmetic gets, within 2x (floor 5e-4) -- for the split-bf16 and the fp32-input MFMA path.
    dz64, gp64 = ys["dz64"], ys["gp64"]
    ref = {"dz0": gu.relerr(dz0.numpy(), dz64.numpy())}
    ref.update({n: gu.relerr(g.numpy(), g64.numpy()) for n, g, g64 in zip(names, gp, gp64)})
    for flags in (gpu_util._lib.FLAG_AUTO, gpu_util._lib.FLAG_FP32_MFMA):
        it = iso if flags == gpu_util._lib.FLAG_AUTO else gpu_util.run_adjoint_direct(sub, z.numpy(), flags=flags)
        err = {"dz0": gu.relerr(it["dz0"], dz64.numpy())}
        err.update({n: gu.relerr(it["grads"][n], g64.numpy()) for n, g64 in zip(names, gp64)})
        print("flags", flags, {k: "%.1e (fp32 oracle %.1e)" % (err[k], ref[k]) for k in err})
        for k in err:
            assert err[k] <= max(2.0 * ref[k], 5e-4), (flags, k, err[k], ref[k])
    # round 5: the 1024-sample batch takes the XCD-cooperative sweep (the 32-sample one cannot: two sample tiles); its rows of dL/dz0 meet the
    # same bar against the fp64 oracle at the full length
    itc = gpu_util.run_adjoint_direct(big, rb["z_out"])
    assert "coop" in itc["kernel"] and "coop" not in iso["kernel"], (itc["kernel"], iso["kernel"])
    e = gu.relerr(itc["dz0"][sel], dz64.numpy())
    print("cooperative sweep, 1024 samples, rows %d..%d of dz0 vs fp64 oracle: %.1e (fp32 oracle %.1e)" % (sel.start, sel.stop, e, ref["dz0"]))
    assert e <= max(2.0 * ref["dz0"], 5e-4), (e, ref["dz0"])


def test_ragged_batch_and_determinism(gpu_lib):
    """B not a multiple of the 16-sample tile; two runs are bit-identical (deterministic reductions)."""
    import gpu_util
    import ncde_oracle as orc
    case = gu.load_case("g2_rect_rk4_final")
    for B in (1, 5, 17):
        sub = dict(case, coeffs=case["coeffs"][:B].copy(), z0=case["z0"][:B].copy(),
                   expect={"grad_out": case["expect"]["grad_out"][:B].copy()})
        r1 = gpu_util.run_case(sub)
        r2 = gpu_util.run_case(sub)
        assert np.array_equal(r1["z_out"], r2["z_out"]) and np.array_equal(r1["dz0"], r2["dz0"])
        for k in r1["grads"]:
            assert np.array_equal(r1["grads"][k], r2["grads"][k]), k
        field = gu.oracle_field(sub)
        ctl = orc.Control(sub["coeffs"], "linear")
        z = orc.solve_forward(ctl, field, sub["z0"], "rk4", False)
        dz0, gp = orc.solve_adjoint(ctl, field, z, sub["expect"]["grad_out"], "rk4", False)
        assert gu.relerr(r1["z_out"], z) <= TIGHT_Z
        assert gu.relerr(r1["dz0"], dz0) <= E2E_G
        for pname, g in zip(case["meta"]["param_names"], gp):
            assert gu.relerr(r1["grads"][pname], g) <= E2E_G, pname


def test_sample_independence_and_linearity_of_adjoint(gpu_lib):
    """Size-independent properties: (a) each sample's solution does not depend on its batch mates;
    (b) the adjoint is linear in grad_out."""
    import gpu_util
    case = gu.load_case("g2_rect_rk4_seq")
    full = gpu_util.run_case(case)
    perm = np.random.RandomState(0).permutation(case["coeffs"].shape[0])
    pc = dict(case, coeffs=case["coeffs"][perm].copy(), z0=case["z0"][perm].copy(),
              expect={"grad_out": case["expect"]["grad_out"][perm].copy()})
    pr = gpu_util.run_case(pc)
    assert np.array_equal(pr["z_out"], full["z_out"][perm])
    assert gu.relerr(pr["dz0"], full["dz0"][perm]) <= 1e-6
    scaled = dict(case, expect={"grad_out": (2.0 * case["expect"]["grad_out"]).astype(np.float32)})
    sr = gpu_util.run_case(scaled)
    assert gu.relerr(sr["dz0"], 2.0 * full["dz0"]) <= 1e-6
    for k in full["grads"]:
        assert gu.relerr(sr["grads"][k], 2.0 * full["grads"][k]) <= 1e-5, k


def test_module_level_matches_reference(gpu_lib):
    """NeuralCDE (h0, static features, readout, rectilinear filtering) against the reference module (golden g7)."""
    import json
    import os
    import ncde_amd
    f = np.load(os.path.join(gu.GOLD, "g7_module.npz"))
    meta = json.loads(str(f["meta"]))
    d = meta["dims"]
    coeffs = torch.from_numpy(f["coeffs"]).cuda()
    static = torch.from_numpy(f["static"]).cuda()
    for vname, kw in meta["variants"].items():
        model = ncde_amd.NeuralCDE(d["C"], d["H"], d["OUT"], hidden_hidden_dim=d["HH"], num_layers=d["nl"],
                                   adjoint=True, solver="rk4", **kw).cuda()
        sd = {k[len(vname) + 6:]: torch.from_numpy(f[k]) for k in f.files if k.startswith(vname + "__sd__")}
        model.load_state_dict(sd)           # the reference's state_dict loads as is
        inp = (static, coeffs) if kw.get("static_dim") else coeffs
        out = model(inp)
        assert gu.relerr(out.detach().cpu().numpy(), f[vname + "__out"]) <= TIGHT_Z, vname
        (out * torch.from_numpy(f[vname + "__w"]).cuda()).sum().backward()
        for k, prm in model.named_parameters():
            ref = f[f"{vname}__grad__{k}"]
            assert gu.relerr(prm.grad.cpu().numpy(), ref) <= E2E_G, (vname, k)
        assert model.nfe == int(f[vname + "__nfe"]), vname


def _seeded_case(interp, method, seq, B, L, C, H, HH, nl, seed, kind="original"):
    """A fresh case (inputs from the deterministic generator, expectation from the oracle)."""
    import torch as _t
    import ncde_oracle as orc
    if interp == "cubic":
        coeffs = gu.data.make_cubic_coeffs(B, L, C - 1, seed=seed)
        x0 = coeffs[:, 0, :C]
    else:
        coeffs = gu.data.make_rectilinear_coeffs(B, L, C - 1, missing=0.3, seed=seed)
        x0 = coeffs[:, 0]
    p = gu.data.make_field_weights(H, HH, C, seed=seed + 1) if kind == "original" else \
        gu.data.make_variant_weights(H, HH, C, seed=seed + 1, kind=kind, mode="matmul")
    rw = gu.data.make_readin_weights(H, C, 1, seed=seed + 1)
    z0 = (x0 @ rw["Wi"].T + rw["bi"]).astype(np.float32)
    layers = [("W0", "b0")] + [("W1", "b1")] * (nl - 1)
    names = ["W0", "b0"] + (["W1", "b1"] if nl > 1 else []) + (["Wr", "br"] if kind == "gru" else []) + \
        (["Wg", "bg"] if kind != "original" else []) + ["Wo", "bo"]
    if nl == 1:
        p = {k: v for k, v in p.items() if k not in ("W1", "b1")}
    case = {"meta": {"kind": interp, "method": method, "sequence": seq, "param_names": names, "field_kind": kind, "field_mode": "matmul",
                     "dims": {"C": C, "H": H, "HH": HH, "nl": nl}, "field": "original"},
            "coeffs": coeffs, "z0": z0, "params": p, "layers": layers, "H": H, "C": C}
    field = gu.oracle_field(case)
    ctl = orc.Control(coeffs, interp)
    z = orc.solve_forward(ctl, field, z0, method, seq)
    gout = (gu.data.normal(seed + 2, z.numel(), stream=1).reshape(z.shape) / np.sqrt(z.shape[1])).astype(np.float32)
    dz0, gp = orc.solve_adjoint(ctl, field, z, gout, method, seq)
    case["expect"] = {"z_out": z.numpy(), "grad_out": gout, "dz0": dz0.numpy()}
    for n_, g_ in zip(names, gp):
        case["expect"]["d" + n_] = g_.numpy()
    bdz0, bgp = orc.solve_discrete_backward(ctl, field, z0, gout, method, seq)      # adjoint=False gradients
    case["expect"]["bp_dz0"] = bdz0.numpy()
    for n_, g_ in zip(names, bgp):
        case["expect"]["bp_d" + n_] = g_.numpy()
    case["stage_record"] = orc.stage_record(ctl, field, z0, method).numpy()
    return case


FAST_SHAPES = [(20, 32, 32, 3), (4, 64, 64, 3)]   # (C, H, HH, nl) with a shape-specialised kernel


@pytest.mark.parametrize("shape", FAST_SHAPES)
@pytest.mark.parametrize("interp", ["linear", "cubic"])
@pytest.mark.parametrize("method", ["rk4", "midpoint", "euler"])
@pytest.mark.parametrize("seq", [False, True])
def test_fast_kernels_vs_oracle(shape, interp, method, seq, gpu_lib):
    """Every instantiation of the shape-specialised family (interp x method x output mode), ragged batch,
    against the oracle; the adjoint kernel is additionally checked in isolation on the oracle's z."""
    import ctypes
    import gpu_util
    from ncde_amd import _lib
    C, H, HH, nl = shape
    case = _seeded_case(interp, method, seq, B=21, L=9, C=C, H=H, HH=HH, nl=nl, seed=100 + C)
    res = gpu_util.run_case(case, flags=_lib.FLAG_AUTO)
    assert res["kernels"][0].startswith("ncde_fwd_fast_bf3") and "fp16x2" in res["kernels"][0], res["kernels"]   # default: split-fp16 GEMMs
    ex = case["expect"]
    assert gu.relerr(res["z_out"], ex["z_out"]) <= TIGHT_Z
    resb = gpu_util.run_case(case, flags=_lib.FLAG_SPLIT_BF16, need_grads=False)  # 3-way split-bf16 variant (also the range-fault re-execution path)
    assert "bf16x3" in resb["kernels"][0], resb["kernels"]
    assert gu.relerr(resb["z_out"], ex["z_out"]) <= TIGHT_Z
    res32 = gpu_util.run_case(case, flags=_lib.FLAG_FP32_MFMA, need_grads=False)  # plain fp32-input MFMA variant
    assert res32["kernels"][0].startswith("ncde_fwd_fast<"), res32["kernels"]
    assert gu.relerr(res32["z_out"], ex["z_out"]) <= TIGHT_Z
    for k, e in _grad_errors(case, res).items():
        assert e <= (TOL_DZ0 if k == "dz0" else TOL_DTHETA), ("end-to-end", k, e)
    iso = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=_lib.FLAG_AUTO)
    for k, e in _grad_errors(case, iso).items():
        assert e <= TIGHT_G, ("adjoint kernel on oracle z_out", res["kernels"][1], k, e)
    if H == 32:
        # default: chain + gradient waves; forward-side GEMMs of the chain waves split-fp16, the cotangent side split-bf16
        assert res["kernels"][1].startswith("ncde_adj_fast3") and "fp16x2 + bf16x3" in res["kernels"][1], res["kernels"]
        isob = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=_lib.FLAG_SPLIT_BF16)             # all split-bf16
        for k, e in _grad_errors(case, isob).items():
            assert e <= TIGHT_G, ("split-bf16 adjoint kernel on oracle z_out", k, e)
        for fl, nm in ((_lib.FLAG_ADJOINT_V1, "single-role"), (_lib.FLAG_ADJOINT_V2, "chain+grad fp32 chain"),
                       (_lib.FLAG_ADJOINT_V4, "decoupled y / cotangent waves")):      # also the other specialised adjoint variants
            iso1 = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=fl)
            for k, e in _grad_errors(case, iso1).items():
                assert e <= TIGHT_G, (nm + " adjoint kernel on oracle z_out", k, e)
        again = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=_lib.FLAG_AUTO)     # hand-off protocol: bit-reproducible
        assert np.array_equal(again["dz0"], iso["dz0"]) and all(np.array_equal(again["grads"][k], iso["grads"][k]) for k in iso["grads"])
    if H == 64:
        # the in-sweep adjoint of ncde_fast64.hip: one / two sample tiles per workgroup, split-fp16 forward side / all fp32-input MFMA
        assert res["kernels"][1].startswith("ncde_adj_h64") and "fp16x2" in res["kernels"][1], res["kernels"]
        assert res["kernels"][2].startswith("ncde_adj_h64") and "discrete" in res["kernels"][2], res["kernels"]
        for fl in (_lib.FLAG_TILED_NS1, _lib.FLAG_TILED_NS2, _lib.FLAG_TILED_NS2 | _lib.FLAG_FP32_MFMA, _lib.FLAG_TILED_NS1 | _lib.FLAG_SPLIT_BF16):
            isn = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=fl)
            for k, e in _grad_errors(case, isn).items():
                assert e <= TIGHT_G, ("ncde_adj_h64 flags %#x on oracle z_out" % fl, k, e)
            isnd = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=fl, stages=case["stage_record"])
            for k, e in _grad_errors(case, isnd, "bp_").items():
                assert e <= TIGHT_G, ("ncde_adj_h64 discrete flags %#x on the oracle's stage record" % fl, k, e)
        again = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=_lib.FLAG_TILED_NS2)
        first = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=_lib.FLAG_TILED_NS2)
        assert np.array_equal(again["dz0"], first["dz0"]) and all(np.array_equal(again["grads"][k], first["grads"][k]) for k in first["grads"])
    # adjoint=False: recording forward + exact discrete backward
    resd = gpu_util.run_case(case, flags=_lib.FLAG_AUTO, adjoint=False)
    assert np.array_equal(resd["z_out"], res["z_out"])                      # recording does not change the solution
    for k, e in _grad_errors(case, resd, "bp_").items():
        assert e <= (TOL_DZ0 if k == "dz0" else TOL_DTHETA), ("discrete end-to-end", k, e)
    isod = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=_lib.FLAG_AUTO, stages=case["stage_record"])
    for k, e in _grad_errors(case, isod, "bp_").items():
        assert e <= TIGHT_G, ("discrete backward kernel on the oracle's stage record", res["kernels"][2], k, e)
    if H == 32:
        assert res["kernels"][2].startswith("ncde_adj_fast3") and "discrete" in res["kernels"][2], res["kernels"]
        iso4 = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=_lib.FLAG_ADJOINT_V4, stages=case["stage_record"])
        for k, e in _grad_errors(case, iso4, "bp_").items():
            assert e <= TIGHT_G, ("decoupled-waves discrete backward on the oracle's stage record", k, e)
        againd = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=_lib.FLAG_AUTO, stages=case["stage_record"])
        assert np.array_equal(againd["dz0"], isod["dz0"]) and all(np.array_equal(againd["grads"][k], isod["grads"][k]) for k in isod["grads"])


@pytest.mark.parametrize("interp,method", [("linear", "rk4"), ("cubic", "midpoint")])
def test_split_fp16_forward_range_fault_is_reexecuted(interp, method, gpu_lib):
    """The default forward kernel multiplies in 2-way split-fp16 and speculates on the fp16 range (|operand| < 6e4); a sample tile
    whose operands leave it is re-executed by the split-bf16 kernel.  Three tiles (B = 37); one sample of the middle tile gets a
    z0 row large enough to overflow fp16 in the first hidden layer.  Expected: the middle tile's rows are bit-identical to the
    split-bf16 run (they came from that kernel), the other tiles' rows are bit-identical to the run without the outlier (samples
    do not interact in the forward solve), and everything agrees with the oracle."""
    import gpu_util
    from ncde_amd import _lib
    for seq in (False, True):
        case = _seeded_case(interp, method, seq, B=37, L=6, C=20, H=32, HH=32, nl=3, seed=311)
        plain = gpu_util.run_case(case, need_grads=False)
        big = dict(case)
        big["z0"] = case["z0"].copy()
        big["z0"][21] *= 4.0e5
        import ncde_oracle as orc
        ctl = orc.Control(big["coeffs"], case["meta"]["kind"])
        zo = orc.solve_forward(ctl, gu.oracle_field(case), big["z0"], method, seq).numpy()
        r16 = gpu_util.run_case(big, need_grads=False)
        rbf = gpu_util.run_case(big, flags=_lib.FLAG_SPLIT_BF16, need_grads=False)
        assert np.isfinite(r16["z_out"]).all()
        assert np.array_equal(r16["z_out"][16:32], rbf["z_out"][16:32])                       # re-executed tile
        assert np.array_equal(r16["z_out"][:16], plain["z_out"][:16]) and np.array_equal(r16["z_out"][32:], plain["z_out"][32:])
        scale = np.abs(zo).max(axis=tuple(range(1, zo.ndim)), keepdims=True)                   # per sample: the outlier is 1e5 x the rest
        assert float((np.abs(r16["z_out"] - zo) / scale).max()) <= TIGHT_Z


@pytest.mark.parametrize("shape", [(20, 32, "linear", "rk4"), (4, 64, "cubic", "midpoint")])
@pytest.mark.parametrize("disc", [False, True])
def test_split_fp16_adjoint_range_fault_is_reexecuted(disc, shape, gpu_lib):
    """The default adjoint / discrete-backward kernel recomputes the forward side of each stage in split-fp16 and speculates on the
    fp16 range like the forward kernel does.  One sample of the middle tile gets a state large enough to overflow: that tile's dz0
    rows must be bit-identical to the all-split-bf16 run (the tile was re-executed by that kernel), the other tiles' rows
    bit-identical to the run without the outlier, and the parameter gradients must agree with the all-split-bf16 run to fp32 round-off."""
    import gpu_util
    from ncde_amd import _lib
    C, H, interp, method = shape      # (4, 64, ...): ncde_adj_h64, whose re-execution instance multiplies in fp32-input MFMA (same flag)
    case = _seeded_case(interp, method, False, B=37, L=6, C=C, H=H, HH=H, nl=3, seed=313)
    z = case["expect"]["z_out"].copy()
    rec = case["stage_record"].copy()
    kw = {}
    plain = gpu_util.run_adjoint_direct(case, z, stages=rec if disc else None)
    z[21] *= 4.0e5
    rec[:, 21] *= 4.0e5
    if disc:
        kw["stages"] = rec
    r16 = gpu_util.run_adjoint_direct(case, z, **kw)
    rbf = gpu_util.run_adjoint_direct(case, z, flags=_lib.FLAG_SPLIT_BF16, **kw)
    assert np.isfinite(r16["dz0"]).all()
    assert np.array_equal(r16["dz0"][16:32], rbf["dz0"][16:32])
    assert np.array_equal(r16["dz0"][:16], plain["dz0"][:16]) and np.array_equal(r16["dz0"][32:], plain["dz0"][32:])
    for k in rbf["grads"]:
        assert gu.relerr(r16["grads"][k], rbf["grads"][k]) <= TIGHT_G, k


@pytest.mark.parametrize("interp,method,seq", [("linear", "rk4", False), ("cubic", "midpoint", True), ("cubic", "euler", False)])
def test_default_adjoint_is_reproducible_under_repetition(interp, method, seq, gpu_lib):
    """The default adjoint / discrete-backward kernel (chain waves + gradient waves handing dP tiles, x images and dL/dx_L partials
    to each other through LDS flags and two barriers per stage): ten launches on the oracle's z, ragged three-workgroup batch, must be
    bit-identical and within the tight tolerance.  (An experimental all-split-fp16 variant of this kernel failed exactly this kind
    of test -- DESIGN.md section 5.4c; `tools/stress_adj.py` runs the full matrix.)"""
    import gpu_util
    case = _seeded_case(interp, method, seq, B=37, L=7, C=20, H=32, HH=32, nl=3, seed=77)
    ex = case["expect"]
    for kw, pre in (({}, ""), ({"stages": case["stage_record"]}, "bp_")):
        first = None
        for _ in range(10):
            iso = gpu_util.run_adjoint_direct(case, ex["z_out"], **kw)
            for k, e in _grad_errors(case, iso, pre).items():
                assert e <= TIGHT_G, (interp, method, pre, k, e)
            if first is None:
                first = iso
            else:
                assert np.array_equal(first["dz0"], iso["dz0"]) and all(np.array_equal(first["grads"][k], iso["grads"][k]) for k in iso["grads"])


@pytest.mark.parametrize("C,nl", [(4, 1), (4, 2), (4, 4), (3, 3), (2, 2)])
@pytest.mark.parametrize("interp,method,seq", [("linear", "rk4", True), ("cubic", "midpoint", False)])
def test_h64_in_sweep_adjoint_other_layer_counts_and_channels(C, nl, interp, method, seq, gpu_lib):
    """ncde_adj_h64 has a runtime layer count (1..4) and takes any C <= 4 (rows of missing channels are zero weights): continuous
    adjoint and exact discrete backward on the oracle's z / stage record, one and two sample tiles per workgroup, ragged batch."""
    import gpu_util
    from ncde_amd import _lib
    case = _seeded_case(interp, method, seq, B=37, L=7, C=C, H=64, HH=64, nl=nl, seed=640 + 10 * C + nl)
    ex = case["expect"]
    for fl in (_lib.FLAG_AUTO, _lib.FLAG_TILED_NS2):
        iso = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=fl)
        assert gpu_util.kernel_names(case, fl)[1].startswith("ncde_adj_h64")
        for k, e in _grad_errors(case, iso).items():
            assert e <= TIGHT_G, ("continuous", fl, k, e)
        isod = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=fl, stages=case["stage_record"])
        for k, e in _grad_errors(case, isod, "bp_").items():
            assert e <= TIGHT_G, ("discrete", fl, k, e)


@pytest.mark.parametrize("nl", [1, 2, 4])
def test_fast_forward_with_other_layer_counts(nl, gpu_lib):
    """The specialised forward kernels have a runtime layer count besides the unrolled nl = 3 instantiation.  nl = 1, 2, 4 at both specialised
    shapes, split-fp16 (default) and split-bf16, against the oracle -- forward and end-to-end gradients."""
    import gpu_util
    from ncde_amd import _lib
    for (C, H, HH, _) in FAST_SHAPES:
        case = _seeded_case("linear", "rk4", True, B=21, L=6, C=C, H=H, HH=HH, nl=nl, seed=400 + nl)
        ex = case["expect"]
        res = gpu_util.run_case(case)
        assert res["kernels"][0].startswith("ncde_fwd_fast_bf3") and "fp16x2" in res["kernels"][0], res["kernels"]
        assert gu.relerr(res["z_out"], ex["z_out"]) <= TIGHT_Z
        resb = gpu_util.run_case(case, flags=_lib.FLAG_SPLIT_BF16, need_grads=False)
        assert gu.relerr(resb["z_out"], ex["z_out"]) <= TIGHT_Z
        for k, e in _grad_errors(case, res).items():
            assert e <= (TOL_DZ0 if k == "dz0" else TOL_DTHETA), (nl, C, k, e)


@pytest.mark.parametrize("shape", [(40, 32, 32, 3), (21, 32, 32, 2), (33, 20, 15, 1), (24, 32, 32, 4)])
@pytest.mark.parametrize("interp,method,seq", [("linear", "rk4", True), ("cubic", "midpoint", False), ("linear", "euler", False)])
def test_forward_only_kernel_set_for_up_to_40_channels(shape, interp, method, seq, gpu_lib):
    """Round 5: 20 < C <= 40 at H, HH <= 32 runs its FORWARD on the register-resident kernel (zero-padded to
    (32, 32, 40): `ncde_fwd_fast_bf3<H32,HH32,C40,..>`), the backward on the batch-tiled family: forward against the oracle (split-fp16
    default, split-bf16, fp32-input MFMA, and against the batch-tiled forward), the recording forward of adjoint=False bit-identical, and
    both kinds of gradient end to end."""
    import gpu_util
    from ncde_amd import _lib
    C, H, HH, nl = shape
    case = _seeded_case(interp, method, seq, B=37, L=7, C=C, H=H, HH=HH, nl=nl, seed=1300 + 7 * C + H)
    ex = case["expect"]
    res = gpu_util.run_case(case)
    assert res["kernels"][0].startswith("ncde_fwd_fast_bf3<H32,HH32,C40") and res["kernels"][1].startswith("ncde_adj_tiled"), res["kernels"]
    assert gu.relerr(res["z_out"], ex["z_out"]) <= TIGHT_Z, gu.relerr(res["z_out"], ex["z_out"])
    for k, e in _grad_errors(case, res).items():
        assert e <= (TOL_DZ0 if k == "dz0" else TOL_DTHETA), ("end-to-end", k, e)
    resd = gpu_util.run_case(case, adjoint=False)      # the recording forward writes the record the batch-tiled discrete backward reads
    assert np.array_equal(resd["z_out"], res["z_out"])
    for k, e in _grad_errors(case, resd, "bp_").items():
        assert e <= (TOL_DZ0 if k == "dz0" else TOL_DTHETA), ("discrete end-to-end", k, e)
    rbf = gpu_util.run_case(case, flags=_lib.FLAG_SPLIT_BF16, need_grads=False)
    assert "bf16x3" in rbf["kernels"][0] and gu.relerr(rbf["z_out"], ex["z_out"]) <= TIGHT_Z, rbf["kernels"]
    r32 = gpu_util.run_case(case, flags=_lib.FLAG_FP32_MFMA, need_grads=False)
    assert r32["kernels"][0].startswith("ncde_fwd_fast<H32,HH32,C40") and gu.relerr(r32["z_out"], ex["z_out"]) <= TIGHT_Z, r32["kernels"]
    rt = gpu_util.run_case(case, flags=_lib.FLAG_FORCE_TILED, need_grads=False)
    assert rt["kernels"][0].startswith("ncde_fwd_tiled") and gu.relerr(rt["z_out"], res["z_out"]) <= TIGHT_Z


@pytest.mark.parametrize("shape", [(5, 16, 15, 3), (20, 32, 15, 1), (3, 7, 15, 2), (17, 30, 32, 3),      # -> (32, 32, 8 | 20 | 4 | 20)
                                   (4, 32, 15, 3), (8, 32, 32, 4), (12, 20, 15, 1), (9, 32, 32, 3),       # -> (32, 32, 4 | 8 | 12 | 12)
                                   (4, 47, 32, 3), (3, 64, 15, 2), (2, 40, 64, 1)])                           # -> (64, 64, 4)
@pytest.mark.parametrize("interp,method,seq", [("linear", "rk4", True), ("cubic", "midpoint", False)])
def test_small_shapes_zero_padded_onto_the_specialised_kernels(shape, interp, method, seq, gpu_lib):
    """A model within (H, HH, C) <= (32, 32, 20) or (64, 64, 4) -- the reference's default hidden_hidden_dim = 15 with a small state --
    is zero-padded by the library onto the specialised kernel sets (they take the real row width of z / gradients / stage record and
    the real channel count of the coefficient tensor): forward, continuous adjoint and exact discrete backward against the oracle,
    ragged batch, and the same results as the batch-tiled family the shape would otherwise run on."""
    import gpu_util
    from ncde_amd import _lib
    C, H, HH, nl = shape
    case = _seeded_case(interp, method, seq, B=37, L=7, C=C, H=H, HH=HH, nl=nl, seed=1200 + 7 * C + H)
    ex = case["expect"]
    res = gpu_util.run_case(case)
    big = H > 32 or HH > 32
    cset = 4 if C <= 4 else (8 if C <= 8 else (12 if C <= 12 else 20))      # round 6: few channels have their own instantiations
    if nl == 4 and interp == "cubic" and not big:      # (the cubic path's LDS plan with four layers: C = 20 does not fit, the smaller sets may)
        cset = None
    want = ("ncde_fwd_fast_bf3<H64,HH64,C4", "ncde_adj_h64<H64") if big else ("ncde_fwd_fast_bf3<H32,HH32,C%s" % (cset or ""), "ncde_adj_fast3<H32,HH32,C%s" % ("%d,NL%d" % (cset, nl) if cset else ""))
    assert res["kernels"][0].startswith(want[0]) and res["kernels"][1].startswith(want[1]) and "discrete" in res["kernels"][2], res["kernels"]
    assert gu.relerr(res["z_out"], ex["z_out"]) <= TIGHT_Z
    for k, e in _grad_errors(case, res).items():
        assert e <= (TOL_DZ0 if k == "dz0" else TOL_DTHETA), ("end-to-end", k, e)
    iso = gpu_util.run_adjoint_direct(case, ex["z_out"])
    for k, e in _grad_errors(case, iso).items():
        assert e <= TIGHT_G, ("adjoint on oracle z_out", res["kernels"][1], k, e)
    isod = gpu_util.run_adjoint_direct(case, ex["z_out"], stages=case["stage_record"])
    for k, e in _grad_errors(case, isod, "bp_").items():
        assert e <= TIGHT_G, ("discrete backward on the oracle's stage record", res["kernels"][2], k, e)
    resd = gpu_util.run_case(case, adjoint=False)      # the recording forward writes the record with the REAL row width
    assert np.array_equal(resd["z_out"], res["z_out"])
    for k, e in _grad_errors(case, resd, "bp_").items():
        assert e <= (TOL_DZ0 if k == "dz0" else TOL_DTHETA), ("discrete end-to-end", k, e)
    r32 = gpu_util.run_case(case, flags=_lib.FLAG_FP32_MFMA, need_grads=False)
    assert gu.relerr(r32["z_out"], ex["z_out"]) <= TIGHT_Z, r32["kernels"]
    rt = gpu_util.run_case(case, flags=_lib.FLAG_FORCE_TILED, need_grads=False)
    assert rt["kernels"][0].startswith("ncde_fwd_tiled") and gu.relerr(rt["z_out"], res["z_out"]) <= TIGHT_Z


@pytest.mark.parametrize("nl", [1, 2, 4])
@pytest.mark.parametrize("interp,method,seq", [("linear", "rk4", False), ("cubic", "midpoint", True), ("linear", "euler", True)])
def test_h32_chain_grad_adjoint_other_layer_counts(nl, interp, method, seq, gpu_lib):
    """ncde_adj_fast3 instantiated for nl = 1, 2 (both control paths) and nl = 4 (linear path; the cubic one's LDS plan does not fit and
    stays on the batch-tiled family): continuous adjoint and exact discrete backward on the oracle's z / stage record, default
    (forward side split-fp16) and all-split-bf16, bit-reproducible; range-fault re-execution as for nl = 3."""
    import gpu_util
    from ncde_amd import _lib
    case = _seeded_case(interp, method, seq, B=37, L=7, C=20, H=32, HH=32, nl=nl, seed=900 + nl)
    ex = case["expect"]
    names = gpu_util.kernel_names(case)
    if nl == 4 and interp == "cubic":
        assert names[1].startswith("ncde_adj_tiled"), names
        return
    assert names[1].startswith("ncde_adj_fast3<H32,HH32,C20,NL%d" % nl) and "discrete" in names[2] and ("NL%d" % nl) in names[2], names
    for fl in (_lib.FLAG_AUTO, _lib.FLAG_SPLIT_BF16):
        iso = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=fl)
        for k, e in _grad_errors(case, iso).items():
            assert e <= TIGHT_G, ("continuous", nl, fl, k, e)
        isod = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=fl, stages=case["stage_record"])
        for k, e in _grad_errors(case, isod, "bp_").items():
            assert e <= TIGHT_G, ("discrete", nl, fl, k, e)
    again = gpu_util.run_adjoint_direct(case, ex["z_out"])
    first = gpu_util.run_adjoint_direct(case, ex["z_out"])
    assert np.array_equal(again["dz0"], first["dz0"]) and all(np.array_equal(again["grads"][k], first["grads"][k]) for k in first["grads"])
    z = ex["z_out"].copy()
    z[21] *= 4.0e5      # one sample of the middle tile leaves the fp16 range: that tile comes from the split-bf16 instance
    r16 = gpu_util.run_adjoint_direct(case, z)
    rbf = gpu_util.run_adjoint_direct(case, z, flags=_lib.FLAG_SPLIT_BF16)
    assert np.isfinite(r16["dz0"]).all() and np.array_equal(r16["dz0"][16:32], rbf["dz0"][16:32])
    assert np.array_equal(r16["dz0"][:16], first["dz0"][:16]) and np.array_equal(r16["dz0"][32:], first["dz0"][32:])


def test_split_fp16_forward_is_reproducible_under_repetition(gpu_lib):
    """Ten launches of the default (split-fp16 + re-execution launch) forward kernels: bit-identical."""
    import gpu_util
    for shape in FAST_SHAPES:
        C, H, HH, nl = shape
        case = _seeded_case("cubic", "rk4", True, B=37, L=7, C=C, H=H, HH=HH, nl=nl, seed=312)
        first = gpu_util.run_case(case, need_grads=False)["z_out"]
        assert gu.relerr(first, case["expect"]["z_out"]) <= TIGHT_Z
        for _ in range(9):
            assert np.array_equal(gpu_util.run_case(case, need_grads=False)["z_out"], first)


@pytest.mark.parametrize("interp,method", [("cubic", "midpoint"), ("linear", "rk4"), ("cubic", "euler")])
def test_decoupled_adjoint_hand_offs_under_repetition(interp, method, gpu_lib):
    """ncde_adj_fast4 (NCDE_FLAG_ADJOINT_V4) hands data between its y waves and cotangent waves through LDS flags (t blocks, dP tiles coming
    back, reduction partials):
    ten launches of the continuous adjoint and of the discrete backward on the oracle's z, ragged two-workgroup batch, must be
    bit-identical and within the tight tolerance.  (cubic + midpoint was the combination that exposed a timing-dependent
    failure of a stage-weight-dependent control flow during development.)"""
    import gpu_util
    from ncde_amd import _lib
    case = _seeded_case(interp, method, False, B=21, L=5, C=20, H=32, HH=32, nl=3, seed=177)
    ex = case["expect"]
    for kw, pre in (({}, ""), ({"stages": case["stage_record"]}, "bp_")):
        first = None
        for _ in range(10):
            iso = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=_lib.FLAG_ADJOINT_V4, **kw)
            for k, e in _grad_errors(case, iso, pre).items():
                assert e <= TIGHT_G, (interp, method, pre, k, e)
            if first is None:
                first = iso
            else:
                assert np.array_equal(first["dz0"], iso["dz0"]) and all(np.array_equal(first["grads"][k], iso["grads"][k]) for k in iso["grads"])


@pytest.mark.parametrize("shape", [(80, 128, 128, 3), (8, 48, 64, 2), (4, 16, 32, 1), (16, 64, 64, 2)])      # the last: four row tiles per wave in pass B,
                                                                                                                # resident hidden fragments (RESH / RES = 2)
@pytest.mark.parametrize("interp,method,seq", [("linear", "rk4", False), ("cubic", "midpoint", True), ("linear", "euler", True)])
def test_tiled_family_vs_oracle(shape, interp, method, seq, gpu_lib):
    """The batch-tiled (large-hidden) family: every sample-tile count NS, ragged batch, against the oracle and,
    bit for bit, against the generic family's operation order is NOT required -- tolerance as everywhere else."""
    import gpu_util
    from ncde_amd import _lib
    C, H, HH, nl = shape
    case = _seeded_case(interp, method, seq, B=37, L=7, C=C, H=H, HH=HH, nl=nl, seed=300 + C)
    ex = case["expect"]
    for flag, ns in ((0, None), (0x1000, 1), (0x2000, 2), (0x4000, 4)):
        res = gpu_util.run_case(case, flags=flag, need_grads=False)
        # (without a batch-tiled knob a shape within (32, 32, 20) / (64, 64, 4) is zero-padded onto the specialised kernels instead)
        assert res["kernels"][0].startswith("ncde_fwd_tiled" if (ns or H > 64 or C > 20 or (H > 32 and C > 4)) else "ncde_fwd_"), res["kernels"]
        if ns:
            assert res["kernels"][0] in ("ncde_fwd_tiled<NS%d>" % ns, "ncde_fwd_tiled<NS1,fp16x2>"), res["kernels"]
            if ns == 1 and HH % 32 == 0:      # one sample tile, last hidden width a multiple of 32: the split-fp16 output tiles (default) ...
                assert res["kernels"][0] == "ncde_fwd_tiled<NS1,fp16x2>"
                rbf = gpu_util.run_case(case, flags=flag | _lib.FLAG_SPLIT_BF16, need_grads=False)     # ... the split-bf16 ones ...
                assert rbf["kernels"][0] == "ncde_fwd_tiled<NS1,bf16>" and gu.relerr(rbf["z_out"], ex["z_out"]) <= TIGHT_Z
                r32 = gpu_util.run_case(case, flags=flag | _lib.FLAG_FP32_MFMA, need_grads=False)      # ... and the fp32-input MFMA ones
                assert r32["kernels"][0] == "ncde_fwd_tiled<NS1>" and gu.relerr(r32["z_out"], ex["z_out"]) <= TIGHT_Z
        assert gu.relerr(res["z_out"], ex["z_out"]) <= TIGHT_Z, (flag, gu.relerr(res["z_out"], ex["z_out"]))
    FT = 0x8000                                                  # force the tiled backward also where generic is preferred
    res = gpu_util.run_case(case, flags=FT)                      # sweep (pass A) + output-layer gradient pass (pass B)
    assert res["kernels"][1].startswith("ncde_adj_tiled") and "discrete" in res["kernels"][2], res["kernels"]
    for k, e in _grad_errors(case, res).items():
        assert e <= (TOL_DZ0 if k == "dz0" else TOL_DTHETA), ("end-to-end", k, e)
    iso = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=FT)
    for k, e in _grad_errors(case, iso).items():
        assert e <= TIGHT_G, ("tiled adjoint on oracle z_out", k, e)
    again = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=FT)
    assert np.array_equal(again["dz0"], iso["dz0"]) and all(np.array_equal(again["grads"][k], iso["grads"][k]) for k in iso["grads"])
    resd = gpu_util.run_case(case, flags=FT, adjoint=False)      # stage record written by the tiled forward
    for k, e in _grad_errors(case, resd, "bp_").items():
        assert e <= (TOL_DZ0 if k == "dz0" else TOL_DTHETA), ("discrete end-to-end", k, e)
    isod = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=FT, stages=case["stage_record"])
    for k, e in _grad_errors(case, isod, "bp_").items():
        assert e <= TIGHT_G, ("tiled discrete backward on the oracle's stage record", k, e)
    # the same two backward passes on fp32-input MFMA and fp32 records (what NCDE_FLAG_FP32_MFMA selects; the default above ran the
    # split-bf16 records + ncde_dwo_pair wherever the last hidden width is a multiple of 32)
    iso32 = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=FT | _lib.FLAG_FP32_MFMA)
    for k, e in _grad_errors(case, iso32).items():
        assert e <= TIGHT_G, ("tiled adjoint (fp32 MFMA) on oracle z_out", k, e)
    isod32 = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=FT | _lib.FLAG_FP32_MFMA, stages=case["stage_record"])
    for k, e in _grad_errors(case, isod32, "bp_").items():
        assert e <= TIGHT_G, ("tiled discrete backward (fp32 MFMA) on the oracle's stage record", k, e)


@pytest.mark.parametrize("shape,kind", [((80, 128, 128, 3), "original"), ((8, 48, 64, 2), "original"), ((12, 32, 32, 3), "minimal"), ((5, 47, 93, 2), "original")])
def test_tiled_split_fp16_forward_range_fault_is_reexecuted(shape, kind, gpu_lib):
    """The batch-tiled forward multiplies its output tiles in 2-way split-fp16 and speculates on the fp16 range of x_L (and of the
    weights); a sample tile whose x_L leaves it is re-executed by the split-bf16 instantiation.  Three tiles (B = 37), one sample
    of the middle tile with a state large enough to overflow: its tile's rows are bit-identical to the split-bf16 run, the other
    tiles' rows to the run without the outlier, everything agrees with the oracle.  The last shape reaches the kernels through the
    library's zero-padding (C 5 -> 8, H 47 -> 48, HH 93 -> 128)."""
    import gpu_util
    import ncde_oracle as orc
    from ncde_amd import _lib
    C, H, HH, nl = shape
    for seq in (False, True):
        case = _seeded_case("linear", "rk4", seq, B=37, L=6, C=C, H=H, HH=HH, nl=nl, seed=411, kind=kind)
        plain = gpu_util.run_case(case, need_grads=False)
        assert "fp16x2" in plain["kernels"][0] and plain["kernels"][0].startswith("ncde_fwd_tiled"), plain["kernels"]
        big = dict(case)
        big["z0"] = case["z0"].copy()
        big["z0"][21] *= 1.0e8      # (only x_L is split here -- the hidden layers run on fp32-input MFMA --, and it must leave the fp16 range)
        zo = orc.solve_forward(orc.Control(big["coeffs"], "linear"), gu.oracle_field(case), big["z0"], "rk4", seq).numpy()
        r16 = gpu_util.run_case(big, need_grads=False)
        rbf = gpu_util.run_case(big, flags=_lib.FLAG_SPLIT_BF16, need_grads=False)
        assert np.isfinite(r16["z_out"]).all()
        assert np.array_equal(r16["z_out"][16:32], rbf["z_out"][16:32])                       # re-executed tile
        assert np.array_equal(r16["z_out"][:16], plain["z_out"][:16]) and np.array_equal(r16["z_out"][32:], plain["z_out"][32:])
        scale = np.abs(zo).max(axis=tuple(range(1, zo.ndim)), keepdims=True)
        assert float((np.abs(r16["z_out"] - zo) / scale).max()) <= TIGHT_Z


@pytest.mark.parametrize("shape,gated", [((80, 128, 128, 3), False), ((8, 48, 64, 2), False), ((20, 32, 32, 3), True),
                                         ((8, 32, 128, 2), True)])      # the last: gated heads of 128 columns = one pair-kernel pass per head
@pytest.mark.parametrize("interp,method,seq", [("linear", "rk4", False), ("cubic", "midpoint", True)])
def test_tiled_backward_time_windows(shape, gated, interp, method, seq, gpu_lib):
    """The batch-tiled backward keeps its per-stage records for a WINDOW of steps only (workspace O(B H W), not O(B H T)):
    forcing windows of one / a few steps (NCDE_FLAG_TILED_WINDOW_STEPS) must reproduce the single-window result -- bit for bit for
    dL/dz0 and the hidden-layer gradients (the carried (y, a) and the hidden-layer partial are exact hand-overs), to summation
    order (2e-6) for the head gradients (pass B adds one partial per window instead of running one long accumulation) -- for
    the continuous adjoint and the exact discrete backward; the workspace query shrinks accordingly."""
    import ctypes
    import gpu_util
    from ncde_amd import _lib, solver
    C, H, HH, nl = shape
    case = _seeded_case(interp, method, seq, B=37, L=7, C=C, H=H, HH=HH, nl=nl, seed=900 + C, kind="minimal" if gated else "original")
    FT = 0x8000
    ref = gpu_util.run_adjoint_direct(case, case["expect"]["z_out"], flags=FT)
    refd = gpu_util.run_adjoint_direct(case, case["expect"]["z_out"], flags=FT, stages=case["stage_record"])
    coeffs = torch.from_numpy(case["coeffs"]).cuda()
    func = gpu_util.case_field(case, "cuda")
    p = solver.build_problem(coeffs, interp, torch.from_numpy(case["z0"]).cuda(), func.fused_spec(), method, int(seq), FT)
    full = _lib.lib().ncde_workspace_bytes(ctypes.byref(p), 1)
    S = {"rk4": 4, "midpoint": 2, "euler": 1}[method]
    n_steps = case["coeffs"].shape[1] - 1 + (interp == "cubic")
    per_step_mb = S * 3 * (2 * HH + H + C) * 16 * 4 / 2 ** 20          # 3 sample tiles of 16
    for steps in (1, 5):
        FW = FT | (steps << 16)                      # NCDE_FLAG_TILED_WINDOW_STEPS(steps)
        p.flags = FW
        small = _lib.lib().ncde_workspace_bytes(ctypes.byref(p), 1)
        assert small < full and full - small >= (n_steps - steps - 1) * per_step_mb * 2 ** 20 * 0.99
        got = gpu_util.run_adjoint_direct(case, case["expect"]["z_out"], flags=FW)
        gotd = gpu_util.run_adjoint_direct(case, case["expect"]["z_out"], flags=FW, stages=case["stage_record"])
        for g_, r_ in ((got, ref), (gotd, refd)):
            assert np.array_equal(g_["dz0"], r_["dz0"])
            for k in r_["grads"]:
                if k in ("Wo", "bo", "Wg", "bg"):
                    assert gu.relerr(g_["grads"][k], r_["grads"][k]) <= 2e-6, (steps, k)      # summation order only (one partial per window)
                else:
                    assert np.array_equal(g_["grads"][k], r_["grads"][k]), (steps, k)


@pytest.mark.parametrize("shape", [(8, 32, 32, 2), (20, 32, 32, 3), (4, 64, 64, 3), (80, 128, 128, 2)])
@pytest.mark.parametrize("interp,method,seq", [("linear", "rk4", False), ("cubic", "midpoint", True)])
def test_tiled_minimal_gated_vs_oracle(shape, interp, method, seq, gpu_lib):
    """The minimal-gated field on the batch-tiled family (second head in the forward, in the sweep's VJP, and one
    gradient pass per head): forward, continuous adjoint and exact discrete backward against the oracle, and against the
    variant kernels on the generic structure."""
    import gpu_util
    C, H, HH, nl = shape
    case = _seeded_case(interp, method, seq, B=37, L=6, C=C, H=H, HH=HH, nl=nl, seed=700 + C, kind="minimal")
    ex = case["expect"]
    res = gpu_util.run_case(case)
    assert res["kernels"][0].startswith("ncde_fwd_tiled") and "gated" in res["kernels"][0], res["kernels"]
    assert "gated" in res["kernels"][1] and "gated" in res["kernels"][2], res["kernels"]
    assert gu.relerr(res["z_out"], ex["z_out"]) <= TIGHT_Z
    gen = gpu_util.run_case(case, flags=1, need_grads=False)
    assert gen["kernels"][0] == "ncde_fwd_variant" and gu.relerr(gen["z_out"], ex["z_out"]) <= TIGHT_Z
    for flag in (0x1000, 0x2000, 0x4000):
        rn = gpu_util.run_case(case, flags=flag, need_grads=False)
        assert gu.relerr(rn["z_out"], ex["z_out"]) <= TIGHT_Z, (flag, rn["kernels"])
    iso = gpu_util.run_adjoint_direct(case, ex["z_out"])
    for k, e in _grad_errors(case, iso).items():
        assert e <= TIGHT_G, ("gated tiled adjoint", k, e)
    again = gpu_util.run_adjoint_direct(case, ex["z_out"])
    assert all(np.array_equal(again["grads"][k], iso["grads"][k]) for k in iso["grads"])
    isod = gpu_util.run_adjoint_direct(case, ex["z_out"], stages=case["stage_record"])
    for k, e in _grad_errors(case, isod, "bp_").items():
        assert e <= TIGHT_G, ("gated tiled discrete backward", k, e)
    for k, e in _grad_errors(case, res).items():
        assert e <= (TOL_DZ0 if k == "dz0" else TOL_DTHETA), ("end-to-end", k, e)


_SWEEP = [  # (B, L, C, H, HH, nl, interp, method, seq)  -- whatever family the dispatcher picks for each
    (5, 4, 3, 7, 15, 1, "linear", "rk4", False),         # reference defaults: hidden_hidden_dim = 15 (odd widths: zero-padded
    (33, 6, 6, 10, 15, 3, "cubic", "midpoint", True),    #   into the batch-tiled family by the library)
    (37, 5, 5, 47, 93, 3, "linear", "rk4", True),        # odd everything inside the reference's hyper-parameter ranges
    (21, 6, 21, 96, 15, 2, "cubic", "midpoint", False),  # C = 21 -> 24, H = 96, HH = 15 -> 16
    (18, 4, 10, 32, 47, 4, "linear", "euler", False),    # HH = 47 -> 64, four layers
    (9, 7, 7, 128, 100, 3, "cubic", "rk4", True),        # HH = 100 -> 128
    (19, 9, 8, 32, 32, 2, "linear", "euler", True),      # multiples of 16/4 without a specialised kernel: tiled
    (40, 5, 12, 48, 16, 3, "cubic", "rk4", False),
    (17, 7, 20, 32, 32, 4, "linear", "rk4", True),       # cfg2 dims but nl = 4: no fast adjoint instantiation
    (70, 3, 4, 64, 64, 3, "linear", "midpoint", False),  # cfg4 dims, linear control
    (16, 2, 16, 16, 128, 2, "linear", "rk4", False),     # a single step, wide hidden layers
    (3, 12, 2, 16, 16, 1, "cubic", "euler", True),       # time + one channel
    # round 5: beyond 80 channels and beyond 128 state units on the batch-tiled backward
    (22, 5, 100, 64, 64, 3, "linear", "rk4", True),      # C = 100: dX/dt read in the sweep's bookkeeping phase (B = 21 has one knife-edge ReLU sample)
    (18, 3, 84, 48, 32, 2, "cubic", "euler", False),     # C = 84, cubic control
    (19, 4, 20, 160, 128, 3, "linear", "rk4", False),    # H = 160: the one-wave-per-SIMD instantiation of the sweep (BIGH)
    (11, 3, 7, 256, 128, 2, "cubic", "midpoint", True),  # H = 256, C = 7 -> 8 (zero-padded), sequence outputs
    (35, 2, 12, 208, 64, 3, "linear", "rk4", True),      # H = 208, last width 64
    # hidden widths beyond 128 (zero-padded to 256 in the backward: the W16 mode of the wide sweep, fp32 records, ncde_dwo_tiled<16>)
    (19, 4, 20, 196, 196, 2, "linear", "rk4", False),    # the reference's largest hidden_hidden_dim (configurations.json5:35)
    (13, 3, 7, 256, 160, 3, "cubic", "midpoint", True),  # H = 256, HH = 160
    (34, 2, 5, 64, 256, 2, "linear", "euler", False),    # H = 64 under a 256-wide stack (no padding)
    (17, 3, 20, 196, 196, 3, "linear", "rk4", True),     # three layers, sequence outputs
    (20, 3, 4, 160, 15, 1, "linear", "midpoint", False), # H = 160 over the reference's default hidden_hidden_dim = 15 (-> 16): last width 16 on the wide sweep
]


@pytest.mark.parametrize("cfg", _SWEEP, ids=lambda c: "B%d_L%d_C%d_H%d_HH%d_nl%d_%s_%s_%s" % (c[:6] + (c[6], c[7], "seq" if c[8] else "final")))
def test_shape_sweep_every_family_vs_oracle(cfg, gpu_lib):
    """Shapes the dispatcher routes to different kernel families (odd widths, no specialised instantiation, tiny and
    wide layers, one channel): forward, continuous adjoint and exact discrete backward against the oracle, and the
    dispatched family against the generic one."""
    import gpu_util
    B, L, C, H, HH, nl, interp, method, seq = cfg
    case = _seeded_case(interp, method, seq, B=B, L=L, C=C, H=H, HH=HH, nl=nl, seed=500 + B)
    ex = case["expect"]
    res = gpu_util.run_case(case)
    assert gu.relerr(res["z_out"], ex["z_out"]) <= TIGHT_Z, res["kernels"]
    # no shape with dims <= 128 runs on the generic family unless asked to: odd shapes are zero-padded
    assert not any("generic" in k for k in res["kernels"]), res["kernels"]
    gen = gpu_util.run_case(case, flags=1, need_grads=False)
    assert "generic" in gen["kernels"][0]
    assert gu.relerr(res["z_out"], gen["z_out"]) <= TIGHT_Z
    iso = gpu_util.run_adjoint_direct(case, ex["z_out"])
    for k, e in _grad_errors(case, iso).items():
        assert e <= TIGHT_G, ("adjoint", res["kernels"][1], k, e)
    isod = gpu_util.run_adjoint_direct(case, ex["z_out"], stages=case["stage_record"])
    for k, e in _grad_errors(case, isod, "bp_").items():
        assert e <= TIGHT_G, ("discrete backward", res["kernels"][2], k, e)
    resd = gpu_util.run_case(case, adjoint=False)
    assert np.array_equal(resd["z_out"], res["z_out"])
    for k, e in _grad_errors(case, resd, "bp_").items():
        assert e <= (TOL_DZ0 if k == "dz0" else TOL_DTHETA), ("discrete end-to-end", k, e)


def test_gpu_coefficient_builders_match_reference(gpu_lib):
    """ncde_prepare_linear / ncde_prepare_cubic against the reference's builders
    (golden g8) -- bit-exact for the rectilinear preparation and the spline, 1e-6 for the NaN fill -- and at
    BASELINE size against the host mirrors; then the prepared coefficients drive the solve end to end."""
    import os
    import ncde_amd
    f = np.load(os.path.join(gu.GOLD, "g8_coeffs.npz"))
    xm = torch.from_numpy(f["x_missing"]).cuda()
    assert gu.relerr(ncde_amd.linear_interpolation_coeffs(xm).cpu().numpy(), f["linear"]) <= 1e-6
    assert np.array_equal(ncde_amd.linear_interpolation_coeffs(xm, rectilinear=0).cpu().numpy(), f["rectilinear"])
    xc = torch.from_numpy(f["x_clean"]).cuda()
    assert np.array_equal(ncde_amd.natural_cubic_coeffs(xc).cpu().numpy(), f["cubic"])
    assert np.array_equal(ncde_amd.natural_cubic_coeffs(xc[:, :2].contiguous()).cpu().numpy(), f["cubic_len2"])
    # missing values: ends filled, spline through the observed knots, re-expanded per unit interval -- bit-exact
    assert np.array_equal(ncde_amd.natural_cubic_coeffs(xm).cpu().numpy(), f["cubic_missing"])
    xr2 = gu.data.synthetic_series(64, 40, 7, missing=0.5, seed=77)
    xr2[:, 0, 1:3] = np.nan
    xr2[3, :, 4] = np.nan
    assert np.array_equal(ncde_amd.natural_cubic_coeffs(torch.from_numpy(xr2).cuda()).cpu().numpy(), gu.data.natural_cubic_coeffs(xr2))
    # extra batch dimensions, as torchcde allows
    assert np.array_equal(ncde_amd.natural_cubic_coeffs(torch.stack([xc, xc])).cpu().numpy(), np.stack([f["cubic"], f["cubic"]]))
    with pytest.raises(AssertionError):
        bad = xm.clone(); bad[0, 1, 0] = float("nan")
        ncde_amd.linear_interpolation_coeffs(bad, rectilinear=0)
    # BASELINE cfg2 / cfg4 sizes against the host mirrors (themselves pinned to the reference)
    x2 = gu.data.synthetic_series(512, 200, 19, missing=0.3, seed=1234)
    got = ncde_amd.linear_interpolation_coeffs(torch.from_numpy(x2).cuda(), rectilinear=0).cpu().numpy()
    assert np.array_equal(got, gu.data.make_rectilinear_coeffs(512, 200, 19, missing=0.3, seed=1234))
    x4 = gu.data.synthetic_series(256, 182, 3, seed=1234)
    got = ncde_amd.natural_cubic_coeffs(torch.from_numpy(x4).cuda()).cpu().numpy()
    assert np.array_equal(got, gu.data.natural_cubic_coeffs(x4))


def test_gpu_coefficient_builders_on_a_user_time_grid(gpu_lib):
    """linear_interpolation_coeffs(x, t=...) / natural_cubic_coeffs(x, t=...) (interpolation_linear.py:131-180, interpolation_cubic.py:56-165):
    goldens = the reference's builders on an irregular grid.  Linear: the grid enters the interior-gap fill; cubic: the non-uniform
    natural spline, with and without missing values; and the result feeds cdeint on the same knots (g11's user-grid path)."""
    import os
    import ncde_amd
    f = np.load(os.path.join(gu.GOLD, "g8_coeffs_user_grid.npz"))
    t = torch.from_numpy(f["t"]).cuda()
    xm, xc = torch.from_numpy(f["x_missing"]).cuda(), torch.from_numpy(f["x_clean"]).cuda()
    assert gu.relerr(ncde_amd.linear_interpolation_coeffs(xm, t=t).cpu().numpy(), f["linear"]) <= 1e-6
    got = ncde_amd.natural_cubic_coeffs(xc, t=t).cpu().numpy()
    assert gu.relerr(got, f["cubic"]) <= 2e-6, gu.relerr(got, f["cubic"])
    assert np.allclose(ncde_amd.natural_cubic_coeffs(xc[:, :2].contiguous(), t=t[:2]).cpu().numpy(), f["cubic_len2"], rtol=1e-6, atol=1e-7)
    gotm = ncde_amd.natural_cubic_coeffs(xm, t=t).cpu().numpy()
    assert gu.relerr(gotm, f["cubic_missing"]) <= 2e-6, gu.relerr(gotm, f["cubic_missing"])
    # the reference's argument checks (misc.py:70-100)
    with pytest.raises(ValueError, match="monotonically increasing"):
        ncde_amd.natural_cubic_coeffs(xc, t=t.flip(0))
    with pytest.raises(ValueError, match="time dimension"):
        ncde_amd.linear_interpolation_coeffs(xm, t=t[:-1])
    # a control built here on the user grid goes straight into cdeint on that grid
    X = ncde_amd.NaturalCubicSpline(ncde_amd.natural_cubic_coeffs(xc, t=t), t=t)
    func = ncde_amd.OriginalVectorField(xc.shape[-1], 8, 16, 2).cuda()
    z = ncde_amd.cdeint(X, func, torch.zeros(xc.shape[0], 8, device="cuda"), X.interval, method="rk4", options={"step_size": 0.5})
    assert torch.isfinite(z).all()


TIMES_CASES = ["g11_times_rk4_half", "g11_times_midpoint_third", "g11_knots_rk4", "g11_knots_cubic_euler",
               "g11_knots_interval_rk4", "g11_times_f64_rk4", "g11_times_cubic_rk4_ragged"]


@pytest.mark.parametrize("name", TIMES_CASES)
def test_general_time_axis_matches_reference_golden(name, gpu_lib):
    """The rest of the cdeint call surface (goldens g11, produced by the imported reference): arbitrary increasing output
    times (outputs linearly interpolated between grid states, solvers.py:103-117, 166-172), step_size != 1
    (solvers.py:78-87), user knot grids, fp64 times -- forward, continuous adjoint (one reverse solve per output
    interval, adjoint.py:116-133) and the exact discrete backward, through the plan-driven generic kernels."""
    import json
    import os
    import gpu_util
    f = dict(np.load(os.path.join(gu.GOLD, name + ".npz")))
    m = json.loads(str(f["meta"]))
    res = gpu_util.run_times_case(f, m, adjoint=True)
    assert res["z_out"].shape == f["z_out"].shape
    assert gu.relerr(res["z_out"], f["z_out"]) <= TIGHT_Z
    assert gu.relerr(res["dz0"], f["dz0"]) <= E2E_G
    for pname in m["param_names"]:
        assert gu.relerr(res["grads"][pname], f["d" + pname]) <= E2E_G, pname
    assert res["nfe"] == m["nfe"]                      # the reference's own counter on the same axis (base.py:90)
    resd = gpu_util.run_times_case(f, m, adjoint=False)
    assert np.array_equal(resd["z_out"], res["z_out"])
    assert gu.relerr(resd["dz0"], f["bp_dz0"]) <= E2E_G
    for pname in m["param_names"]:
        assert gu.relerr(resd["grads"][pname], f["bp_d" + pname]) <= E2E_G, pname


def test_user_knot_grid_with_the_controls_own_time_tensors(gpu_lib):
    """`t = X.interval` of a control built on a user knot grid (golden g11_knots_interval_rk4: the reference on exactly that
    call): the tagged-tensor shortcut of `_time_mode` must not route it to the default-axis kernels."""
    import json
    import os
    import gpu_util
    f = dict(np.load(os.path.join(gu.GOLD, "g11_knots_interval_rk4.npz")))
    m = json.loads(str(f["meta"]))
    res = gpu_util.run_times_case(f, m, adjoint=True, tagged="interval")
    assert gu.relerr(res["z_out"], f["z_out"]) <= TIGHT_Z
    assert gu.relerr(res["dz0"], f["dz0"]) <= E2E_G
    for pname in m["param_names"]:
        assert gu.relerr(res["grads"][pname], f["d" + pname]) <= E2E_G, pname
    plain = gpu_util.run_times_case(f, m, adjoint=True)
    assert np.array_equal(plain["z_out"], res["z_out"])
    # every knot of the user grid as output (X.grid_points, tagged) == the same values as a plain tensor
    f2 = dict(f)
    f2["t_out"] = f["knots"]
    f2["grad_out"] = np.ones((f["z0"].shape[0], f["knots"].shape[0], f["z0"].shape[1]), dtype=np.float32)
    a = gpu_util.run_times_case(f2, m, adjoint=True, tagged="knots")
    b = gpu_util.run_times_case(f2, m, adjoint=True)
    assert np.array_equal(a["z_out"], b["z_out"]) and np.array_equal(a["dz0"], b["dz0"])


@pytest.mark.parametrize("C,H,HH,nl,interp,method,step", [(8, 160, 32, 2, "linear", "rk4", 0.5),        # H = 160: the one-wave-per-SIMD sweep (BIGH), split records
                                                          (4, 48, 160, 2, "cubic", "midpoint", 0.4),     # HH = 160 -> 256: its W16 mode, fp32 records
                                                          (100, 32, 64, 3, "linear", "euler", 0.25),     # C = 100: dX/dt read in the bookkeeping phase
                                                          (7, 256, 196, 2, "cubic", "rk4", 0.75)])       # H = 256, HH = 196, C = 7 -> 8: everything padded
def test_general_time_axis_on_the_wide_sweeps_vs_oracle(C, H, HH, nl, interp, method, step, gpu_lib):
    """Round 5: the batch-tiled backward's new instantiations (H up to 256, hidden widths up to 256, any channel count) walk the time plan
    like the rest of the family: any output times / step size / user knot grid against the oracle's general-time functions -- forward,
    continuous adjoint (one reverse solve per output interval) and exact discrete backward, with one and with several time windows."""
    import gpu_util
    import ncde_oracle as orc
    B, L = 21, 7
    rng = np.random.RandomState(7)
    x = (gu.data.normal(51, B * L * C, stream=3).reshape(B, L, C) * 0.5).astype(np.float32)
    if interp == "linear":      # user knot grid, spacing 0.6 .. 1.4
        kn = np.cumsum(np.concatenate([[0.0], 0.6 + 0.8 * rng.rand(L - 1)])).astype(np.float32)
        x[:, :, 0] = kn[None, :]
        coeffs = x
    else:
        kn = np.arange(L, dtype=np.float32)
        x[:, :, 0] = kn[None, :]
        coeffs = gu.data.natural_cubic_coeffs(x)
    p = gu.data.make_field_weights(H, HH, C, seed=29)
    if nl == 1:
        p = {k: v for k, v in p.items() if k not in ("W1", "b1")}
    z0 = (gu.data.normal(53, B * H, stream=2).reshape(B, H) * 0.5).astype(np.float32)
    tout = np.array([kn[0], 0.5 * (kn[1] + kn[2]), kn[3], kn[4] + 0.05, kn[-1] - 0.125], np.float32)
    meta = {"kind": interp, "method": method, "step_size": step, "dims": {"nl": nl}}
    field = orc.Field.variant(p, H, C, nl, "original", "matmul")
    ctl = orc.Control(coeffs, interp, t=kn if interp == "linear" else None)
    z = orc.solve_forward_times(ctl, field, z0, tout, method, step)
    gout = (gu.data.normal(27, z.numel(), stream=1).reshape(z.shape) / 2.0).astype(np.float32)
    dz0, gp = orc.solve_adjoint_times(ctl, field, tout, z, gout, method, step)
    bdz0, bgp = orc.solve_discrete_backward_times(ctl, field, z0, tout, gout, method, step)
    g = {"coeffs": coeffs, "z0": z0, "t_out": tout, "grad_out": gout}
    if interp == "linear":
        g["knots"] = kn
    names = [n for n in ("W0", "b0", "W1", "b1", "Wo", "bo") if n in p]
    import warnings
    from ncde_amd import unfused
    unfused._WARNED.clear()
    for wflags in (0, 3 << 16):            # default record budget, then NCDE_FLAG_TILED_WINDOW_STEPS(3)
        with warnings.catch_warnings():      # (a shape without a fused backward would be routed to the unfused solver WITH a warning: not here)
            warnings.filterwarnings("error", message=".*unfused.*")
            res = gpu_util.run_times_case(g, meta, adjoint=True, kind="original", mode="matmul", params=p, flags=wflags)
            resd = gpu_util.run_times_case(g, meta, adjoint=False, kind="original", mode="matmul", params=p, flags=wflags)
        assert gu.relerr(res["z_out"], z) <= TIGHT_Z, gu.relerr(res["z_out"], z)
        assert gu.relerr(res["dz0"], dz0) <= E2E_G, gu.relerr(res["dz0"], dz0)
        for n_, g_ in zip(names, gp):
            assert gu.relerr(res["grads"][n_], g_) <= E2E_G, (n_, gu.relerr(res["grads"][n_], g_))
        assert gu.relerr(resd["dz0"], bdz0) <= E2E_G
        for n_, g_ in zip(names, bgp):
            assert gu.relerr(resd["grads"][n_], g_) <= E2E_G, ("discrete", n_, gu.relerr(resd["grads"][n_], g_))


@pytest.mark.parametrize("C,H,HH,nl,B,interp,method,step", [(20, 128, 128, 2, 128, "linear", "rk4", 0.5),       # one group of 8 members
                                                            (40, 64, 128, 3, 250, "cubic", "midpoint", 0.4)])   # two groups of 8, ragged last tile
def test_general_time_axis_on_the_cooperative_kernels_vs_oracle(C, H, HH, nl, B, interp, method, step, gpu_lib):
    """Round 6: the XCD-cooperative forward, sweep and ncde_dwo_h2 walk the time plan like the per-workgroup
    kernels of the family -- any output times / step size / user knot grid against the oracle's general-time functions: forward,
    continuous adjoint (one reverse solve per output interval), exact discrete backward, one and several time windows; and against the
    per-workgroup kernels (NCDE_FLAG_NO_COOP)."""
    import ctypes
    import gpu_util
    import ncde_amd
    import ncde_oracle as orc
    from ncde_amd import _lib, solver
    L = 7
    rng = np.random.RandomState(11)
    x = (gu.data.normal(71, B * L * C, stream=3).reshape(B, L, C) * 0.5).astype(np.float32)
    if interp == "linear":      # user knot grid, spacing 0.6 .. 1.4
        kn = np.cumsum(np.concatenate([[0.0], 0.6 + 0.8 * rng.rand(L - 1)])).astype(np.float32)
        x[:, :, 0] = kn[None, :]
        coeffs = x
    else:
        kn = np.arange(L, dtype=np.float32)
        x[:, :, 0] = kn[None, :]
        coeffs = gu.data.natural_cubic_coeffs(x)
    p = gu.data.make_field_weights(H, HH, C, seed=31)
    z0 = (gu.data.normal(73, B * H, stream=2).reshape(B, H) * 0.5).astype(np.float32)
    tout = np.array([kn[0], 0.5 * (kn[1] + kn[2]), kn[3], kn[4] + 0.05, kn[-1] - 0.125], np.float32)
    meta = {"kind": interp, "method": method, "step_size": step, "dims": {"nl": nl}}
    field = orc.Field.variant(p, H, C, nl, "original", "matmul")
    ctl = orc.Control(coeffs, interp, t=kn if interp == "linear" else None)
    z = orc.solve_forward_times(ctl, field, z0, tout, method, step)
    gout = (gu.data.normal(27, z.numel(), stream=1).reshape(z.shape) / 2.0).astype(np.float32)
    dz0, gp = orc.solve_adjoint_times(ctl, field, tout, z, gout, method, step)
    bdz0, bgp = orc.solve_discrete_backward_times(ctl, field, z0, tout, gout, method, step)
    g = {"coeffs": coeffs, "z0": z0, "t_out": tout, "grad_out": gout}
    if interp == "linear":
        g["knots"] = kn
    names = ["W0", "b0", "W1", "b1", "Wo", "bo"]
    # the planned problem dispatches to the cooperative kernels in every pass
    cd = torch.from_numpy(coeffs).cuda()
    X = (ncde_amd.LinearInterpolation if interp == "linear" else ncde_amd.NaturalCubicSpline)(cd, t=torch.from_numpy(kn).cuda() if interp == "linear" else None)
    func = gpu_util.CaseField(p, [("W0", "b0")] + [("W1", "b1")] * (nl - 1), "cuda")
    plan = solver._time_plan(X, torch.from_numpy(tout).cuda(), method, step, cd.device)
    prob = solver.build_problem(X.fused_coeffs, interp, torch.from_numpy(z0).cuda(), func.fused_spec(), method, _lib.OUT_TIMES, 0, plan)
    kn_ = [(_lib.lib().ncde_kernel_name(ctypes.byref(prob), k) or b"?").decode() for k in (0, 1, 2)]
    assert all("coop" in k for k in kn_), kn_
    for wflags in (0, _lib.FLAG_TILED_WINDOW_STEPS(3)):
        res = gpu_util.run_times_case(g, meta, adjoint=True, kind="original", mode="matmul", params=p, flags=wflags)
        resd = gpu_util.run_times_case(g, meta, adjoint=False, kind="original", mode="matmul", params=p, flags=wflags)
        assert gu.relerr(res["z_out"], z) <= TIGHT_Z, gu.relerr(res["z_out"], z)
        # dz0 per SAMPLE row: a pre-activation within rounding of zero may get the other ReLU mask than the oracle's and move that one
        # sample's row (the knife-edge below); every other row at the end-to-end bar, and the parameter gradients -- sums over all
        # samples -- at the bar as a whole
        bars_ = []
        for got_, ref_ in ((res["dz0"], dz0.numpy()), (resd["dz0"], bdz0.numpy())):
            rows_ = np.abs(got_ - ref_).max(axis=1) / np.abs(ref_).max()
            assert int((rows_ > E2E_G).sum()) <= 1 and float(rows_.max()) <= 50 * E2E_G, (int((rows_ > E2E_G).sum()), float(rows_.max()))
            bars_.append(E2E_G if float(rows_.max()) <= E2E_G else 2 * E2E_G)      # (that sample's share of a parameter gradient)
        for n_, g_ in zip(names, gp):
            assert gu.relerr(res["grads"][n_], g_) <= bars_[0], (n_, gu.relerr(res["grads"][n_], g_))
        for n_, g_ in zip(names, bgp):
            assert gu.relerr(resd["grads"][n_], g_) <= bars_[1], ("discrete", n_, gu.relerr(resd["grads"][n_], g_))
    # against the per-workgroup kernels: two fp32 implementations of the same sweep -- a pre-activation within rounding of zero may flip
    # a ReLU mask in one of them and move THAT sample's row (the knife-edge of DESIGN.md / HISTORY.md 5.5d): all but at most two rows
    old = gpu_util.run_times_case(g, meta, adjoint=True, kind="original", mode="matmul", params=p, flags=_lib.FLAG_NO_COOP)
    assert gu.relerr(old["z_out"], res["z_out"]) <= TIGHT_Z
    per = np.abs(old["dz0"] - res["dz0"]).max(axis=1) / np.abs(res["dz0"]).max()
    pero = np.abs(old["dz0"] - dz0.numpy()).max(axis=1) / np.abs(dz0.numpy()).max()
    assert int((per > E2E_G).sum()) <= 2, (int((per > E2E_G).sum()), float(per.max()), int((pero > E2E_G).sum()), float(pero.max()), np.argsort(per)[-3:].tolist())


@pytest.mark.parametrize("kind,interp,method,step", [("original", "linear", "rk4", 0.5), ("original", "cubic", "midpoint", 0.4),
                                                     ("minimal", "linear", "euler", 0.25), ("original", "cubic", "rk4", 0.75)])
def test_general_time_axis_on_the_batch_tiled_family_vs_oracle(kind, interp, method, step, gpu_lib):
    """Any output times / step size / user knot grid on the batch-tiled family (every width a multiple of 16, C of 4: the
    plan-driven default for such shapes) against the oracle's general-time functions (pinned to the reference on g11):
    forward, continuous adjoint (one reverse solve per output interval) and exact discrete backward, with more than one time
    window, for the original and the minimal-gated field; and the same problem on the generic / variant kernels agrees."""
    import gpu_util
    import ncde_oracle as orc
    B, L, C, H, HH, nl = 37, 9, 8, 32, 32, 2
    rng = np.random.RandomState(5)
    x = (gu.data.normal(41, B * L * C, stream=3).reshape(B, L, C) * 0.5).astype(np.float32)
    if interp == "linear":      # user knot grid, spacing 0.6 .. 1.4
        kn = np.cumsum(np.concatenate([[0.0], 0.6 + 0.8 * rng.rand(L - 1)])).astype(np.float32)
        x[:, :, 0] = kn[None, :]
        coeffs = x
    else:                       # integer knots (the numpy spline builder's grid); the step size and the output times are general
        kn = np.arange(L, dtype=np.float32)
        x[:, :, 0] = kn[None, :]
        coeffs = gu.data.natural_cubic_coeffs(x)
    p = gu.data.make_field_weights(H, HH, C, seed=19) if kind == "original" else gu.data.make_variant_weights(H, HH, C, seed=19, kind=kind, mode="matmul")
    z0 = (gu.data.normal(43, B * H, stream=2).reshape(B, H) * 0.5).astype(np.float32)
    tout = np.array([kn[0], 0.5 * (kn[1] + kn[2]), kn[4], kn[6] + 0.05, kn[-1] - 0.125], np.float32)
    meta = {"kind": interp, "method": method, "step_size": step, "dims": {"nl": nl}}
    field = orc.Field.variant(p, H, C, nl, kind, "matmul")
    ctl = orc.Control(coeffs, interp, t=kn if interp == "linear" else None)
    z = orc.solve_forward_times(ctl, field, z0, tout, method, step)
    gout = (gu.data.normal(23, z.numel(), stream=1).reshape(z.shape) / 2.0).astype(np.float32)
    dz0, gp = orc.solve_adjoint_times(ctl, field, tout, z, gout, method, step)
    bdz0, bgp = orc.solve_discrete_backward_times(ctl, field, z0, tout, gout, method, step)
    g = {"coeffs": coeffs, "z0": z0, "t_out": tout, "grad_out": gout}
    if interp == "linear":
        g["knots"] = kn
    names = [n for n in ("W0", "b0", "W1", "b1", "Wg", "bg", "Wo", "bo") if n in p]
    for wflags in (0x8000, 0x8000 | (3 << 16)):            # NCDE_FLAG_FORCE_TILED (round 4: a shape this small would otherwise be zero-padded
        # onto the plan-capable specialised kernels); default record budget, then NCDE_FLAG_TILED_WINDOW_STEPS(3)
        res = gpu_util.run_times_case(g, meta, adjoint=True, kind=kind, mode="matmul", params=p, flags=wflags)
        resd = gpu_util.run_times_case(g, meta, adjoint=False, kind=kind, mode="matmul", params=p, flags=wflags)
        assert gu.relerr(res["z_out"], z) <= TIGHT_Z, gu.relerr(res["z_out"], z)
        assert gu.relerr(res["dz0"], dz0) <= E2E_G, gu.relerr(res["dz0"], dz0)
        for n_, g_ in zip(names, gp):
            assert gu.relerr(res["grads"][n_], g_) <= E2E_G, (n_, gu.relerr(res["grads"][n_], g_))
        assert gu.relerr(resd["dz0"], bdz0) <= E2E_G
        for n_, g_ in zip(names, bgp):
            assert gu.relerr(resd["grads"][n_], g_) <= E2E_G, ("discrete", n_, gu.relerr(resd["grads"][n_], g_))
    ref = gpu_util.run_times_case(g, meta, adjoint=True, flags=1, kind=kind, mode="matmul", params=p)      # generic / variant kernels
    assert gu.relerr(ref["z_out"], z) <= TIGHT_Z
    # per sample: a pre-activation within rounding of zero flips a ReLU mask in ONE of two fp32 implementations and moves that
    # sample's dL/dz0 by percents (seen here: one sample of 37 at 2e-2, the rest at 8e-8) -- so all but a few samples must agree
    per = np.abs(ref["dz0"] - dz0.numpy()).max(axis=1) / np.abs(dz0.numpy()).max()
    assert np.sum(per <= E2E_G) >= B - 2, np.sort(per)[-4:]


@pytest.mark.parametrize("shape", [(20, 32, 32, 3), (4, 64, 64, 3), (5, 16, 15, 3), (3, 47, 32, 2), (20, 32, 32, 2)])
@pytest.mark.parametrize("interp,method,step", [("linear", "rk4", 0.5), ("cubic", "midpoint", 0.4), ("cubic", "rk4", 0.75), ("linear", "euler", 1.3)])
def test_general_time_axis_on_the_specialised_kernels_vs_oracle(shape, interp, method, step, gpu_lib):
    """Round 4: the register-resident kernel sets walk the time plan -- any output times / step size / user knot grid -- instead of
    leaving the general time axis to the batch-tiled family: `ncde_fwd_fast_bf3<..., time plan>` for both shapes (and everything
    zero-padded onto them), `ncde_adj_fast3<..., time plan>` for (32, 32, 20) with nl = 3, `ncde_adj_h64` for H = 64.  Against the
    oracle's general-time functions (pinned to the reference on g11): forward, continuous adjoint (one reverse solve per output
    interval), and -- on whatever family takes it -- the exact discrete backward; ragged batch, split-bf16 variant as well."""
    import ctypes
    import gpu_util
    import ncde_amd
    import ncde_oracle as orc
    from ncde_amd import _lib, solver
    C, H, HH, nl = shape
    B, L = 37, 9
    rng = np.random.RandomState(7)
    x = (gu.data.normal(61, B * L * C, stream=3).reshape(B, L, C) * 0.5).astype(np.float32)
    if interp == "linear":      # user knot grid, spacing 0.6 .. 1.4
        kn = np.cumsum(np.concatenate([[0.0], 0.6 + 0.8 * rng.rand(L - 1)])).astype(np.float32)
        x[:, :, 0] = kn[None, :]
        coeffs = x
    else:
        kn = np.arange(L, dtype=np.float32)
        x[:, :, 0] = kn[None, :]
        coeffs = gu.data.natural_cubic_coeffs(x)
    p = gu.data.make_field_weights(H, HH, C, seed=29)
    if nl == 1:
        p = {k: v for k, v in p.items() if k not in ("W1", "b1")}
    z0 = (gu.data.normal(63, B * H, stream=2).reshape(B, H) * 0.5).astype(np.float32)
    tout = np.array([kn[0], 0.5 * (kn[1] + kn[2]), kn[4], kn[6] + 0.05, kn[-1] - 0.125], np.float32)
    meta = {"kind": interp, "method": method, "step_size": step, "dims": {"nl": nl}}
    field = orc.Field.variant(p, H, C, nl, "original", "matmul")
    ctl = orc.Control(coeffs, interp, t=kn if interp == "linear" else None)
    z = orc.solve_forward_times(ctl, field, z0, tout, method, step)
    gout = (gu.data.normal(25, z.numel(), stream=1).reshape(z.shape) / 2.0).astype(np.float32)
    dz0, gp = orc.solve_adjoint_times(ctl, field, tout, z, gout, method, step)
    bdz0, bgp = orc.solve_discrete_backward_times(ctl, field, z0, tout, gout, method, step)
    g = {"coeffs": coeffs, "z0": z0, "t_out": tout, "grad_out": gout}
    if interp == "linear":
        g["knots"] = kn
    names = [n for n in ("W0", "b0", "W1", "b1", "Wo", "bo") if n in p]
    # which kernels the planned problem dispatches to
    cc = torch.from_numpy(coeffs).cuda()
    X = (ncde_amd.LinearInterpolation if interp == "linear" else ncde_amd.NaturalCubicSpline)(cc, t=torch.from_numpy(kn).cuda() if interp == "linear" else None)
    plan = solver._time_plan(X, torch.from_numpy(tout), method, step, cc.device)
    func = gpu_util.CaseField(p, [("W0", "b0")] + [("W1", "b1")] * (nl - 1), "cuda")
    prob = solver.build_problem(cc, interp, torch.from_numpy(z0).cuda(), func.fused_spec(), method, _lib.OUT_TIMES, 0, plan)
    kn_ = [(_lib.lib().ncde_kernel_name(ctypes.byref(prob), k) or b"?").decode() for k in (0, 1, 2)]
    big = H > 32 or HH > 32
    assert kn_[0].startswith("ncde_fwd_fast_bf3<H64" if big else "ncde_fwd_fast_bf3<H32") and "time plan" in kn_[0], kn_
    if big:
        assert kn_[1].startswith("ncde_adj_h64"), kn_
    elif nl == 3:
        assert kn_[1].startswith("ncde_adj_fast3") and "time plan" in kn_[1], kn_
    else:
        assert kn_[1].startswith("ncde_adj_tiled"), kn_      # other layer counts of the planned adjoint: batch-tiled family
    assert kn_[2].startswith("ncde_adj_tiled"), kn_
    def check(res, want_dz0, want_gp, rows_off, tag):
        # A pre-activation within rounding of zero flips a ReLU mask in one implementation and not the other, and moves ONE sample's
        # row (seen: sample 22 of shape2-linear-rk4-0.5 at 1e-3 under the split-fp16 forward-side recompute, 2^-22 products, where
        # the split-bf16 run of the SAME planned kernel agrees to 1e-7; and one row of shape1-cubic-rk4-0.75's discrete backward at
        # 3e-5 on which all four GPU families agree with each other).  So: every row but `rows_off` within E2E_G, and the batch-summed
        # parameter gradients within E2E_G whenever no row shows a flip (all rows at fp32 rounding).
        per = np.abs(res["dz0"] - want_dz0.numpy()).max(axis=1) / np.abs(want_dz0.numpy()).max()
        assert int((per > E2E_G).sum()) <= rows_off, (tag, np.sort(per)[-3:])
        flipped = int((per > 1e-5).sum())
        assert flipped <= 1, (tag, np.sort(per)[-3:])
        for n_, g_ in zip(names, want_gp):
            assert gu.relerr(res["grads"][n_], g_) <= (2e-2 if flipped else E2E_G), (tag, n_, gu.relerr(res["grads"][n_], g_))

    for flags in (_lib.FLAG_SPLIT_BF16, 0):
        res = gpu_util.run_times_case(g, meta, adjoint=True, params=p, flags=flags)
        assert gu.relerr(res["z_out"], z) <= TIGHT_Z, (flags, gu.relerr(res["z_out"], z))
        check(res, dz0, gp, 0 if flags else 1, ("adjoint", flags))
    resd = gpu_util.run_times_case(g, meta, adjoint=False, params=p)      # recording forward (planned specialised kernel) + discrete backward
    assert gu.relerr(resd["z_out"], z) <= TIGHT_Z
    check(resd, bdz0, bgp, 0, "discrete")


@pytest.mark.parametrize("kind,nl", [("original", 3), ("minimal", 3), ("original", 1)])
@pytest.mark.parametrize("mode,interp,method", [("evaluate", "linear", "rk4"), ("derivative", "cubic", "midpoint"), ("evaluate", "cubic", "euler")])
def test_evaluate_derivative_inputs_on_the_batch_tiled_family(kind, nl, mode, interp, method, gpu_lib):
    """The evaluate / derivative input modes (field input [z, X(t)] / [z, dX/dt]; H-row heads) where every width is a multiple
    of 16 and C of 4: forward and backward run on the batch-tiled family (layer 0 re-laid out with its H + C columns zero-padded;
    the heads' VJP and parameter gradients inside the sweep, no gradient pass).  Against the oracle (variant fields pinned to the
    reference on g9): forward, continuous adjoint and exact discrete backward end to end, default axis and a general one; and
    the variant kernels agree on the same problem."""
    import gpu_util
    import ncde_oracle as orc
    from ncde_amd import _lib
    B, L, C, H, HH = 37, 7, 8, 32, 48
    coeffs = gu.data.make_cubic_coeffs(B, L, C - 1, seed=61) if interp == "cubic" else gu.data.make_rectilinear_coeffs(B, L, C - 1, missing=0.3, seed=61)
    pfull = gu.data.make_variant_weights(H, HH, C, seed=29, kind=kind, mode=mode)
    p = {k: v for k, v in pfull.items() if nl > 1 or k not in ("W1", "b1")}      # a single inner layer: no shared second matrix
    z0 = (gu.data.normal(47, B * H, stream=2).reshape(B, H) * 0.5).astype(np.float32)
    field = orc.Field.variant(pfull, H, C, nl, kind, mode)
    ctl = orc.Control(coeffs, interp)
    names = [n for n in ("W0", "b0", "W1", "b1", "Wg", "bg", "Wo", "bo") if n in p]
    layers = [("W0", "b0")] + [("W1", "b1")] * (nl - 1)
    meta = {"kind": interp, "method": method, "sequence": True, "param_names": names, "field_kind": kind, "field_mode": mode,
            "dims": {"C": C, "H": H, "HH": HH, "nl": nl}, "field": "original"}
    z = orc.solve_forward(ctl, field, z0, method, True)
    gout = (gu.data.normal(31, z.numel(), stream=1).reshape(z.shape) / np.sqrt(z.shape[1])).astype(np.float32)
    dz0, gp = orc.solve_adjoint(ctl, field, z, gout, method, True)
    bdz0, bgp = orc.solve_discrete_backward(ctl, field, z0, gout, method, True)
    case = {"meta": meta, "coeffs": coeffs, "z0": z0, "params": p, "layers": layers, "H": H, "C": C, "expect": {"grad_out": gout}}
    res = gpu_util.run_case(case)
    assert res["kernels"][0] == ("ncde_fwd_tiled<NS1,gated,direct>" if kind == "minimal" else "ncde_fwd_tiled<NS1,direct>"), res["kernels"]
    assert res["kernels"][1].startswith("ncde_adj_tiled<") and "direct" in res["kernels"][1] and "discrete" in res["kernels"][2], res["kernels"]
    assert gu.relerr(res["z_out"], z) <= TIGHT_Z, gu.relerr(res["z_out"], z)
    assert gu.relerr(res["dz0"], dz0) <= E2E_G
    for n_, g_ in zip(names, gp):
        assert gu.relerr(res["grads"][n_], g_) <= E2E_G, n_
    resd = gpu_util.run_case(case, adjoint=False)                 # the tiled forward writes the stage record the variant backward reads
    assert np.array_equal(resd["z_out"], res["z_out"])
    assert gu.relerr(resd["dz0"], bdz0) <= E2E_G
    for n_, g_ in zip(names, bgp):
        assert gu.relerr(resd["grads"][n_], g_) <= E2E_G, ("discrete", n_)
    ref = gpu_util.run_case(case, flags=_lib.FLAG_FORCE_GENERIC)      # the variant kernels on the same problem
    assert ref["kernels"][0] == "ncde_fwd_variant" and ref["kernels"][1] == "ncde_adj_variant" and gu.relerr(ref["z_out"], z) <= TIGHT_Z
    for n_, g_ in zip(names, gp):
        assert gu.relerr(ref["grads"][n_], g_) <= E2E_G, ("variant kernels", n_)
    # general time axis (step 0.5, off-grid outputs) through the same tiled forward
    tout = np.array([0.0, 1.25, 3.0, float(ctl.n_knots - 1) - 0.5], np.float32)
    zt = orc.solve_forward_times(ctl, field, z0, tout, method, 0.5)
    gt = (gu.data.normal(33, zt.numel(), stream=1).reshape(zt.shape) / 2.0).astype(np.float32)
    dzt, gpt = orc.solve_adjoint_times(ctl, field, tout, zt, gt, method, 0.5)
    rt = gpu_util.run_times_case({"coeffs": coeffs, "z0": z0, "t_out": tout, "grad_out": gt}, {"kind": interp, "method": method, "step_size": 0.5, "dims": {"nl": nl}},
                                 adjoint=True, kind=kind, mode=mode, params=p)
    assert gu.relerr(rt["z_out"], zt) <= TIGHT_Z and gu.relerr(rt["dz0"], dzt) <= E2E_G
    for n_, g_ in zip(names, gpt):
        assert gu.relerr(rt["grads"][n_], g_) <= E2E_G, ("times", n_)
    bdzt, bgpt = orc.solve_discrete_backward_times(ctl, field, z0, tout, gt, method, 0.5)
    rtd = gpu_util.run_times_case({"coeffs": coeffs, "z0": z0, "t_out": tout, "grad_out": gt}, {"kind": interp, "method": method, "step_size": 0.5, "dims": {"nl": nl}},
                                  adjoint=False, kind=kind, mode=mode, params=p)
    assert gu.relerr(rtd["dz0"], bdzt) <= E2E_G
    for n_, g_ in zip(names, bgpt):
        assert gu.relerr(rtd["grads"][n_], g_) <= E2E_G, ("times, discrete", n_)


def test_variant_gradient_partial_in_global_memory_is_reproducible(gpu_lib):
    """GRU-gated field with the matmul input at cfg2 widths: the two 640 x 32 heads do not fit LDS, so the variant adjoint keeps
    its per-workgroup gradient partial in global memory and accumulates into it with no-return float atomics (every address
    is touched by one wave, in program order).  Against the oracle, and twice: bit-identical."""
    import gpu_util
    case = _seeded_case("linear", "rk4", False, B=40, L=9, C=20, H=32, HH=32, nl=3, seed=910, kind="gru")
    ex = case["expect"]
    names = gpu_util.kernel_names(case)
    assert names[1].startswith("ncde_adj_variant") and names[2].startswith("ncde_adj_variant"), names      # the family under test is the one dispatched
    a = gpu_util.run_adjoint_direct(case, ex["z_out"])
    for k, e in _grad_errors(case, a).items():
        assert e <= TIGHT_G, (k, e)
    b = gpu_util.run_adjoint_direct(case, ex["z_out"])
    assert np.array_equal(a["dz0"], b["dz0"]) and all(np.array_equal(a["grads"][k], b["grads"][k]) for k in a["grads"])
    d1 = gpu_util.run_adjoint_direct(case, ex["z_out"], stages=case["stage_record"])
    d2 = gpu_util.run_adjoint_direct(case, ex["z_out"], stages=case["stage_record"])
    for k, e in _grad_errors(case, d1, "bp_").items():
        assert e <= TIGHT_G, ("discrete", k, e)
    assert np.array_equal(d1["dz0"], d2["dz0"]) and all(np.array_equal(d1["grads"][k], d2["grads"][k]) for k in d1["grads"])


@pytest.mark.parametrize("kind,mode,interp,method,step", [("gru", "evaluate", "cubic", "rk4", 0.5), ("minimal", "matmul", "linear", "midpoint", 0.4),
                                                         ("original", "derivative", "linear", "euler", 0.25)])
def test_general_time_axis_field_variants_vs_oracle(kind, mode, interp, method, step, gpu_lib):
    """Gated fields / evaluate / derivative inputs on a general time axis (user knots, off-grid outputs) against the oracle
    (whose general-time functions are pinned to the reference on g11 and whose variant fields are pinned on g9)."""
    import os
    import gpu_util
    import ncde_oracle as orc
    f = dict(np.load(os.path.join(gu.GOLD, "g11_knots_rk4.npz" if interp == "linear" else "g11_knots_cubic_euler.npz")))
    C, H, HH, nl = 5, 16, 24, 3
    p = gu.data.make_variant_weights(H, HH, C, seed=17, kind=kind, mode=mode)
    kn = f["knots"]
    tout = np.array([kn[0], 0.5 * (kn[1] + kn[2]), kn[4], kn[-1] - 0.125], np.float32)
    meta = {"kind": interp, "method": method, "step_size": step, "dims": {"nl": nl}}
    field = orc.Field.variant(p, H, C, nl, kind, mode)
    ctl = orc.Control(f["coeffs"], interp, t=kn)
    z = orc.solve_forward_times(ctl, field, f["z0"], tout, method, step)
    gout = (gu.data.normal(23, z.numel(), stream=1).reshape(z.shape) / 2.0).astype(np.float32)
    dz0, gp = orc.solve_adjoint_times(ctl, field, tout, z, gout, method, step)
    bdz0, bgp = orc.solve_discrete_backward_times(ctl, field, f["z0"], tout, gout, method, step)
    g = dict(f, t_out=tout, grad_out=gout)
    names = [n for n in ("W0", "b0", "W1", "b1", "Wr", "br", "Wg", "bg", "Wo", "bo") if n in p]
    res = gpu_util.run_times_case(g, meta, adjoint=True, kind=kind, mode=mode, params=p)
    assert gu.relerr(res["z_out"], z) <= TIGHT_Z
    assert gu.relerr(res["dz0"], dz0) <= E2E_G
    for n_, g_ in zip(names, gp):
        assert gu.relerr(res["grads"][n_], g_) <= E2E_G, n_
    resd = gpu_util.run_times_case(g, meta, adjoint=False, kind=kind, mode=mode, params=p)
    assert gu.relerr(resd["dz0"], bdz0) <= E2E_G
    for n_, g_ in zip(names, bgp):
        assert gu.relerr(resd["grads"][n_], g_) <= E2E_G, n_


def test_default_axis_through_the_time_plan_equals_the_default_kernels(gpu_lib):
    """The same solve requested as explicit times (t = arange(T) as a plain tensor with step 1 is recognised as the default
    axis; step 0.5 is not) -- a plan whose steps are the integer grid must reproduce the default-path generic kernels bit
    for bit, forward and both backward modes (the plan only replaces WHERE the times come from)."""
    import gpu_util
    from ncde_amd import solver
    case = gu.load_case("g2_rect_rk4_seq")
    coeffs = torch.from_numpy(case["coeffs"]).cuda()
    X = ncde_amd_mod().LinearInterpolation(coeffs)
    T = coeffs.shape[1]
    plan = solver._time_plan(X, torch.arange(T, dtype=torch.float32), "rk4", 1.0, coeffs.device)
    assert plan[1] == (T, T - 1, T - 1)
    # family by family: generic (flag 1) and batch-tiled (flag 0x8000; the plan-driven default for this aligned shape)
    # ... and (round 4) the specialised register-resident kernels (flag 0: this shape zero-pads onto the (32, 32, 20) set; dt = 1
    # multiplies exactly and the stage combinations are the same expressions)
    for fam, adjoint in ((1, True), (1, False), (0x8000, True), (0x8000, False), (0, True), (0, False)):
        want = gpu_util.run_case(case, flags=fam, adjoint=adjoint)
        func = gpu_util.case_field(case, "cuda")
        z0 = torch.from_numpy(case["z0"]).cuda().requires_grad_(True)
        cfg = {"spec": func.fused_spec(), "interp": "linear", "method": "rk4", "output": 2, "flags": fam, "plan": plan, "adjoint": adjoint,
               "func": None, "nfe_per_solve": 0, "nfe_adjoint": 0, "adjoint_param_ids": None}
        out = solver._FusedCdeint.apply(z0, coeffs, cfg, *func.fused_spec().unique_params())
        (out * torch.from_numpy(case["expect"]["grad_out"]).cuda()).sum().backward()
        assert np.array_equal(out.detach().cpu().numpy(), want["z_out"]), (fam, adjoint, gu.relerr(out.detach().cpu().numpy(), want["z_out"]))
        # the planned exact discrete backward of the specialised family is the batch-tiled sweep, the default-axis one is
        # ncde_adj_fast3<discrete>: two kernels, so equal to fp32 rounding there and bit for bit everywhere else
        same = np.array_equal if (fam, adjoint) != (0, False) else (lambda a, b: gu.relerr(a, b) <= 2e-6)
        assert same(z0.grad.cpu().numpy(), want["dz0"]), (fam, adjoint, gu.relerr(z0.grad.cpu().numpy(), want["dz0"]))
        for k, v in func.p.items():
            assert same(v.grad.cpu().numpy(), want["grads"][k]), (fam, adjoint, k, gu.relerr(v.grad.cpu().numpy(), want["grads"][k]))


def ncde_amd_mod():
    import ncde_amd
    return ncde_amd


def test_reference_cdeint_shape_test_ported(gpu_lib):
    """Port of the reference's own cdeint test (/root/reference/modules/torchcde/test/test_cdeint.py:5-41, rk4 branch):
    random batch dimensions (0..2), natural cubic spline of random values, random fp64 output times inside the interval,
    step_size = 1/num_points; the vector field is an OriginalVectorField (arbitrary Python fields are outside the fused
    path).  Checks the output shape as the reference does, plus finiteness and z(t[0]) = z0."""
    import ncde_amd
    gen = torch.Generator().manual_seed(0)
    ri = lambda lo, hi: int(torch.randint(low=lo, high=hi, size=(1,), generator=gen).item())     # noqa: E731
    for _ in range(10):
        num_points, num_channels, num_hidden = ri(5, 100), ri(1, 3), ri(1, 5)
        batch_dims = [ri(1, 3) for _ in range(ri(0, 3))]
        values = torch.rand(*batch_dims, num_points, num_channels, generator=gen).cuda()
        coeffs = ncde_amd.natural_cubic_coeffs(values)
        spline = ncde_amd.NaturalCubicSpline(coeffs)
        torch.manual_seed(1)
        f = ncde_amd.OriginalVectorField(num_channels, num_hidden, 7, 2).cuda()
        z0 = torch.rand(*batch_dims, num_hidden, generator=gen).cuda()
        num_out_times = ri(2, 10)
        start, end = spline.interval
        out_times = torch.rand(num_out_times, dtype=torch.float64, generator=gen).sort().values.cuda() * (end - start) + start
        out = ncde_amd.cdeint(spline, f, z0, out_times, method="rk4", options={"step_size": 1.0 / num_points}, rtol=1e-4, atol=1e-6)
        assert out.shape == (*batch_dims, num_out_times, num_hidden)
        assert torch.isfinite(out).all()
        assert torch.equal(out[..., 0, :], z0)


DOPRI5_CASES = ["g10_toy_dopri5_seq", "g10_ncde_dopri5_rect_final", "g10_ncde_dopri5_rect_seq", "g10_ncde_dopri5_cubic_final",
                "g10_ncde_dopri5_cubic_seq", "g10_adaptive_cubic_final"]


@pytest.mark.parametrize("name", DOPRI5_CASES)
def test_dopri5_matches_reference_golden(name, gpu_lib):
    """Adaptive dopri5 (goldens g10 = the imported reference): the toy's default call and the
    NeuralCDE(solver='dopri5') setting (min_step 0.5, rtol 1e-3, atol 1e-5), forward + adaptive continuous adjoint.
    An adaptive solve turns last-bit differences into different step sequences: the embedded error estimate is a small
    difference of large terms (fp32 noise of 1e-7 in the stages is 1e-3 .. 1e-2 of a small error ratio, hence 1e-4 of the
    next dt), and steps cluster at the kinks of a piecewise-linear control (oracle/gen_golden.py documents the same effect
    between two CPU runs of the reference's own algorithm).  Both runs are solves of the same ODE at tolerance rtol:
    z within 20 rtol, gradients within 5e-2.  The arithmetic itself (tableau, stage times, dense output, adaptive adjoint
    with its mixed norm and time-gradient component) is pinned tightly by the forced-step-sequence test below."""
    import json
    import os
    import gpu_util
    import ncde_amd
    f = dict(np.load(os.path.join(gu.GOLD, name + ".npz")))
    m = json.loads(str(f["meta"]))
    coeffs = torch.from_numpy(f["coeffs"]).cuda()
    X = (ncde_amd.LinearInterpolation if m["kind"] == "linear" else ncde_amd.NaturalCubicSpline)(coeffs)
    params = {k[2:]: f[k] for k in f if k.startswith("p_")}
    layers = [("W0", "b0"), ("W1", "b1")] if m["field"] == "toy" else [("W0", "b0")] + [("W1", "b1")] * (m["dims"]["nl"] - 1)
    func = gpu_util.CaseField(params, layers, "cuda")
    z0 = torch.from_numpy(f["z0"]).cuda().requires_grad_(True)
    t = X.grid_points if m["sequence"] else X.interval
    kw = {"options": {"_trace": 4096}} if m["field"] == "toy" else {"method": "dopri5", "rtol": m["rtol"], "atol": m["atol"],
                                                                        "options": dict(m["options"], _trace=4096)}
    out = ncde_amd.cdeint(X, func, z0, t, adjoint=True, **kw)        # the toy omits the method: dopri5 is cdeint's default
    nfe_fwd = func.nfe
    # The CONTROLLER, free-running, against the reference's own step sequence (golden meta `trace_fwd`: t0, dt, accepted): initial
    # step, accept / reject with the min_step override, next dt.  Same decisions and dt within 1 % over the first 20 attempts (all of
    # them in practice: the sequences drift apart at the 1e-4 .. 1e-3 level per attempt -- dt_next = 0.9 dt / ratio^0.2 of error
    # ratios that are cancellation noise when small -- long before a decision flips).
    tr, ref_tr = func.dopri5_trace, np.asarray(m["trace_fwd"], dtype=np.float64)
    n_same = 0
    while n_same < min(len(tr), len(ref_tr)) and tr[n_same, 2] == ref_tr[n_same, 2] and abs(tr[n_same, 1] - ref_tr[n_same, 1]) <= 1e-2 * ref_tr[n_same, 1]:
        n_same += 1
    # (at rtol <= 1e-4 the first error ratios are ~1e-5, i.e. pure fp32 noise, and so is the second dt: there only the initial step
    # of _select_initial_step is comparable)
    assert n_same >= (min(20, len(ref_tr)) if m["rtol"] >= 1e-3 else 1), (n_same, tr[:n_same + 1, :3], ref_tr[:n_same + 1])
    assert out.shape == f["z_out"].shape
    ez = gu.relerr(out.detach().cpu().numpy(), f["z_out"])
    same_f = nfe_fwd == m["nfe_fwd"]
    (out * torch.from_numpy(f["grad_out"]).cuda()).sum().backward()
    nfe_bwd = func.nfe - nfe_fwd
    same_b = same_f and nfe_bwd == m["nfe_bwd"]
    eg = {"dz0": gu.relerr(z0.grad.cpu().numpy(), f["dz0"])}
    for pname in m["param_names"]:
        eg[pname] = gu.relerr(func.p[pname].grad.cpu().numpy(), f["d" + pname])
    print("%s: nfe fwd %d (ref %d) bwd %d (ref %d); z %.2e; grads %s" % (name, nfe_fwd, m["nfe_fwd"], nfe_bwd, m["nfe_bwd"], ez,
                                                                        {k: "%.1e" % v for k, v in eg.items()}))
    # below rtol ~ 1e-4 the embedded error estimate sits under fp32 resolution (the reference's own runs scatter the same way,
    # MANIFEST_dopri5.json): floor of 2e-3
    assert ez <= _free_running_z_bar(m), (ez, same_f)
    for k, e in eg.items():
        assert e <= _free_running_grad_bar(m), (k, e, same_b)
    assert (nfe_fwd - 2) % 6 == 0 and nfe_bwd % 2 == 0
    assert abs(nfe_fwd - m["nfe_fwd"]) <= 0.25 * m["nfe_fwd"] and abs(nfe_bwd - m["nfe_bwd"]) <= 0.25 * m["nfe_bwd"]      # same amount of work


def _free_running_z_bar(m):
    """How far a FREE-RUNNING dopri5 solve may sit from the reference's: 20 rtol (floor 2e-3) for smooth solves; 50 rtol where
    `min_step` forces acceptance on a piecewise-linear control -- there the accepted steps at the kinks carry error ratios of 5 - 10 (the
    reference's own trace), the next dt is 0.9 dt / ratio^0.2 of error ratios that are cancellation noise when small (7.6e-4 +- 5e-4
    relative between two fp32 implementations -> 1e-4 in dt), and two fp32 implementations part ways within ~15 attempts: the fused
    attempt kernels (split-bf16 GEMMs, exp2-based tanh) and the per-launch kernels (fp32 MFMA) agree to 1e-6 on the first output rows
    of g10_ncde_dopri5_rect_seq and differ by 2e-2 on the last, with the same number of attempts (tools/dbg_dp5.py).  The arithmetic is
    pinned by the replay / forced-sequence tests at 2e-5."""
    forced = m["kind"] == "linear" and m["options"].get("min_step", 0) > 0
    return max((50 if forced else 20) * m["rtol"], 2e-3)


def _free_running_grad_bar(m):
    """Gradients of a free-running solve: 5e-2; 0.3 in the forced-acceptance regime above, where the two implementations' forward
    solutions already differ by 2e-2 at the last rows and each backward solve then takes its own step sequence (measured: parameter
    gradients 4e-2 .. 2.5e-1 apart between the fused and the per-launch kernels, both within 2 % of the reference's number of
    attempts; the same kernels agree with the oracle at 2e-4 when they replay one sequence)."""
    forced = m["kind"] == "linear" and m["options"].get("min_step", 0) > 0
    return 0.3 if forced else 5e-2


def _dopri5_golden_setup(name):
    import json
    import os
    import gpu_util
    import ncde_amd
    f = dict(np.load(os.path.join(gu.GOLD, name + ".npz")))
    m = json.loads(str(f["meta"]))
    coeffs = torch.from_numpy(f["coeffs"]).cuda()
    X = (ncde_amd.LinearInterpolation if m["kind"] == "linear" else ncde_amd.NaturalCubicSpline)(coeffs)
    params = {k[2:]: f[k] for k in f if k.startswith("p_")}
    layers = [("W0", "b0"), ("W1", "b1")] if m["field"] == "toy" else [("W0", "b0")] + [("W1", "b1")] * (m["dims"]["nl"] - 1)
    func = gpu_util.CaseField(params, layers, "cuda")
    z0 = torch.from_numpy(f["z0"]).cuda().requires_grad_(True)
    t = X.grid_points if m["sequence"] else X.interval
    return f, m, X, func, z0, t


@pytest.mark.parametrize("name", DOPRI5_CASES)
def test_dopri5_replay_of_the_reference_step_sequence(name, gpu_lib):
    """The adaptive kernels made to take the REFERENCE's own step sequence (goldens g10: `trace_fwd` / `trace_bwd` = dt and accept /
    reject of every attempt of the imported reference, forward and all reverse solves): the solve then does the reference's
    arithmetic step for step, and z, dL/dz0, dL/dtheta and nfe are held to the tight tolerances of the fixed-step path instead of
    solver-tolerance level.  Separates "rounding flipped an accept" (the free-running test above) from "the controller's or the
    stages' arithmetic differs" (this test)."""
    import ncde_amd
    f, m, X, func, z0, t = _dopri5_golden_setup(name)
    opts = dict(m["options"])
    fwd = dict(opts, _replay=[[r[1], r[2]] for r in m["trace_fwd"]])
    bwd = dict(opts, _replay=[[r[1], r[2]] for r in m["trace_bwd"]])
    out = ncde_amd.cdeint(X, func, z0, t, adjoint=True, method="dopri5", rtol=m["rtol"], atol=m["atol"], options=fwd, adjoint_options=bwd)
    assert func.nfe == m["nfe_fwd"]
    assert gu.relerr(out.detach().cpu().numpy(), f["z_out"]) <= TIGHT_Z
    (out * torch.from_numpy(f["grad_out"]).cuda()).sum().backward()
    assert func.nfe - m["nfe_fwd"] == m["nfe_bwd"]
    assert gu.relerr(z0.grad.cpu().numpy(), f["dz0"]) <= E2E_G
    for pname in m["param_names"]:
        assert gu.relerr(func.p[pname].grad.cpu().numpy(), f["d" + pname]) <= E2E_G, pname


DOPRI5_TAPED_CASES = ["g12_ncde_dopri5_rect_final", "g12_ncde_dopri5_rect_seq", "g12_ncde_dopri5_linear_final", "g12_ncde_dopri5_cubic_final",
                      "g12_ncde_dopri5_cubic_seq", "g12_adaptive_cubic_final", "g12_first_step_given_rect_seq"]


@pytest.mark.parametrize("name", DOPRI5_TAPED_CASES)
@pytest.mark.parametrize("replay", [True, False])
def test_dopri5_adjoint_false_matches_reference_golden(name, replay, gpu_lib):
    """dopri5 with adjoint=False -- the setting of the reference's shipped "interpolation" experiments
    (experiments/configurations/configurations.json5:187-191 -> NeuralCDE(solver='dopri5', adjoint=False), options {'min_step': 0.5}):
    goldens g12 = the imported reference's autograd through its taped adaptive solve, first-step-size gradient included (2-7 % of
    dL/dz0 on these cases).  replay = the reference's forward step sequence forced (tight: the reverse sweep over the recorded steps
    has no step control of its own); free-running = solver-tolerance level, as for the adjoint=True goldens."""
    import ncde_amd
    f, m, X, func, z0, t = _dopri5_golden_setup(name)
    opts = dict(m["options"])
    if replay:
        opts["_replay"] = [[r[1], r[2]] for r in m["trace_fwd"]]
    out = ncde_amd.cdeint(X, func, z0, t, adjoint=False, method="dopri5", rtol=m["rtol"], atol=m["atol"], options=opts)
    nfe = func.nfe
    (out * torch.from_numpy(f["grad_out"]).cuda()).sum().backward()
    assert func.nfe == nfe                          # autograd's backward evaluates no vector field the reference counts
    ez = gu.relerr(out.detach().cpu().numpy(), f["z_out"])
    eg = {"dz0": gu.relerr(z0.grad.cpu().numpy(), f["bp_dz0"])}
    for pname in m["param_names"]:
        eg[pname] = gu.relerr(func.p[pname].grad.cpu().numpy(), f["bp_d" + pname])
    print("%s replay=%s: nfe %d (ref %d); z %.2e; grads %s" % (name, replay, nfe, m["nfe_fwd"], ez, {k: "%.1e" % v for k, v in eg.items()}))
    if replay:
        assert nfe == m["nfe_fwd"]
        assert ez <= TIGHT_Z
        assert all(e <= E2E_G for e in eg.values()), eg
    elif nfe == m["nfe_fwd"]:                       # same step sequence, dt differing in the last bits: solver-tolerance level
        assert ez <= _free_running_z_bar(m)
        assert all(e <= _free_running_grad_bar(m) for e in eg.values()), eg
    else:
        # a different accept / reject somewhere (fp32 noise at an error ratio within 1 % of 1, MANIFEST_dopri5*.json): another valid
        # run of the same algorithm.  With min_step forcing acceptance the solution itself is not tolerance-controlled, and whether
        # attempt 1 is accepted switches the first-step gradient on or off -- only sanity bounds hold; the tight statement is replay.
        assert abs(nfe - m["nfe_fwd"]) <= 0.25 * m["nfe_fwd"]
        assert ez <= 0.2 and all(np.isfinite(e) and e <= 0.5 for e in eg.values()), (ez, eg)


@pytest.mark.parametrize("interp,seq", [("linear", False), ("linear", True), ("cubic", False), ("cubic", True)])
def test_dopri5_adjoint_false_forced_steps_vs_oracle(interp, seq, gpu_lib):
    """adjoint=False with min_step = max_step = 0.5 and NO first_step: dt_1 comes from the initial-step rule (differentiable, misc.py:33-74)
    and is accepted (dt <= min_step), every later dt is 0.5 -- GPU and oracle (pinned to the reference's autograd on g12) walk the same
    sequence, so the reverse sweep incl. the first-step gradient through the batch-wide norms is compared tightly, ragged batch."""
    import ncde_amd
    import gpu_util
    import ncde_oracle as orc
    B, L, C, H, HH, nl = 21, 7, 5, 16, 24, 3
    if interp == "linear":
        coeffs = gu.data.make_rectilinear_coeffs(B, L, C - 1, missing=0.3, seed=95)
        x0 = coeffs[:, 0]
    else:
        coeffs = gu.data.make_cubic_coeffs(B, 2 * L, C - 1, seed=96)
        x0 = coeffs[:, 0, :C]
    p = gu.data.make_field_weights(H, HH, C, seed=9)
    rw = gu.data.make_readin_weights(H, C, 1, seed=9)
    z0n = (x0 @ rw["Wi"].T + rw["bi"]).astype(np.float32)
    opts = {"min_step": 0.5, "max_step": 0.5}
    field = orc.Field.original(p, H, C, nl)
    ctl = orc.Control(coeffs, interp)
    T = ctl.n_knots
    tt = torch.arange(T, dtype=torch.float32) if seq else torch.tensor([0.0, T - 1.0])
    st = {}
    gshape = (B, T if seq else 2, H)
    gout = (gu.data.normal(33, int(np.prod(gshape)), stream=1).reshape(gshape) / np.sqrt(gshape[1])).astype(np.float32)
    z, dz0, gp = orc.dopri5_discrete_backward(ctl, field, z0n, tt, gout, 1e-3, 1e-5, opts, stats=st)
    assert st["delta_active"]
    X = (ncde_amd.LinearInterpolation if interp == "linear" else ncde_amd.NaturalCubicSpline)(torch.from_numpy(coeffs).cuda())
    func = gpu_util.CaseField(p, [("W0", "b0")] + [("W1", "b1")] * (nl - 1), "cuda")
    z0 = torch.from_numpy(z0n).cuda().requires_grad_(True)
    out = ncde_amd.cdeint(X, func, z0, X.grid_points if seq else X.interval, adjoint=False, method="dopri5", rtol=1e-3, atol=1e-5, options=dict(opts))
    assert func.nfe == st["nfe"]
    assert gu.relerr(out.detach().cpu().numpy(), z.numpy()) <= TIGHT_Z
    (out * torch.from_numpy(gout).cuda()).sum().backward()
    assert gu.relerr(z0.grad.cpu().numpy(), dz0.numpy()) <= E2E_G
    for n_, g_ in zip(["W0", "b0", "W1", "b1", "Wo", "bo"], gp):
        assert gu.relerr(func.p[n_].grad.cpu().numpy(), g_.numpy()) <= E2E_G, n_


def test_dopri5_max_num_steps_counts_per_output_and_frozen_parameters(gpu_lib):
    """Two behaviours of the reference the adaptive kernels follow:  (1) `max_num_steps` bounds the attempts of ONE
    _advance(next_t) call, i.e. per output time (rk_common.py:196-203), not of the whole solve;  (2) the adjoint's augmented state holds
    only the parameters that require a gradient / are listed in adjoint_params (adjoint.py:176-189): the others take no part in the
    mixed error norm and receive no gradient."""
    import ncde_amd
    f, m, X, func, z0, t = _dopri5_golden_setup("g10_ncde_dopri5_rect_seq")
    n_attempts = sum(m["steps_fwd"])
    kw = dict(method="dopri5", rtol=m["rtol"], atol=m["atol"])
    out = ncde_amd.cdeint(X, func, z0, t, adjoint=True, options=dict(m["options"], max_num_steps=n_attempts // 2), **kw)     # fewer than the
    assert gu.relerr(out.detach().cpu().numpy(), f["z_out"]) <= _free_running_z_bar(m)                                # whole solve takes
    with pytest.raises(AssertionError, match="max_num_steps"):
        ncde_amd.cdeint(X, func, z0, t, adjoint=True, options=dict(m["options"], max_num_steps=1), **kw)
    # frozen parameter: no gradient, and it is not a segment of the error norm (the solve still runs and the others agree with the
    # all-parameters run to solver tolerance)
    (out * torch.from_numpy(f["grad_out"]).cuda()).sum().backward()
    full = {k: v.grad.clone() for k, v in func.p.items()}
    for v in func.p.values():
        v.grad = None
    func.p["W1"].requires_grad_(False)
    z0b = z0.detach().clone().requires_grad_(True)
    out2 = ncde_amd.cdeint(X, func, z0b, t, adjoint=True, options=dict(m["options"]), **kw)
    (out2 * torch.from_numpy(f["grad_out"]).cuda()).sum().backward()
    assert func.p["W1"].grad is None
    for k, v in func.p.items():
        if k != "W1":
            assert v.grad is not None and gu.relerr(v.grad.cpu().numpy(), full[k].cpu().numpy()) <= 5e-2, k
    # adjoint_params subset
    func.p["W1"].requires_grad_(True)
    for v in func.p.values():
        v.grad = None
    z0c = z0.detach().clone().requires_grad_(True)
    out3 = ncde_amd.cdeint(X, func, z0c, t, adjoint=True, options=dict(m["options"]), adjoint_params=(func.p["Wo"], func.p["bo"]), **kw)
    (out3 * torch.from_numpy(f["grad_out"]).cuda()).sum().backward()
    assert func.p["W0"].grad is None and func.p["W1"].grad is None and func.p["Wo"].grad is not None


@pytest.mark.parametrize("interp,seq", [("linear", False), ("linear", True), ("cubic", False), ("cubic", True)])
def test_dopri5_forced_step_sequence_vs_oracle(interp, seq, gpu_lib):
    """dopri5 with first_step = min_step = max_step = 0.5: every attempt has dt = 0.5 and is accepted (rk_common.py:262-266), so
    the GPU and the oracle (pinned to the reference on g10) walk the SAME step sequence and the comparison is tight: forward
    (stage times incl. the one-ulp perturbation of the alpha = 1 stages, Butcher sums, 4th-order dense output at the knots),
    adaptive adjoint (per-interval restarts, theta part, vjp_t for the cubic spline, dense output at the interval ends)."""
    import ncde_amd
    import gpu_util
    import ncde_oracle as orc
    B, L, C, H, HH, nl = 21, 7, 5, 16, 24, 3
    if interp == "linear":
        coeffs = gu.data.make_rectilinear_coeffs(B, L, C - 1, missing=0.3, seed=95)
        x0 = coeffs[:, 0]
    else:
        coeffs = gu.data.make_cubic_coeffs(B, 2 * L, C - 1, seed=96)
        x0 = coeffs[:, 0, :C]
    p = gu.data.make_field_weights(H, HH, C, seed=9)
    rw = gu.data.make_readin_weights(H, C, 1, seed=9)
    z0n = (x0 @ rw["Wi"].T + rw["bi"]).astype(np.float32)
    opts = {"first_step": 0.5, "min_step": 0.5, "max_step": 0.5}
    field = orc.Field.original(p, H, C, nl)
    ctl = orc.Control(coeffs, interp)
    T = ctl.n_knots
    tt = torch.arange(T, dtype=torch.float32) if seq else torch.tensor([0.0, T - 1.0])
    so, sb = {}, {}
    z = orc.dopri5_forward(ctl, field, z0n, tt, 1e-3, 1e-5, opts, stats=so)
    gout = (gu.data.normal(31, z.numel(), stream=1).reshape(z.shape) / np.sqrt(z.shape[1])).astype(np.float32)
    dz0, gp = orc.dopri5_adjoint(ctl, field, tt, z, gout, 1e-3, 1e-5, opts, stats=sb)
    X = (ncde_amd.LinearInterpolation if interp == "linear" else ncde_amd.NaturalCubicSpline)(torch.from_numpy(coeffs).cuda())
    func = gpu_util.CaseField(p, [("W0", "b0")] + [("W1", "b1")] * (nl - 1), "cuda")
    z0 = torch.from_numpy(z0n).cuda().requires_grad_(True)
    out = ncde_amd.cdeint(X, func, z0, X.grid_points if seq else X.interval, adjoint=True, method="dopri5", rtol=1e-3, atol=1e-5, options=dict(opts))
    assert func.nfe == so["nfe"]                 # first_step given: no probe evaluation, 1 + 6 per attempt
    assert gu.relerr(out.detach().cpu().numpy(), z) <= TIGHT_Z
    (out * torch.from_numpy(gout).cuda()).sum().backward()
    assert func.nfe - so["nfe"] == sb["nfe"]
    assert gu.relerr(z0.grad.cpu().numpy(), dz0) <= E2E_G
    for n_, g_ in zip(["W0", "b0", "W1", "b1", "Wo", "bo"], gp):
        assert gu.relerr(func.p[n_].grad.cpu().numpy(), g_) <= E2E_G, n_


@pytest.mark.parametrize("C,H,HH,nl,interp", [(20, 32, 32, 3, "linear"), (3, 48, 64, 2, "cubic"), (5, 16, 24, 1, "cubic"), (7, 40, 40, 2, "linear"),
                                               (20, 32, 32, 4, "linear"), (4, 64, 64, 3, "linear")])
def test_dopri5_every_kernel_set_forced_sequence_vs_oracle(C, H, HH, nl, interp, gpu_lib):
    """Round 4: the fused attempt kernels (forward: (32, 32, 20) and (64, 64, 4) sets incl. zero-padded shapes; adjoint and taped
    sweep: the (32, 32, 20) set up to three layers) and the per-launch kernels they fall back to, each on the forced step sequence
    (first_step = min_step = max_step: GPU and oracle walk the same steps): forward at TIGHT_Z, adaptive adjoint and
    `adjoint=False` gradients at E2E_G, ragged batch, sequence outputs (dense output inside steps: 0.75 does not divide the knots)."""
    import ctypes
    import gpu_util
    import ncde_amd
    import ncde_oracle as orc
    from ncde_amd import _lib, solver
    B, L = 21, 6
    if interp == "linear":
        coeffs = gu.data.make_rectilinear_coeffs(B, L, C - 1, missing=0.3, seed=95)
        x0 = coeffs[:, 0]
    else:
        coeffs = gu.data.make_cubic_coeffs(B, 2 * L, C - 1, seed=96)
        x0 = coeffs[:, 0, :C]
    p = gu.data.make_field_weights(H, HH, C, seed=9)
    rw = gu.data.make_readin_weights(H, C, 1, seed=9)
    z0n = (x0 @ rw["Wi"].T + rw["bi"]).astype(np.float32)
    opts = {"first_step": 0.75, "min_step": 0.75, "max_step": 0.75}
    field = orc.Field.original(p, H, C, nl)
    ctl = orc.Control(coeffs, interp)
    tt = torch.arange(ctl.n_knots, dtype=torch.float32)
    z = orc.dopri5_forward(ctl, field, z0n, tt, 1e-3, 1e-5, opts)
    gout = (gu.data.normal(31, z.numel(), stream=1).reshape(z.shape) / np.sqrt(z.shape[1])).astype(np.float32)
    dz0, gp = orc.dopri5_adjoint(ctl, field, tt, z, gout, 1e-3, 1e-5, opts)
    _zb, bdz0, bgp = orc.dopri5_discrete_backward(ctl, field, z0n, tt, gout, 1e-3, 1e-5, opts)
    names = ["W0", "b0"] + (["W1", "b1"] if nl > 1 else []) + ["Wo", "bo"]      # (one layer: W1 / b1 exist in `p` but are in no layer)
    if nl == 1:
        gp, bgp = [gp[0], gp[1]] + list(gp[-2:]), [bgp[0], bgp[1]] + list(bgp[-2:])
    X = (ncde_amd.LinearInterpolation if interp == "linear" else ncde_amd.NaturalCubicSpline)(torch.from_numpy(coeffs).cuda())
    layers = [("W0", "b0")] + [("W1", "b1")] * (nl - 1)
    # which kernels
    func = gpu_util.CaseField(p, layers, "cuda")
    prob = solver.build_problem(torch.from_numpy(coeffs).cuda(), interp, torch.from_numpy(z0n).cuda(), func.fused_spec(), "rk4", _lib.OUT_INTERVAL, 0)
    kn = [(_lib.lib().ncde_dopri5_kernel_name(ctypes.byref(prob), k) or b"?").decode() for k in (0, 1, 2)]
    small, mid = max(H, HH) <= 32 and C <= 20, max(H, HH) <= 64 and C <= 4
    assert kn[0].startswith("ncde_dpf_fwd<H32" if small else ("ncde_dpf_fwd<H64" if mid else "ncde_dp_stage")), kn
    assert kn[1].startswith("ncde_dpf_adj" if small and nl <= 3 else "ncde_dp_stage"), kn
    assert kn[2].startswith("ncde_dpf_tape" if small and nl <= 3 else "ncde_dp_tape_backward"), kn
    for adjoint, want_dz0, want_gp in ((True, dz0, gp), (False, bdz0, bgp)):
        func = gpu_util.CaseField(p, layers, "cuda")
        z0 = torch.from_numpy(z0n).cuda().requires_grad_(True)
        out = ncde_amd.cdeint(X, func, z0, X.grid_points, adjoint=adjoint, method="dopri5", rtol=1e-3, atol=1e-5, options=dict(opts))
        assert gu.relerr(out.detach().cpu().numpy(), z) <= TIGHT_Z, (adjoint, gu.relerr(out.detach().cpu().numpy(), z))
        (out * torch.from_numpy(gout).cuda()).sum().backward()
        assert gu.relerr(z0.grad.cpu().numpy(), want_dz0) <= E2E_G, (adjoint, gu.relerr(z0.grad.cpu().numpy(), want_dz0))
        for n_, g_ in zip(names, want_gp):
            assert gu.relerr(func.p[n_].grad.cpu().numpy(), g_) <= E2E_G, (adjoint, n_, gu.relerr(func.p[n_].grad.cpu().numpy(), g_))


def test_dopri5_fused_kernels_at_the_benchmarked_size_replay_and_reproducibility(gpu_lib):
    """The fused attempt kernels at cfg2 dims (B = 4096: 256 workgroups, one per CU; T = 399 knots; ~500 attempts per solve) against
    the per-launch kernels: the per-launch forward / adjoint solves run free with a trace, the fused ones REPLAY those step sequences
    (same attempts, same decisions), so the comparison is at the fixed-step tolerances although the solve is adaptive; then the fused
    solve run free twice: bit-identical outputs and gradients (the controller reads the workgroups' partial sums in a fixed order,
    whichever workgroup arrives last), and the same for adjoint=False."""
    import ncde_amd
    import bench
    c = dict(bench.CONFIGS["cfg2"])
    coeffs = bench.make_inputs(c, c["B"], 0, torch.device("cuda", 0))
    torch.manual_seed(0)
    m = ncde_amd.NeuralCDE(c["C"], c["H"], 1, hidden_hidden_dim=c["HH"], num_layers=c["nl"], interpolation="rectilinear", solver="dopri5").cuda()
    X = ncde_amd.LinearInterpolation(coeffs)
    func = m.func
    with torch.no_grad():
        z0v = m.initial_linear(coeffs[:, 0, :c["C"]]).contiguous()
    gout = torch.randn(c["B"], 2, c["H"], device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) / c["B"]
    kw = dict(method="dopri5", rtol=1e-3, atol=1e-5)

    def run(adjoint, flags, options):
        for q in func.parameters():
            q.grad = None
        z0 = z0v.clone().requires_grad_(True)
        out = ncde_amd.cdeint(X, func, z0, X.interval, adjoint=adjoint, options=dict(options), kernel_flags=flags, **kw)
        tr_f = getattr(func, "dopri5_trace", None)
        (out * gout).sum().backward()
        tr_b = getattr(func, "dopri5_trace_backward", None)
        return out.detach().clone(), z0.grad.clone(), [q.grad.clone() for q in func.parameters()], tr_f, tr_b

    base = {"min_step": 0.5, "_trace": 4096}
    ref = run(True, 1, base)                                      # per-launch kernels, free-running, traced
    assert 300 < len(ref[3]) < 4096 and 300 < len(ref[4]) < 4096
    for q in func.parameters():
        q.grad = None
    z0 = z0v.clone().requires_grad_(True)                        # fused forward replays the forward sequence, fused adjoint the backward one
    out = ncde_amd.cdeint(X, func, z0, X.interval, adjoint=True, options={"min_step": 0.5, "_replay": ref[3][:, 1:3]},
                          adjoint_options={"min_step": 0.5, "_replay": ref[4][:, 1:3]}, **kw)
    assert gu.relerr(out.detach().cpu().numpy(), ref[0].cpu().numpy()) <= TIGHT_Z
    (out * gout).sum().backward()
    # gradients: two fp32 implementations of ~4,000 stage VJPs on 4,096 samples -- a pre-activation within rounding of zero flips a
    # ReLU mask in one of them and moves THAT sample's row by percents (the fixed-step kernels' knife-edge, test_full_size_cfg4_*):
    # >= 99 % of the rows within E2E_G, everything within the full-size bars TOL_DZ0 / 2 TOL_DTHETA (measured: 1.1e-3 / 1.2e-3)
    per = (z0.grad - ref[1]).abs().amax(dim=1) / ref[1].abs().max()
    assert float((per <= E2E_G).float().mean()) >= 0.99 and float(per.max()) <= TOL_DZ0, (float((per <= E2E_G).float().mean()), float(per.max()))
    for q, want in zip(func.parameters(), ref[2]):
        assert gu.relerr(q.grad.cpu().numpy(), want.cpu().numpy()) <= 2 * TOL_DTHETA, gu.relerr(q.grad.cpu().numpy(), want.cpu().numpy())
    for adjoint in (True, False):                                 # free-running, twice: bit-identical
        a, b = run(adjoint, 0, {"min_step": 0.5}), run(adjoint, 0, {"min_step": 0.5})
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and all(torch.equal(x, y) for x, y in zip(a[2], b[2])), adjoint
        assert torch.isfinite(a[1]).all()


def test_neuralcde_module_with_dopri5(gpu_lib):
    """NeuralCDE(solver='dopri5') no longer raises: forward + adaptive adjoint end to end, against the oracle run with the
    module's own parameters (same tolerance logic as above)."""
    import ncde_amd
    import ncde_oracle as orc
    B, L, C, H, HH, nl, OUT = 10, 7, 5, 16, 24, 3, 2
    coeffs = gu.data.make_rectilinear_coeffs(B, L, C - 1, missing=0.3, seed=91)
    torch.manual_seed(5)
    model = ncde_amd.NeuralCDE(C, H, OUT, hidden_hidden_dim=HH, num_layers=nl, interpolation="rectilinear", solver="dopri5",
                               adjoint=True, return_sequences=True).cuda()
    out = model(torch.from_numpy(coeffs).cuda())
    assert out.shape == (B, L, OUT) and torch.isfinite(out).all()
    out.square().sum().backward()
    assert all(torch.isfinite(q.grad).all() for q in model.parameters())
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    field = orc.Field([(sd["func.net_to_hh.0.weight"], sd["func.net_to_hh.0.bias"])] + [(sd["func.net_to_hh.2.weight"], sd["func.net_to_hh.2.bias"])] * (nl - 1),
                      sd["func.tanh_output_layer.0.weight"], sd["func.tanh_output_layer.0.bias"], H, C)
    z0 = torch.from_numpy(coeffs[:, 0]) @ sd["initial_linear.weight"].t() + sd["initial_linear.bias"]
    ctl = orc.Control(coeffs, "linear")
    st = {}
    z = orc.dopri5_forward(ctl, field, z0, torch.arange(ctl.n_knots, dtype=torch.float32), 1e-3, 1e-5, {"min_step": 0.5}, stats=st)
    ref = (z @ sd["final_linear.weight"].t() + sd["final_linear.bias"])[:, ::2]
    e = gu.relerr(out.detach().cpu(), ref)
    print("NeuralCDE dopri5: out vs oracle %.2e, oracle nfe %d, module nfe %d" % (e, st["nfe"], model.nfe))
    assert e <= 2e-2


def test_neuralcde_module_with_dopri5_adjoint_false(gpu_lib):
    """NeuralCDE(solver='dopri5', adjoint=False) -- the model of the reference's "interpolation" experiment grid
    (experiments/configurations/configurations.json5:187-191, setup_model.py:26): runs end to end (forward with the record of its
    accepted steps, reverse sweep incl. the first-step gradient), every parameter of the module receives a finite gradient, and with
    the step sequence pinned (min_step = max_step through cdeint on the module's own field) the gradients agree with the oracle."""
    import ncde_amd
    import ncde_oracle as orc
    B, L, C, H, HH, nl, OUT = 10, 7, 5, 16, 24, 3, 2
    coeffs = gu.data.make_rectilinear_coeffs(B, L, C - 1, missing=0.3, seed=91)
    torch.manual_seed(5)
    model = ncde_amd.NeuralCDE(C, H, OUT, hidden_hidden_dim=HH, num_layers=nl, interpolation="rectilinear", solver="dopri5",
                               adjoint=False, return_sequences=True).cuda()
    x = torch.from_numpy(coeffs).cuda()
    out = model(x)
    assert out.shape == (B, L, OUT) and torch.isfinite(out).all()
    nfe = model.nfe
    out.square().sum().backward()
    assert model.nfe == nfe
    assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in model.parameters())
    # the module's field through cdeint on a pinned step sequence against the oracle
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    field = orc.Field([(sd["func.net_to_hh.0.weight"], sd["func.net_to_hh.0.bias"])] + [(sd["func.net_to_hh.2.weight"], sd["func.net_to_hh.2.bias"])] * (nl - 1),
                      sd["func.tanh_output_layer.0.weight"], sd["func.tanh_output_layer.0.bias"], H, C)
    z0n = (torch.from_numpy(coeffs[:, 0]) @ sd["initial_linear.weight"].t() + sd["initial_linear.bias"]).numpy()
    ctl = orc.Control(coeffs, "linear")
    tt = torch.arange(ctl.n_knots, dtype=torch.float32)
    gout = (gu.data.normal(41, B * ctl.n_knots * H, stream=1).reshape(B, ctl.n_knots, H) / np.sqrt(ctl.n_knots)).astype(np.float32)
    opts = {"min_step": 0.5, "max_step": 0.5}
    z, dz0, gp = orc.dopri5_discrete_backward(ctl, field, z0n, tt, gout, 1e-3, 1e-5, opts)
    X = ncde_amd.LinearInterpolation(x)
    for q in model.func.parameters():
        q.grad = None
    z0 = torch.from_numpy(z0n).cuda().requires_grad_(True)
    o2 = ncde_amd.cdeint(X, model.func, z0, X.grid_points, adjoint=False, method="dopri5", rtol=1e-3, atol=1e-5, options=dict(opts))
    (o2 * torch.from_numpy(gout).cuda()).sum().backward()
    assert gu.relerr(o2.detach().cpu().numpy(), z.numpy()) <= TIGHT_Z
    assert gu.relerr(z0.grad.cpu().numpy(), dz0.numpy()) <= E2E_G
    got = [q.grad.cpu().numpy() for q in model.func.fused_spec().unique_params()]
    for a_, b_ in zip(got, gp):
        assert gu.relerr(a_, b_.numpy()) <= E2E_G


def test_integration_md_stub_with_version1_struct(gpu_lib):
    """The ctypes stub of INTEGRATION.md, verbatim in spirit: a caller that only knows the VERSION-1 struct (no trailing
    field_kind .. br members) drives ncde_forward on a reference-shaped module and gets what cdeint returns."""
    import ctypes
    import ncde_amd
    from ncde_amd import _lib

    class NcdeProblemV1(ctypes.Structure):          # mirrors include/ncde_hip.h up to z0
        _fields_ = [("abi_version", ctypes.c_int32), ("batch", ctypes.c_int32), ("n_knots", ctypes.c_int32),
                    ("channels", ctypes.c_int32), ("hidden", ctypes.c_int32), ("interp", ctypes.c_int32),
                    ("method", ctypes.c_int32), ("output", ctypes.c_int32), ("flags", ctypes.c_uint32),
                    ("n_layers", ctypes.c_int32), ("layer_in", ctypes.c_int32 * 8), ("layer_out", ctypes.c_int32 * 8),
                    ("layer_W", ctypes.c_void_p * 8), ("layer_b", ctypes.c_void_p * 8),
                    ("Wo", ctypes.c_void_p), ("bo", ctypes.c_void_p), ("coeffs", ctypes.c_void_p),
                    ("coeffs_stride_b", ctypes.c_int64), ("coeffs_stride_t", ctypes.c_int64), ("z0", ctypes.c_void_p)]

    assert ctypes.sizeof(NcdeProblemV1) == _lib.NcdeProblem.field_kind.offset
    lib = ctypes.CDLL(_lib.LIB_PATH)
    B, L, C, H, HH, nl = 40, 9, 20, 32, 32, 3
    coeffs = torch.from_numpy(gu.data.make_rectilinear_coeffs(B, L, C - 1, missing=0.3, seed=8)).cuda()
    torch.manual_seed(1)
    func = ncde_amd.OriginalVectorField(C, H, HH, nl).cuda()
    z0 = torch.randn(B, H, device="cuda")
    for every_knot in (False, True):
        p = NcdeProblemV1(abi_version=1, batch=B, n_knots=coeffs.shape[1], channels=C, hidden=H, interp=0, method=2,
                          output=int(every_knot), flags=0)
        lins = [m for m in func.net_to_hh if isinstance(m, torch.nn.Linear)]
        p.n_layers = len(lins)
        for i, m in enumerate(lins):
            p.layer_out[i], p.layer_in[i] = m.weight.shape
            p.layer_W[i], p.layer_b[i] = m.weight.data_ptr(), m.bias.data_ptr()
        out_lin = func.tanh_output_layer[0]
        p.Wo, p.bo = out_lin.weight.data_ptr(), out_lin.bias.data_ptr()
        p.coeffs, p.coeffs_stride_b, p.coeffs_stride_t = coeffs.data_ptr(), coeffs.stride(0), coeffs.stride(1)
        p.z0 = z0.data_ptr()
        lib.ncde_workspace_bytes.restype = ctypes.c_int64
        ws = torch.empty(max(lib.ncde_workspace_bytes(ctypes.byref(p), 0), 256), dtype=torch.uint8, device="cuda")
        n_out = coeffs.shape[1] if every_knot else 2
        out = torch.empty(B, n_out, H, device="cuda")
        rc = lib.ncde_forward(ctypes.byref(p), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
                              ctypes.c_size_t(ws.numel()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        X = ncde_amd.LinearInterpolation(coeffs)
        with torch.no_grad():
            ref = ncde_amd.cdeint(X, func, z0, X.grid_points if every_knot else X.interval, method="rk4", options={"step_size": 1})
        torch.cuda.synchronize()
        assert torch.equal(out, ref)


def test_training_beyond_the_fused_backward_kernels_runs_on_the_unfused_solver(gpu_lib):
    """NeuralCDE(hidden_dim=256) -- inside the reference's hyper-parameter range (configurations.json5:34-35,
    adjoint: false) -- used to pass its forward and raise inside loss.backward().  cdeint now asks for the backward kernel before the
    forward and routes the call to the unfused torch-op solver with a warning naming the shape; the result is the solve the oracle
    computes (forward and both kinds of gradient), and evaluation under no_grad keeps the fused forward kernel."""
    import warnings
    import ncde_amd
    import ncde_oracle as orc
    from ncde_amd import unfused
    B, L, C, H, HH, nl = 9, 5, 6, 256, 272, 2      # (round 5: hidden widths up to 256 have fused backward kernels; 272 does not)
    coeffs = gu.data.make_rectilinear_coeffs(B, L, C - 1, missing=0.3, seed=77)
    for adjoint in (False, True):
        torch.manual_seed(3)
        model = ncde_amd.NeuralCDE(C, H, 3, hidden_hidden_dim=HH, num_layers=nl, interpolation="rectilinear", adjoint=adjoint).cuda()
        x = torch.from_numpy(coeffs).cuda()
        unfused._WARNED.clear()
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            out = model(x)
        assert any("unfused torch-op solver" in str(w.message) and "hidden=256" in str(w.message) for w in rec), [str(w.message) for w in rec]
        out.square().sum().backward()      # (used to raise NotImplementedError here)
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
        # same numbers as the oracle's restatement of the reference
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        field = orc.Field([(sd["func.net_to_hh.0.weight"], sd["func.net_to_hh.0.bias"])] + [(sd["func.net_to_hh.2.weight"], sd["func.net_to_hh.2.bias"])] * (nl - 1),
                          sd["func.tanh_output_layer.0.weight"], sd["func.tanh_output_layer.0.bias"], H, C)
        z0 = torch.from_numpy(coeffs[:, 0]) @ sd["initial_linear.weight"].t() + sd["initial_linear.bias"]
        ctl = orc.Control(coeffs, "linear")
        z = orc.solve_forward(ctl, field, z0, "rk4", False)
        ref = z[:, -1] @ sd["final_linear.weight"].t() + sd["final_linear.bias"]
        assert gu.relerr(out.detach().cpu().numpy(), ref.numpy()) <= 2e-5
        gz = torch.zeros_like(z)
        gz[:, -1] = (2 * ref) @ sd["final_linear.weight"]
        if adjoint:
            dz0, gp = orc.solve_adjoint(ctl, field, z, gz, "rk4", False)
        else:
            dz0, gp = orc.solve_discrete_backward(ctl, field, z0, gz, "rk4", False)[-2:]
        got = model.func.tanh_output_layer[0].weight.grad.cpu().numpy()
        assert gu.relerr(got, np.asarray(gp[-2])) <= 2e-4
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("error")      # no gradient needed: the fused forward kernel, no warning
            out2 = model(x)
        assert gu.relerr(out2.cpu().numpy(), ref.numpy()) <= 2e-5


@pytest.mark.parametrize("adjoint", [False, True])
def test_neuralcde_hidden_256_trains_on_the_fused_kernels(adjoint, gpu_lib):
    """VERDICT round 4, item 1 (done criterion): NeuralCDE(hidden_dim=256, hidden_hidden_dim=196) -- the corner of the reference's
    hyper-parameter range (configurations.json5:34-35), `adjoint: false` as its experiments train -- runs forward and backward on fused,
    non-generic kernels (no unfused-solver warning) and gets the oracle's numbers."""
    import warnings
    import ncde_amd
    import ncde_oracle as orc
    from ncde_amd import unfused
    B, L, C, H, HH, nl = 9, 5, 6, 256, 196, 2
    coeffs = gu.data.make_rectilinear_coeffs(B, L, C - 1, missing=0.3, seed=78)
    torch.manual_seed(4)
    model = ncde_amd.NeuralCDE(C, H, 3, hidden_hidden_dim=HH, num_layers=nl, interpolation="rectilinear", adjoint=adjoint).cuda()
    x = torch.from_numpy(coeffs).cuda()
    unfused._WARNED.clear()
    with warnings.catch_warnings():
        warnings.filterwarnings("error", message=".*unfused.*")
        out = model(x)
        out.square().sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    field = orc.Field([(sd["func.net_to_hh.0.weight"], sd["func.net_to_hh.0.bias"])] + [(sd["func.net_to_hh.2.weight"], sd["func.net_to_hh.2.bias"])] * (nl - 1),
                      sd["func.tanh_output_layer.0.weight"], sd["func.tanh_output_layer.0.bias"], H, C)
    z0 = torch.from_numpy(coeffs[:, 0]) @ sd["initial_linear.weight"].t() + sd["initial_linear.bias"]
    ctl = orc.Control(coeffs, "linear")
    z = orc.solve_forward(ctl, field, z0, "rk4", False)
    ref = z[:, -1] @ sd["final_linear.weight"].t() + sd["final_linear.bias"]
    assert gu.relerr(out.detach().cpu().numpy(), ref.numpy()) <= 2e-5
    gz = torch.zeros_like(z)
    gz[:, -1] = (2 * ref) @ sd["final_linear.weight"]
    if adjoint:
        dz0, gp = orc.solve_adjoint(ctl, field, z, gz, "rk4", False)
    else:
        dz0, gp = orc.solve_discrete_backward(ctl, field, z0, gz, "rk4", False)[-2:]
    got = model.func.tanh_output_layer[0].weight.grad.cpu().numpy()
    assert gu.relerr(got, np.asarray(gp[-2])) <= 2e-4, gu.relerr(got, np.asarray(gp[-2]))
    got0 = model.func.net_to_hh[0].weight.grad.cpu().numpy()
    assert gu.relerr(got0, np.asarray(gp[0])) <= 2e-4, gu.relerr(got0, np.asarray(gp[0]))


@pytest.mark.parametrize("B,L,C,H,HH,nl,interp,method,seq", [
    (256, 3, 20, 128, 128, 3, "cubic", "midpoint", True),   # two groups of 8: the group of workgroup 1 gives up, the other one runs on until it checks
    (123, 4, 40, 64, 128, 2, "linear", "rk4", False),       # one group, ragged last tile
])
def test_cooperative_timeout_is_reexecuted_not_nan(B, L, C, H, HH, nl, interp, method, seq, gpu_lib):
    """Round 6: a cooperative launch whose workgroups do not all arrive must neither hang nor
    return NaN / half-finished gradients.  NCDE_FLAG_COOP_FAULT_INJECT withholds ONE arrival of workgroup 1 in the first cooperative
    launch (and shortens the spin limit): that launch gives up, sets the call's status word, the cooperative launches of the later time
    windows return at once, and the per-workgroup kernels enqueued behind them redo the pass -- every output is bit-identical to a
    NCDE_FLAG_NO_COOP call.  The Python host then sees the word, warns once and stops asking for cooperative launches."""
    import gpu_util
    import ncde_amd
    from ncde_amd import _lib, solver
    INJ, NOC = _lib.FLAG_COOP_FAULT_INJECT, _lib.FLAG_NO_COOP
    case = _seeded_case(interp, method, seq, B=B, L=L, C=C, H=H, HH=HH, nl=nl, seed=940 + C)
    ex = case["expect"]
    assert all("coop" in k for k in gpu_util.kernel_names(case, flags=INJ))
    dev = torch.cuda.current_device()
    solver._COOP_DISABLED.clear()
    try:
        # the library alone (C-ABI): forward, continuous adjoint, exact discrete backward, several time windows
        for win in (0, _lib.FLAG_TILED_WINDOW_STEPS(1)):
            ref = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=NOC | win)
            got = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=INJ | win)
            assert gpu_util.coop_status_word(case, 1, flags=INJ | win) == 1
            assert np.array_equal(got["dz0"], ref["dz0"]) and np.isfinite(got["dz0"]).all()
            for k in ref["grads"]:
                assert np.array_equal(got["grads"][k], ref["grads"][k]), k
            ok = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=win)      # the same call without the fault: word stays 0
            assert gpu_util.coop_status_word(case, 1, flags=win) == 0
            assert gu.relerr(ok["dz0"], ref["dz0"]) <= 2e-5
        refd = gpu_util.run_adjoint_direct(case, ex["z_out"], stages=case["stage_record"], flags=NOC)
        gotd = gpu_util.run_adjoint_direct(case, ex["z_out"], stages=case["stage_record"], flags=INJ)
        assert gpu_util.coop_status_word(case, 2, flags=INJ) == 1
        assert np.array_equal(gotd["dz0"], refd["dz0"]) and all(np.array_equal(gotd["grads"][k], refd["grads"][k]) for k in refd["grads"])
        # through cdeint (forward + backward, both with the fault): values, then the host's reaction
        solver._COOP_DISABLED.clear()
        solver._COOP_PENDING.clear()
        reff = gpu_util.run_case(case, flags=NOC)
        gotf = gpu_util.run_case(case, flags=INJ)
        assert np.array_equal(gotf["z_out"], reff["z_out"]) and np.array_equal(gotf["dz0"], reff["dz0"])
        for k in reff["grads"]:
            assert np.array_equal(gotf["grads"][k], reff["grads"][k]), k
        with pytest.warns(RuntimeWarning, match="cooperative"):
            st = ncde_amd.coop_status(wait=True)
        assert st.get(dev, 0) >= 1
        assert solver._coop_flags(0, torch.device("cuda", dev)) & NOC
        again = gpu_util.run_case(case)      # later calls of this process: per-workgroup kernels, no new warning
        assert np.array_equal(again["z_out"], reff["z_out"])
    finally:
        ncde_amd.coop_status(wait=True)
        solver._COOP_DISABLED.clear()
    # without the fault nothing is recorded
    gpu_util.run_case(case)
    assert ncde_amd.coop_status(wait=True) == {}


def test_cooperative_kernels_on_more_sample_tiles_than_cus(gpu_lib):
    """Round 6: a batch with more 16-sample tiles than the device has CUs cannot be ONE cooperative launch (every
    workgroup must be resident); the library runs it as several launches over chunks of the batch -- here 288 tiles = 256 + 32 --, the
    hidden-layer partials per workgroup, ncde_dwo_h2 adding the chunks' output-layer gradients up.  Forward, continuous adjoint and exact
    discrete backward against the oracle and against the per-workgroup kernels; bit-reproducible."""
    import gpu_util
    from ncde_amd import _lib
    case = _seeded_case("linear", "rk4", False, B=4608, L=2, C=80, H=128, HH=128, nl=3, seed=960)
    ex = case["expect"]
    names = gpu_util.kernel_names(case)
    assert all("coop" in k for k in names), names
    fw = gpu_util.run_case(case, need_grads=False)
    assert gu.relerr(fw["z_out"], ex["z_out"]) <= TIGHT_Z, gu.relerr(fw["z_out"], ex["z_out"])
    assert gpu_util.coop_status_word(case, 0) == 0
    iso = gpu_util.run_adjoint_direct(case, ex["z_out"])
    assert gpu_util.coop_status_word(case, 1) == 0
    old = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=_lib.FLAG_NO_COOP)
    # (4608 samples: a handful have a pre-activation within rounding of zero, where any two fp32 implementations may disagree on a ReLU
    # mask and THAT sample's dz0 row moves -- the per-workgroup sweep deviates from the oracle on such rows just the same: per row)
    def rows_off(x, y):
        per = np.abs(x - y).max(axis=1) / np.abs(y).max()
        return int((per > TIGHT_G).sum()), float(per.max())
    for k, e in _grad_errors(case, iso).items():
        if k == "dz0":
            n_off, worst = rows_off(iso["dz0"], ex["dz0"])
            assert n_off <= 8 and worst <= TOL_DZ0, ("cooperative adjoint in two chunks", n_off, worst, rows_off(old["dz0"], ex["dz0"]))
        else:      # (sums over 4608 samples x 16 stages: the oracle's own summation order is worth a few 1e-5 here -- the tight bar is the
            assert e <= E2E_G, ("cooperative adjoint in two chunks", k, e)      # comparison with the per-workgroup kernels below)
    n_off, worst = rows_off(iso["dz0"], old["dz0"])
    assert n_off <= 8 and worst <= TOL_DZ0, (n_off, worst)
    for k in iso["grads"]:      # (the handful of mask-flip samples also moves the batch sums by their share: 1e-4, not 2e-5)
        assert gu.relerr(iso["grads"][k], old["grads"][k]) <= 1e-4, k
    again = gpu_util.run_adjoint_direct(case, ex["z_out"])
    assert np.array_equal(again["dz0"], iso["dz0"]) and all(np.array_equal(again["grads"][k], iso["grads"][k]) for k in iso["grads"])
    isod = gpu_util.run_adjoint_direct(case, ex["z_out"], stages=case["stage_record"])
    oldd = gpu_util.run_adjoint_direct(case, ex["z_out"], stages=case["stage_record"], flags=_lib.FLAG_NO_COOP)
    n_off, worst = rows_off(isod["dz0"], oldd["dz0"])
    assert n_off <= 8 and worst <= TOL_DZ0, (n_off, worst)
    for k in isod["grads"]:
        assert gu.relerr(isod["grads"][k], oldd["grads"][k]) <= 1e-4, k
    fd = gpu_util.run_case(case, adjoint=False)      # the recording forward in chunks writes ONE record with the batch stride of the whole batch
    assert np.array_equal(fd["z_out"], fw["z_out"])
    # (end to end the backward runs on the GPU forward's own stage record: last-bit differences of z flip ReLU masks in a few of the 4608
    # samples, and a batch SUM with cancellation moves by such a sample's whole share -- 3e-3 measured; the isolated passes above are tight)
    for k, e in _grad_errors(case, fd, "bp_").items():
        assert e <= 1e-2, ("cooperative forward + discrete backward in chunks, end to end", k, e)


def test_cooperative_calls_on_two_streams_do_not_wait_for_each_other(gpu_lib):
    """two cooperative launches on different streams could each hold CUs the other needs.  The library lets only ONE
    cooperative sequence per device be in flight: a call that finds another stream's sequence unfinished runs the per-workgroup
    kernels.  Here: two streams, interleaved calls, every result equal (to fp32 round-off) to the single-stream one."""
    import gpu_util
    from ncde_amd import _lib
    case = _seeded_case("linear", "rk4", False, B=512, L=6, C=80, H=128, HH=128, nl=3, seed=951)
    assert "coop" in gpu_util.kernel_names(case)[0]
    ref = gpu_util.run_case(case, need_grads=False, flags=_lib.FLAG_NO_COOP)["z_out"]
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for it in range(3):
        for st in (s1, s2):
            with torch.cuda.stream(st):
                outs.append(gpu_util.run_case_async(case))
    torch.cuda.synchronize()
    for o in outs:
        assert gu.relerr(o.cpu().numpy(), ref) <= TIGHT_Z


@pytest.mark.parametrize("B,L,C,H,HH,nl,interp,method,seq", [
    (123, 3, 40, 64, 128, 2, "linear", "rk4", False),       # 8 sample tiles = ONE group of 8 members (2 state-unit blocks each), ragged last tile
    (256, 3, 20, 128, 128, 3, "cubic", "midpoint", True),   # 16 tiles = two groups of 8 (4 blocks per member, one wave each), sequence outputs
    (512, 2, 80, 128, 128, 3, "linear", "rk4", True),       # cfg5 dims: 32 tiles = one group of 32 (one block per member)
])
def test_cooperative_output_phase_vs_oracle(B, L, C, H, HH, nl, interp, method, seq, gpu_lib):
    """Round 5: the XCD-cooperative, weight-stationary output phase of the batch-tiled sweep (csrc/ncde_coop.h) --
    each workgroup keeps 20 row tiles of Wo in registers and applies them to every sample tile of its group, activations and partial
    sums travel through L2 with write-through stores / L1-bypassing loads and monotonic group counters.  Continuous adjoint and exact
    discrete backward on the oracle's forward solution at the tight tolerances, against the per-workgroup sweep (NCDE_FLAG_NO_COOP),
    and bit-reproducible run to run."""
    import gpu_util
    from ncde_amd import _lib
    case = _seeded_case(interp, method, seq, B=B, L=L, C=C, H=H, HH=HH, nl=nl, seed=900 + C)
    ex = case["expect"]
    names = gpu_util.kernel_names(case)
    assert "coop" in names[0] and "coop" in names[1] and "coop" in names[2], names
    assert not any("coop" in k for k in gpu_util.kernel_names(case, flags=_lib.FLAG_NO_COOP))
    # the forward's cooperative output phase (ncde_fwd_tiled<.., COOP>): against the oracle, against the per-workgroup kernel, run to run,
    # and as the recording forward of adjoint=False (same kernel: the same z bit for bit)
    fw = gpu_util.run_case(case, need_grads=False)
    assert "coop" in fw["kernels"][0] and gu.relerr(fw["z_out"], ex["z_out"]) <= TIGHT_Z, (fw["kernels"], gu.relerr(fw["z_out"], ex["z_out"]))
    fo = gpu_util.run_case(case, need_grads=False, flags=_lib.FLAG_NO_COOP)
    assert "coop" not in fo["kernels"][0] and gu.relerr(fw["z_out"], fo["z_out"]) <= TIGHT_Z
    assert np.array_equal(gpu_util.run_case(case, need_grads=False)["z_out"], fw["z_out"])
    fd = gpu_util.run_case(case, adjoint=False)
    assert np.array_equal(fd["z_out"], fw["z_out"])
    if B != 512:
        for k, e in _grad_errors(case, fd, "bp_").items():
            assert e <= (TOL_DZ0 if k == "dz0" else TOL_DTHETA), ("cooperative forward + discrete backward end to end", k, e)
    iso = gpu_util.run_adjoint_direct(case, ex["z_out"])
    for k, e in _grad_errors(case, iso).items():
        assert e <= TIGHT_G, ("cooperative adjoint", k, e)
    old = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=_lib.FLAG_NO_COOP)
    assert gu.relerr(iso["dz0"], old["dz0"]) <= 2e-5
    for k in iso["grads"]:
        assert gu.relerr(iso["grads"][k], old["grads"][k]) <= 2e-5, k
    again = gpu_util.run_adjoint_direct(case, ex["z_out"])
    assert np.array_equal(again["dz0"], iso["dz0"]) and all(np.array_equal(again["grads"][k], iso["grads"][k]) for k in iso["grads"])
    isod = gpu_util.run_adjoint_direct(case, ex["z_out"], stages=case["stage_record"])
    oldd = gpu_util.run_adjoint_direct(case, ex["z_out"], stages=case["stage_record"], flags=_lib.FLAG_NO_COOP)
    assert gu.relerr(isod["dz0"], oldd["dz0"]) <= 2e-5
    for k in isod["grads"]:
        assert gu.relerr(isod["grads"][k], oldd["grads"][k]) <= 2e-5, k
    if B != 512:      # (the 512-sample case has ONE sample whose ReLU mask flips between any two fp32 implementations -- both sweeps deviate
        for k, e in _grad_errors(case, isod, "bp_").items():      # from the oracle there by the same 1e-3 --)
            assert e <= TIGHT_G, ("cooperative discrete backward", k, e)
    # several time windows (records of two steps at a time): the state, the hidden-layer partial and the group counters carry over
    win = gpu_util.run_adjoint_direct(case, ex["z_out"], flags=_lib.FLAG_TILED_WINDOW_STEPS(2))
    assert gu.relerr(win["dz0"], iso["dz0"]) <= 1e-6
    for k in iso["grads"]:
        assert gu.relerr(win["grads"][k], iso["grads"][k]) <= 2e-6, k
