#!/usr/bin/env python
"""Benchmark of the Neural-CDE hot path on MI355X (contract: see the task statement / DESIGN.md §Measurement).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic input already resident in HBM:
forward solve (fused kernel) + loss + continuous-adjoint backward (fused kernel) + [N>1: one RCCL
all-reduce of the flat gradient] + Adam update, on BASELINE.json configs[1]/[2] (B=4096 per GPU, 200 raw
observations -> 399 rectilinear knots, 20 channels incl. time, H=HH=32, nl=3, RK4-3/8, step 1).
`value` = sample-steps/s = (samples processed by all ranks) * (T-1) / time, fp32 throughout.
The headline line is WEAK scaling (B=4096 per GPU, the sharding rule of the task statement); for N>1 the same run also
times the north-star's STRONG-scaling workload (global B=4096 sharded over the N ranks) and reports it under "strong".
Inputs: raw synthetic series (ncde_amd.data) turned into coefficients by the GPU builders (csrc/ncde_prepare.hip).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def _self_launch():
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start `python -m torch.distributed.run
    --nproc-per-node N bench.py ...` as a CHILD process -- before this process has imported torch or touched the GPU (never an
    exec) --, relay rank 0's JSON line and exit with the child's return code."""
    if "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return
    n = 1
    for i, a in enumerate(sys.argv[1:], 1):
        if a == "--gpus" and i + 1 < len(sys.argv):
            n = int(sys.argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1:
        return
    import socket
    import subprocess
    with socket.socket() as sk:      # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    child = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = [ln for ln in child.stdout.splitlines() if ln.strip()]
    js = [ln for ln in lines if ln.lstrip().startswith("{") and "\"metric\"" in ln]
    for ln in lines:
        if ln not in js:
            sys.stderr.write(ln + "\n")
    if js:
        print(js[-1])
    sys.exit(child.returncode if child.returncode != 0 or js else 1)


if __name__ == "__main__":
    _self_launch()

import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, ROOT)
import ncde_amd  # noqa: E402
from ncde_amd import _lib, distributed as D, solver  # noqa: E402

CONFIGS = {
    # name: (B per GPU, raw length, channels incl. time, H, HH, nl, interpolation, solver, missing)
    "cfg2": dict(B=4096, L=200, C=20, H=32, HH=32, nl=3, interpolation="rectilinear", solver="rk4", missing=0.3),
    "cfg4": dict(B=8192, L=182, C=4, H=64, HH=64, nl=3, interpolation="cubic", solver="midpoint", missing=0.0),
    "cfg5": dict(B=4096, L=400, C=80, H=128, HH=128, nl=3, interpolation="rectilinear", solver="rk4", missing=0.6),
}
PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 vector == fp32-input MFMA peak
PEAK_HBM_GBS = 8000.0
CPU_REPS = 3               # cpu_baseline: 1 warm-up + median of CPU_REPS timed runs


def stages_of(method):
    return {"rk4": 4, "midpoint": 2, "euler": 1}[method]


def flops_forward_per_sample_step(c):
    """S * 2 * (H*HH + (nl-1)*HH^2 + HH*H*C + H*C)."""
    return stages_of(c["solver"]) * 2 * (c["H"] * c["HH"] + (c["nl"] - 1) * c["HH"] ** 2 + c["HH"] * c["H"] * c["C"] + c["H"] * c["C"])


def bytes_forward_per_sample_step(c):
    return (12 if c["interpolation"] == "cubic" else 4) * c["C"]


def make_inputs(c, B, offset, dev):
    """Deterministic raw series [B, L, C] (host generator, shardable by `offset`) -> control-path coefficients on `dev`
    by the product's own GPU builders (bit-identical to the reference's builders, tests/golden/g8)."""
    x = ncde_amd.data.synthetic_series(B, c["L"], c["C"] - 1, missing=c["missing"], seed=1234, batch_offset=offset)
    x = torch.from_numpy(x).to(dev)
    if c["interpolation"] == "rectilinear":
        return ncde_amd.linear_interpolation_coeffs(x, rectilinear=0)
    if c["interpolation"] == "cubic":
        return ncde_amd.natural_cubic_coeffs(x)
    return x


def make_model(c, device):
    model = ncde_amd.NeuralCDE(c["C"], c["H"], 1, hidden_hidden_dim=c["HH"], num_layers=c["nl"],
                               interpolation=c["interpolation"], adjoint=True, solver=c["solver"])
    fw = ncde_amd.data.make_field_weights(c["H"], c["HH"], c["C"], seed=0)
    rw = ncde_amd.data.make_readin_weights(c["H"], c["C"], 1, seed=0)
    sd = {"initial_linear.weight": rw["Wi"], "initial_linear.bias": rw["bi"],
          "final_linear.weight": rw["Wf"], "final_linear.bias": rw["bf"],
          "func.net_to_hh.0.weight": fw["W0"], "func.net_to_hh.0.bias": fw["b0"],
          "func.tanh_output_layer.0.weight": fw["Wo"], "func.tanh_output_layer.0.bias": fw["bo"]}
    for i in range(1, c["nl"]):
        sd[f"func.net_to_hh.{2 * i}.weight"] = fw["W1"]
        sd[f"func.net_to_hh.{2 * i}.bias"] = fw["b1"]
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return model.to(device), fw, rw


def time_kernels(model, c, coeffs, iters=3):
    """HIP-event timing (on the launch stream, inside the C-ABI) of the forward and adjoint kernels."""
    dev = coeffs.device
    spec = model.func.fused_spec()
    interp = "cubic" if c["interpolation"] == "cubic" else "linear"
    with torch.no_grad():
        z0 = model.initial_linear(coeffs[:, 0, :c["C"]]).contiguous()
    p = solver.build_problem(coeffs, interp, z0, spec, c["solver"], _lib.OUT_INTERVAL, 0)
    lib = _lib.lib()
    out = torch.empty(z0.shape[0], 2, c["H"], device=dev)
    ws0 = solver._workspace(p, 0, dev)
    ms = ctypes.c_float(0)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.ncde_time_kernel(ctypes.byref(p), 0, out.data_ptr(), None, None, ws0.data_ptr(), ws0.numel(), stream,
                                    iters, ctypes.byref(ms)), "time fwd")
    ms_fwd = ms.value
    gout = torch.randn_like(out)
    uniq = spec.unique_params()
    gbuf = {id(q): torch.empty_like(q) for q in uniq}
    g = _lib.NcdeGrads()
    gz0 = torch.empty_like(z0)
    g.grad_z0 = gz0.data_ptr()
    for i, (w, b) in enumerate(spec.layers):
        g.grad_layer_W[i], g.grad_layer_b[i] = gbuf[id(w)].data_ptr(), gbuf[id(b)].data_ptr()
    g.grad_Wo, g.grad_bo = gbuf[id(spec.Wo)].data_ptr(), gbuf[id(spec.bo)].data_ptr()
    ws1 = solver._workspace(p, 1, dev)
    _lib.check(lib.ncde_time_kernel(ctypes.byref(p), 1, out.data_ptr(), gout.data_ptr(), ctypes.byref(g), ws1.data_ptr(),
                                    ws1.numel(), stream, iters, ctypes.byref(ms)), "time adj")
    names = ((lib.ncde_kernel_name(ctypes.byref(p), 0) or b"?").decode(), (lib.ncde_kernel_name(ctypes.byref(p), 1) or b"?").decode())
    return ms_fwd, ms.value, names


def golden_z_err(model, c, coeffs):
    """max |z_T - z_T(reference)| / max |z_T(reference)| of the DEFAULT forward kernel (the one the step times) on the full cfg2
    workload, against the reference's own z_T (tests/golden/g5_cfg2_full.npz, written by oracle/gen_golden.py from the imported
    reference on the same deterministic inputs and weights).  Outside the timed region, before any optimizer step."""
    path = os.path.join(ROOT, "tests", "golden", "g5_cfg2_full.npz")
    if not os.path.exists(path):
        return None
    ref = np.load(path)["zT"]
    if ref.shape != (coeffs.shape[0], c["H"]):
        return None
    spec = model.func.fused_spec()
    with torch.no_grad():
        z0 = model.initial_linear(coeffs[:, 0, :c["C"]]).contiguous()
    p = solver.build_problem(coeffs, "linear", z0, spec, c["solver"], _lib.OUT_INTERVAL, 0)
    out = torch.empty(z0.shape[0], 2, c["H"], device=coeffs.device)
    ws = solver._workspace(p, 0, coeffs.device)
    _lib.check(_lib.lib().ncde_forward(ctypes.byref(p), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "ncde_forward")
    zT = out[:, -1].cpu().numpy()
    return float(np.abs(zT - ref).max() / np.abs(ref).max())


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota (the GPU box
    shows 256 logical CPUs but grants 16; spinning 256 OpenMP threads on 16 CPUs takes minutes per solve)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(c, fw, rw, sample_B):
    """The CPU restatements of the reference op sequence (both pinned to the reference through the golden fixtures), timed
    on this host's cores on a bounded sample, forward + adjoint:  (1) oracle/ncde_cpu.cpp behind the SAME C-ABI as the HIP
    library (scalar C++ + OpenMP over samples);  (2) oracle/ncde_oracle.py (torch-CPU ops, what the reference itself runs
    on).  The faster of the two is reported as the baseline, the other one beside it."""
    sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import coeff_oracle
    import ncde_oracle as orc
    torch.set_num_threads(host_cores())
    x = ncde_amd.data.synthetic_series(sample_B, c["L"], c["C"] - 1, missing=c["missing"], seed=1234)
    coeffs = coeff_oracle.natural_cubic_coeffs(x) if c["interpolation"] == "cubic" else \
        (coeff_oracle.rectilinear_prep(x, 0) if c["interpolation"] == "rectilinear" else x)
    cabi = None
    try:
        import cpu_lib_util as cu
        cu.cpu_lib().ncde_cpu_set_threads(host_cores())
        z0n = (coeffs[:, 0, :c["C"]] @ rw["Wi"].T + rw["bi"]).astype(np.float32)
        layers = [("W0", "b0")] + [("W1", "b1")] * (c["nl"] - 1)
        cc = cu.CpuCase(coeffs, "cubic" if c["interpolation"] == "cubic" else "linear", z0n, fw, layers, c["solver"], False)
        gones = np.ones((sample_B, 2, c["H"]), np.float32)
        cc.backward(cc.forward(), gones)      # warm-up (first touch, thread pool)
        runs = []
        for _ in range(CPU_REPS):
            t0 = time.time()
            zc = cc.forward()
            t1 = time.time()
            cc.backward(zc, gones)
            t2 = time.time()
            runs.append((t2 - t0, t1 - t0))
        tot, fwd = sorted(runs)[len(runs) // 2]      # median of CPU_REPS by total time
        n_k = coeffs.shape[1] + (1 if c["interpolation"] == "cubic" else 0)
        cabi = {"value": sample_B * (n_k - 1) / tot, "forward_only": sample_B * (n_k - 1) / fwd, "seconds": tot}
    except OSError:
        pass
    kind = "cubic" if c["interpolation"] == "cubic" else "linear"
    field = orc.Field.original(fw, c["H"], c["C"], c["nl"])
    ctl = orc.Control(coeffs, kind)
    z0 = torch.from_numpy(coeffs[:, 0, :c["C"]]) @ torch.from_numpy(rw["Wi"]).t() + torch.from_numpy(rw["bi"])
    gout = torch.ones(sample_B, 2, c["H"])
    zw = orc.solve_forward(orc.Control(coeffs[:32], kind), field, z0[:32], c["solver"], False)  # warm-up: thread pool, allocator
    orc.solve_adjoint(orc.Control(coeffs[:32], kind), field, zw, gout[:32], c["solver"], False)
    runs = []
    for _ in range(CPU_REPS):
        t0 = time.time()
        z = orc.solve_forward(ctl, field, z0, c["solver"], False)
        t1 = time.time()
        orc.solve_adjoint(ctl, field, z, gout, c["solver"], False)
        t2 = time.time()
        runs.append((t2 - t0, t1 - t0))
    tot, fwd = sorted(runs)[len(runs) // 2]
    steps = sample_B * (ctl.n_knots - 1)
    torch_port = {"value": steps / tot, "forward_only": steps / fwd, "seconds": tot}
    best, other = (cabi, torch_port) if cabi and cabi["value"] >= torch_port["value"] else (torch_port, cabi)
    name = {id(cabi): "oracle/ncde_cpu.cpp (C-ABI restatement, C++ + OpenMP)", id(torch_port): "oracle/ncde_oracle.py (torch-CPU ops)"}
    rec = {"value": best["value"], "unit": "sample-steps/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": "%s forward+adjoint on B=%d of the same workload (T=%d), 1 warm-up + median of %d runs, %.1f s per run; "
                     "forward only: %.3e sample-steps/s" % (name[id(best)], sample_B, ctl.n_knots, CPU_REPS, best["seconds"], best["forward_only"])}
    if other:
        rec["other_port"] = {"impl": name[id(other)], "value": other["value"], "forward_only": other["forward_only"]}
    return rec


class Workload:
    """One rank's shard of a configuration: inputs resident in HBM, model, bucket, optimizer."""

    def __init__(self, c, B_local, B_total, lo, dev):
        self.c, self.B_local, self.B_total = c, B_local, B_total
        self.coeffs = make_inputs(c, B_local, lo, dev)
        self.T = self.coeffs.shape[1] + (1 if c["interpolation"] == "cubic" else 0)
        self.labels = (torch.from_numpy(ncde_amd.data.uniform01(7, B_total, stream=5)[lo:lo + B_local]) > 0.5).float().to(dev).unsqueeze(1)
        self.model, self.fw, self.rw = make_model(c, dev)
        self.bucket = D.FlatGradAllReduce(self.model.parameters())
        try:     # single-kernel Adam: ~0.5 ms less launch overhead per step than the foreach implementation
            self.opt = torch.optim.Adam(self.model.parameters(), lr=1e-3, fused=True)
        except (TypeError, RuntimeError):
            self.opt = torch.optim.Adam(self.model.parameters(), lr=1e-3)
        self.loss_fn = torch.nn.BCEWithLogitsLoss()

    def step(self):
        return D.train_step(self.model, self.bucket, self.opt, self.coeffs, self.labels, self.loss_fn)

    def timed(self, steps, warmup, world, dev):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks (seconds)."""
        for _ in range(warmup):
            self.step()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = self.step()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        assert torch.isfinite(loss).item(), "training diverged"
        return float(tmax.item()), float(loss)


def _pmc_summary(config, B_local):
    """The committed rocprofv3 PMC summary of this config (bench.py cannot run the profiler on itself) -- only if it was taken on
    exactly the kernel sources this library is built from (fingerprint) and on this workload; otherwise None."""
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        path = os.path.join(ROOT, "profiles", "%s_pmc_%s_summary.json" % (rnd, config))
        try:
            with open(path) as fh:
                pmc = json.load(fh)
        except (OSError, ValueError):
            continue
        meta = pmc.get("_meta", {})
        if meta.get("source_fingerprint") == _lib.source_fingerprint() and meta.get("batch") == B_local:
            return pmc, os.path.relpath(path, ROOT)
        return None, None
    return None, None


def pmc_pass(name, config, B_local):
    """Counters of ONE forward / backward pass: `name` is the C-ABI's kernel name of the pass, "a<...>+b<...>" when the pass is
    several kernels (the windowed batch-tiled backward launches its sweep and its output-layer gradient kernel once per time
    window).  HBM bytes are SUMMED over every launch of every one of them in a pass (`*_per_pass` of tools/pmc_summary.py);
    MFMA-busy / VALU-active are those of the kernel with the most wave cycles.  None where no valid summary is committed."""
    pmc, src = _pmc_summary(config, B_local)
    if pmc is None:
        return None
    bases = [part.split("<")[0] for part in name.replace("+ncde_", "\x00ncde_").split("\x00")]      # ("+" also occurs inside template tags)
    recs = [pmc[b] for b in bases if b in pmc]
    if len(recs) != len(bases) or not all("hbm_read_MB_per_pass_corrected_x2" in r for r in recs):
        return None
    dom = max(recs, key=lambda r: r.get("SQ_WAVE_CYCLES", 0.0) * r.get("launches_per_pass", 1.0))
    per_pass = lambda key: sum(r.get(key, 0.0) * r.get("launches_per_pass", 1.0) for r in recs)      # noqa: E731
    return {"traffic": round(sum(r["hbm_read_MB_per_pass_corrected_x2"] + r.get("hbm_write_MB_per_pass", 0.0) for r in recs) * 1e6),
            "launches_per_pass": {b: r.get("launches_per_pass") for b, r in zip(bases, recs)},
            # dynamic instruction counters, summed over every launch of a pass (all waves of the chip)
            "mfma_busy_cycles": per_pass("SQ_VALU_MFMA_BUSY_CYCLES"), "insts_mfma": per_pass("SQ_INSTS_MFMA"),
            "insts_valu": per_pass("SQ_INSTS_VALU") if all("SQ_INSTS_VALU" in r for r in recs) else None,
            "mfma_busy": round(dom["MfmaUtil_pct"] / 100.0, 4) if "MfmaUtil_pct" in dom else None,
            "valu_active": round(dom["VALU_active_frac_of_wave_cycles"], 4) if "VALU_active_frac_of_wave_cycles" in dom else None,
            "wait": round(dom["wait_frac_of_wave_cycles"], 4) if "wait_frac_of_wave_cycles" in dom else None,
            "source": src}


PEAK_F16_TFLOPS = 2500.0   # dense bf16 / fp16 MFMA peak (MI355X_MICROARCH.md)
TRANS_PER_S = 256 * 4 * 8 * 2.4e9   # quarter-rate transcendentals: 8 lanes per clock and SIMD, 1024 SIMDs, 2.4 GHz


def issued_pipe_peak(name, backward):
    """fp32-equivalent peak (TFLOP/s) of the matrix pipe the kernel actually issues on, read off its name tags: a 2-way split-fp16
    product is 3 f16 MFMAs (2500 / 3), a 3-way split-bf16 product 6 bf16 MFMAs (2500 / 6), fp32-input MFMA runs at 157.3.  A backward
    pass is 1/3 forward-side flops (stage recompute) and 2/3 cotangent-side (VJP wrt z and wrt theta): harmonic mix of the two."""
    f16, bf16, f32 = PEAK_F16_TFLOPS / 3, PEAK_F16_TFLOPS / 6, PEAK_FP32_TFLOPS
    if "fwd-side fp16x2 + bf16x3" in name:
        return 1.0 / ((1 / 3) / f16 + (2 / 3) / bf16), "1/3 split-fp16 (3 f16 MFMAs per product) + 2/3 split-bf16 (6 per product)"
    if "fwd-side fp16x2 + fp32" in name:
        return 1.0 / ((1 / 3) / f16 + (2 / 3) / f32), "1/3 split-fp16 (3 f16 MFMAs per product) + 2/3 fp32-input MFMA"
    if "fp16x2" in name:
        return f16, "split-fp16: 3 f16 MFMAs per fp32 product"
    if "bf16" in name or "bf3" in name:
        if backward:      # batch-tiled backward: P on split-bf16, the transposed products and hidden layers on fp32-input MFMA
            return 1.0 / ((1 / 2) / bf16 + (1 / 2) / f32), "about half split-bf16 (6 bf16 MFMAs per product), half fp32-input MFMA"
        return bf16, "split-bf16: 6 bf16 MFMAs per fp32 product"
    return f32, "fp32-input MFMA"


def dtype_note(names):
    """What the arithmetic runs on, read off the dispatched kernels' names (their `fp16x2` / `bf16x3` / `bf3` tags)."""
    parts = []
    if any("fp16x2" in n for n in names):
        parts.append("forward GEMMs as 2-way split-fp16 MFMA (operands to 2^-24, products to 2^-22; sample tiles that leave the fp16 range "
                     "are re-executed in split-bf16)")
    if any("bf16" in n or "bf3" in n for n in names):
        parts.append("the dependency-chain GEMMs of the adjoint (recompute, VJP) as exact 3-way split-bf16 MFMA")
    if parts:
        return ("fp32 in/out and fp32 accumulation; " + "; ".join(parts) + " -- fp32-equivalent (z error 5e-7, gradients 1e-6 vs the "
                "reference); the remaining GEMMs as fp32-input MFMA")
    return "fp32 throughout (fp32-input MFMA)"


N_SIMD = 256 * 4
CLOCK_HZ = 2.4e9      # (device_clock_hz() replaces it once a device is known: ADVICE round 5)


def device_clock_hz():
    """Peak engine clock of the current device as the runtime reports it (hipDeviceAttributeClockRate, kHz); 2.4 GHz if it does not."""
    try:
        khz = torch.cuda.get_device_properties(torch.cuda.current_device()).clock_rate
        return float(khz) * 1e3 if khz and khz > 1e5 else CLOCK_HZ
    except Exception:
        return CLOCK_HZ


def rocprof_avg_ms(config, which):
    """Average duration of the pass's dominant kernel in the committed `rocprofv3 --kernel-trace --stats` summary of this bench command
    (profiles/r06_bench_<config>_kernel_stats.csv + its .meta.json with the source fingerprint), so the line can be checked against
    the profile without opening the CSV; None if none of these sources is committed."""
    import csv
    for rnd in ("r06",):
        base = os.path.join(ROOT, "profiles", "%s_bench_%s_kernel_stats" % (rnd, config))
        try:
            with open(base + ".meta.json") as fh:
                if json.load(fh).get("source_fingerprint") != _lib.source_fingerprint():
                    continue
            rows = list(csv.DictReader(open(base + ".csv")))
        except (OSError, ValueError):
            continue
        key = {"forward": ("ncde_fwd_",), "backward": ("ncde_adj_", "ncde_dwo_")}[which]
        hit = [r for r in rows if any(k in r.get("Name", "") for k in key) and "only_faulted" not in r.get("Name", "")]
        hit = [r for r in hit if float(r.get("AverageNs", 0) or 0) > 2e4]      # (the re-execution launches return at once: not the pass)
        if hit:
            return {r["Name"][:120]: round(float(r["AverageNs"]) * 1e-6, 4) for r in sorted(hit, key=lambda r: -float(r["TotalDurationNs"]))[:2]}
    return None
_CENSUS_EXPECT = {"cfg2.forward": "ncde_fwd_fast_bf3<H32,HH32,C20,NW4,fp16x2>", "cfg2.backward": "ncde_adj_fast3<",
                  "cfg4.forward": "ncde_fwd_fast_bf3<H64,HH64,C4,NW4,fp16x2>", "cfg4.backward": "ncde_adj_h64<H64,HH64,NS1"}


def isa_census(config, which, name, B_local):
    """Static ISA census of the dispatched kernel (tools/isa_census.py: instruction counts by issue class inside the step loop of the
    built code object -> `valu_ceiling_ms`, `mfma_ceiling_ms`, `issue_ceiling_ms`), run LIVE on the objects this library was linked
    from; None for kernels it has no loop model for (the batch-tiled family: data-dependent loop nests) or another batch."""
    key = "%s.%s" % (config, which)
    if key not in _CENSUS_EXPECT or not name.startswith(_CENSUS_EXPECT[key]) or B_local != CONFIGS[config]["B"]:
        return None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import isa_census as ic
        return ic.census_for([key])[key]
    except Exception as e:      # no llvm-objdump / objects on this box: the committed census, if it is of these sources
        try:
            with open(os.path.join(ROOT, "profiles", "r06_isa_census.json")) as fh:
                j = json.load(fh)
            if j.get("_meta", {}).get("source_fingerprint") == _lib.source_fingerprint():
                return j.get(key)
        except (OSError, ValueError):
            pass
        sys.stderr.write("isa census unavailable for %s: %r\n" % (key, e))
        return None


def rooflines(model, c, coeffs, config, B_local, T):
    """(forward, backward) roofline records from HIP-event kernel times measured here + the committed PMC traffic."""
    ms_fwd, ms_adj, names = time_kernels(model, c, coeffs)
    steps_per_launch = B_local * (T - 1)
    f_fwd = flops_forward_per_sample_step(c)
    f_adj = 3 * f_fwd     # stage recompute + VJP wrt z + VJP wrt theta
    by_fwd = bytes_forward_per_sample_step(c)

    tanh_per_step = stages_of(c["solver"]) * c["H"] * c["C"] * 2      # exp + rcp per tanh; the backward recomputes every tanh once

    clock = device_clock_hz()

    def roof(ms, flops, nbytes, name, backward):
        tf_s = flops * steps_per_launch / (ms * 1e-3) / 1e12
        pm = pmc_pass(name, config, B_local)
        cen = isa_census(config, "backward" if backward else "forward", name, B_local)
        pipe_peak, pipe_note = issued_pipe_peak(name, backward)
        alg = nbytes * steps_per_launch
        trans_ms = tanh_per_step * steps_per_launch / TRANS_PER_S * 1e3
        # ---- ceilings: the least time each unit needs for THIS kernel's instruction stream / bytes (floors on ms_per_launch) -------
        # matrix pipe: busy cycles the SQ counted (dynamic: skipped branches are not in it) / 1024 SIMDs, else the static census
        # HBM: the bytes the TCC counters saw for this pass (the kernel's OWN traffic, as the other units are priced on its own
        # instruction stream), else the algorithmic bytes; the algorithmic floor is always listed beside it
        ceil = {"hbm_algorithmic": alg / (PEAK_HBM_GBS * 1e9) * 1e3}
        src = {"hbm_algorithmic": "algorithmic bytes / 8 TB/s"}
        if pm is not None and pm.get("traffic"):
            ceil["hbm"] = pm["traffic"] / (PEAK_HBM_GBS * 1e9) * 1e3
            src["hbm"] = "FETCH_SIZE + WRITE_SIZE of the pass (gfx950 corrections) / 8 TB/s (%s)" % pm["source"]
        else:
            ceil["hbm"] = ceil["hbm_algorithmic"]
            src["hbm"] = src["hbm_algorithmic"]
        if pm is not None and pm.get("mfma_busy_cycles"):
            ceil["mfma"] = pm["mfma_busy_cycles"] / N_SIMD / clock * 1e3
            src["mfma"] = "SQ_VALU_MFMA_BUSY_CYCLES per pass / 1024 SIMDs / %.2f GHz (%s)" % (clock / 1e9, pm["source"])
        elif cen is not None:
            ceil["mfma"] = cen["mfma_ceiling_ms"]
            src["mfma"] = "static ISA census (tools/isa_census.py): MFMAs in the step loop x their pipe cycles"
        if cen is not None:
            ceil["valu"] = cen["valu_ceiling_ms"]
            ceil["issue"] = cen["issue_ceiling_ms"]
            src["valu"] = "static ISA census: VALU 2 / packed-fp32 4 / transcendental 8 cycles per wave64 instruction, per SIMD"
            src["issue"] = "static ISA census: every instruction of the longest wave x its 4-cycle issue slot"
        elif pm is not None and pm.get("insts_valu"):
            # dynamic: VALU instructions (all waves, MFMAs excluded) at 2 cycles + the quarter-rate surcharge of the tanh work
            ceil["valu"] = (pm["insts_valu"] - (pm.get("insts_mfma") or 0.0)) * 2.0 / N_SIMD / clock * 1e3 + 0.75 * trans_ms
            src["valu"] = "SQ_INSTS_VALU - SQ_INSTS_MFMA per pass x 2 cycles / 1024 SIMDs + the transcendentals' quarter-rate surcharge"
        # ---- serial-issue model (round 6; measured: tools/mfma_valu_overlap.hip, profiles/r06_mfma_valu_overlap.txt): the matrix pipe
        # and the VALU of a SIMD do not work side by side -- next to a saturated MFMA stream the SIMD's other wave gets one VALU
        # instruction per 12.8 cycles, mixed streams of two waves finish one after the other --, so a mixed stream needs about the SUM
        # over the SIMD's waves of VALU 4.1 / packed 6.5 / transcendental 10 / MFMA 17.1 (16x16x32), 32.1 (32x32x16, fp32 16x16x4) cycles
        serial_ms, serial_src = None, None
        if cen is not None and cen.get("serial_issue_ceiling_ms"):
            serial_ms = cen["serial_issue_ceiling_ms"]
            serial_src = "static ISA census priced at the measured serial issue costs (tools/isa_census.py, profiles/r06_mfma_valu_overlap.txt)"
        elif pm is not None and pm.get("insts_valu") and pm.get("mfma_busy_cycles"):
            n_trans = tanh_per_step * steps_per_launch / 64.0      # wave instructions
            serial_ms = ((pm["insts_valu"] - (pm.get("insts_mfma") or 0.0)) * 4.1 + n_trans * (10.0 - 4.1) + pm["mfma_busy_cycles"] * (17.1 / 16.0)) / N_SIMD / clock * 1e3
            serial_src = "SQ_INSTS_VALU - SQ_INSTS_MFMA at 4.1 cycles + the transcendentals' surcharge + matrix-pipe busy cycles x 17.1 / 16, per SIMD (%s)" % pm["source"]
        if serial_ms is not None:
            ceil["serial_issue"] = serial_ms
            src["serial_issue"] = serial_src
        # ---- which unit the counters show busiest; `wait` above it = the waves stall more than any unit works ----------------------
        bound, stalled, busy = "unmeasured", None, {}
        if pm is not None and pm["mfma_busy"] is not None and pm["valu_active"] is not None:
            busy = {"mfma": pm["mfma_busy"], "valu": pm["valu_active"], "hbm": pm["traffic"] / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS}
            bound = max(busy, key=busy.get)
            stalled = pm["wait"] is not None and pm["wait"] > busy[bound]
        elif "valu" in ceil and "mfma" in ceil:
            bound = max(("mfma", "valu", "hbm"), key=lambda k: ceil[k])
        # PRIMARY: algorithmic fp32 flops per launch / HIP-event time against the fp32
        # MFMA / vector peak -- `frac` = `frac_fp32_peak` = achieved / 157.3 TFLOP/s, not clamped (above 1 only where the products run as
        # split-fp16 on the 2.5 PF pipe: `frac_of_issued_pipe` prices those).  The busiest-unit ceiling of the kernel's own instruction
        # stream is reported beside it, RAW: a model that exceeds the measured time is flagged, not hidden.
        frac = tf_s / PEAK_FP32_TFLOPS
        unit_frac = ceil[bound] / ms if bound in ceil else None
        r = {"bound": bound, "kernel": name, "achieved": round(tf_s, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
             "frac": round(frac, 4), "frac_fp32_peak": round(frac, 4),
             "frac_of_busiest_unit_ceiling": round(unit_frac, 4) if unit_frac is not None else None,
             "ceiling_model_exceeds_measured": bool(unit_frac is not None and unit_frac > 1.0),
             "rocprof_avg_ms": rocprof_avg_ms(config, "backward" if backward else "forward"),
             "traffic": pm["traffic"] if pm else None,
             "traffic_ratio": round(pm["traffic"] / alg, 3) if pm else None,
             "algorithmic_bytes_per_launch": alg, "ms_per_launch": round(ms, 4),
             "ceilings_ms": {k: round(v, 4) for k, v in ceil.items()},
             "valu_ceiling_ms": round(ceil["valu"], 4) if "valu" in ceil else None,
             "mfma_ceiling_ms": round(ceil["mfma"], 4) if "mfma" in ceil else None,
             "issue_ceiling_ms": round(ceil["issue"], 4) if "issue" in ceil else None,
             "tanh_ceiling_ms": round(trans_ms, 4),
             "frac_of_issue_ceiling": round(ceil["issue"] / ms, 4) if "issue" in ceil else None,
             "serial_issue_model_ms": round(serial_ms, 4) if serial_ms is not None else None,
             "frac_of_serial_issue_model": round(serial_ms / ms, 4) if serial_ms is not None else None,
             "ceiling_sources": src,
             "stalled": stalled,
             "busy": {k: round(v, 4) for k, v in busy.items()} if busy else None,
             "mfma_busy": pm["mfma_busy"] if pm else None, "valu_active": pm["valu_active"] if pm else None,
             "wait": pm["wait"] if pm else None,
             "hbm_algorithmic_GBs": round(alg / (ms * 1e-3) / 1e9, 2),
             "hbm_frac": round(alg / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 6),
             "x_fp32_peak": round(tf_s / PEAK_FP32_TFLOPS, 4), "fp32_peak": PEAK_FP32_TFLOPS,
             "issued_pipe": pipe_note, "issued_pipe_peak": round(pipe_peak, 1), "frac_of_issued_pipe": round(tf_s / pipe_peak, 4),
             "clock_hz": clock,
             "bound_note": "`achieved` = SURVEY.md section 8d algorithmic fp32 flops per sample-step x B (T-1) sample-steps / ms_per_launch (HIP "
                           "events); `peak` = 157.3 TFLOP/s (fp32 MFMA = fp32 vector peak); `frac` = `frac_fp32_peak` = achieved / peak; "
                           "`bound` = the unit the SQ / TCC counters show busiest (mfma: matrix-pipe busy cycles, valu: VALU + transcendental "
                           "issue, hbm: counter bytes against 8 TB/s); `frac_of_busiest_unit_ceiling` = that unit's ceiling for this kernel's "
                           "own instruction stream / bytes (`ceilings_ms`) over ms_per_launch, unclamped; `stalled` = the waves wait "
                           "(SQ_WAIT_INST_ANY) more than the busiest unit works: latency-, not throughput-bound; `frac_of_issued_pipe` = the "
                           "same flops against the pipe the products are issued on (2.5 PF / 3 for split-fp16, / 6 for split-bf16); "
                           "`rocprof_avg_ms` = the committed rocprofv3 average of the pass's kernels (same sources)"}
        if pm is not None:
            r["traffic_unit"] = "HBM bytes per pass, summed over all launches of all kernels of the pass (rocprofv3 PMC passes on these kernel sources, %s)" % pm["source"]
            r["launches_per_pass"] = pm["launches_per_pass"]
        return r

    return (roof(ms_fwd, f_fwd, by_fwd, names[0], False), roof(ms_adj, f_adj, by_fwd, names[1], True)), names


def other_config(name, dev, steps=None, warmup=None):
    """One BASELINE shape besides the headline, single GPU: `steps` training steps (forward + adjoint + Adam) and the kernel rooflines.
    (cfg5's step takes 0.85 s: three of them; the millisecond configs get ten, so that one hiccup does not double their mean)"""
    steps = steps if steps is not None else (3 if name == "cfg5" else 10)
    warmup = warmup if warmup is not None else (1 if name == "cfg5" else 2)
    c = dict(CONFIGS[name])
    w = Workload(c, c["B"], c["B"], 0, dev)
    dt, loss = w.timed(steps, warmup, 1, dev)
    roofs, names = rooflines(w.model, c, w.coeffs, name, c["B"], w.T)
    rec = {"workload": "BASELINE %s: %s interpolation, %s step 1, B=%d, raw L=%d -> T=%d knots, C=%d, H=HH=%d, nl=%d; step = forward + "
                       "adjoint backward + Adam" % (name, c["interpolation"], c["solver"], c["B"], c["L"], w.T, c["C"], c["H"], c["nl"]),
           "value": c["B"] * (w.T - 1) * steps / dt, "unit": "sample-steps/s", "steps": steps, "warmup": warmup,
           "ms_per_step": dt / steps * 1e3, "dtype": "f32", "dtype_note": dtype_note(names), "loss": loss,
           "roofline": roofs[1], "roofline_forward": roofs[0]}
    del w
    torch.cuda.empty_cache()
    return rec

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="which workload the headline `value` is quoted on; with N>1 the other one is timed too")
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU (weak) / global (strong) batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the forward-only / adjoint=False / other-scaling legs")
    ap.add_argument("--cpu-sample", type=int, default=1024)
    args = ap.parse_args()

    rank, local_rank, world = D.env_world()
    # (a bare `python bench.py --gpus N` never gets here with N > 1: _self_launch() has started the N ranks as a child process)
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d: launch with torch.distributed.run --nproc-per-node %d" % (world, args.gpus, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs a GPU: there is no CPU fallback for the product path"
    # (NCDE_BENCH_BACKEND=gloo: the test-suite's way to run the world > 1 branches on a ONE-GPU box, both ranks on cuda:0 -- RCCL
    # refuses two ranks on one device; the driver's runs use the default, nccl = RCCL, one rank per GPU)
    backend = os.environ.get("NCDE_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    D.init_process_group(backend)
    c = dict(CONFIGS[args.config])

    def shard(scaling):
        if scaling == "weak":
            B_local = args.batch or c["B"]
            return B_local, B_local * world, rank * B_local
        B_total = args.batch or c["B"]
        lo, hi = D.shard_bounds(B_total, rank, world)
        # equal shards: the averaged all-reduce of per-shard mean-loss gradients is then the global-mean gradient
        assert B_total % world == 0, "strong scaling needs a global batch divisible by the number of ranks"
        return hi - lo, B_total, lo

    # the collective the training step uses, exercised once on a tiny tensor before anything is timed: `rccl_ranks` in the record is
    # what the backend itself reports after a real all-reduce (N ranks each contributing 1)
    rccl_ranks, ms_allreduce = 1, 0.0
    if world > 1:
        probe = torch.ones(1, device=dev)
        torch.distributed.all_reduce(probe)
        rccl_ranks = int(round(float(probe.item())))
        assert rccl_ranks == torch.distributed.get_world_size() == world

    B_local, B_total, lo = shard(args.scaling)
    w = Workload(c, B_local, B_total, lo, dev)
    model, coeffs, T = w.model, w.coeffs, w.T
    z_err = golden_z_err(model, c, coeffs) if (rank == 0 and args.config == "cfg2" and lo == 0) else None
    dt, loss = w.timed(args.steps, args.warmup, world, dev)

    if world > 1:      # the step's one exchange in isolation: all-reduce of the flat fp32 gradient bucket, per rank, HIP events
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            torch.distributed.all_reduce(w.bucket.flat)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            torch.distributed.all_reduce(w.bucket.flat)
        e1.record()
        torch.cuda.synchronize()
        tar = torch.tensor([e0.elapsed_time(e1) / 20], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tar, op=torch.distributed.ReduceOp.MAX)
        ms_allreduce = float(tar.item())
        w.bucket.flat.zero_()

    tf = td = t32 = None
    other = None
    if not args.no_extras:
        # forward-only throughput (inference), same inputs
        with torch.no_grad():
            model(coeffs)
            torch.cuda.synchronize()
            tf0 = time.perf_counter()
            for _ in range(args.steps):
                model(coeffs)
            torch.cuda.synchronize()
            tf = (time.perf_counter() - tf0) / args.steps
        # adjoint=False training step (recording forward + exact discrete backward), reported beside the headline
        model.adjoint = False
        w.step()
        torch.cuda.synchronize()
        td0 = time.perf_counter()
        for _ in range(args.steps):
            w.step()
        torch.cuda.synchronize()
        td = (time.perf_counter() - td0) / args.steps
        model.adjoint = True
        # the same training step with every GEMM as plain fp32-input MFMA (NCDE_FLAG_FP32_MFMA): the headline beside it does not rest
        # on the split-fp16 / split-bf16 argument
        model.kernel_flags = _lib.FLAG_FP32_MFMA
        w.step()
        torch.cuda.synchronize()
        t320 = time.perf_counter()
        for _ in range(args.steps):
            w.step()
        torch.cuda.synchronize()
        t32 = (time.perf_counter() - t320) / args.steps
        model.kernel_flags = 0
        if world > 1:      # the other scaling mode, same contract (barrier + synchronize, max over ranks)
            oname = "strong" if args.scaling == "weak" else "weak"
            oB_local, oB_total, olo = shard(oname)
            ow = Workload(c, oB_local, oB_total, olo, dev)
            odt, _ = ow.timed(args.steps, args.warmup, world, dev)
            other = {"scaling": oname, "value": oB_total * (ow.T - 1) * args.steps / odt, "unit": "sample-steps/s",
                     "ms_per_step": odt / args.steps * 1e3, "global_batch": oB_total, "batch_per_gpu": oB_local}
            del ow

    if rank == 0:
        roofs, names = rooflines(model, c, coeffs, args.config, B_local, T)
        rec = {
            "metric": "solved integration steps/sec (fwd+adjoint)",
            "value": B_total * (T - 1) * args.steps / dt,
            "unit": "sample-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "dtype_note": dtype_note(names),
            "config": {"workload": "BASELINE %s: %s interpolation, %s step 1, B=%d per GPU (global %d), raw L=%d -> T=%d knots, "
                                   "C=%d, H=HH=%d, nl=%d; step = forward + adjoint backward + %sAdam"
                                   % (args.config, c["interpolation"], c["solver"], B_local, B_total, c["L"], T, c["C"], c["H"],
                                      c["nl"], "RCCL grad all-reduce + " if world > 1 else ""),
                       "global_batch": B_total, "seq_len": c["L"], "parallelism": "dp%d" % world},
            "roofline": roofs[1],
            "roofline_forward": roofs[0],
            "loss": loss,
            "rccl_ranks": rccl_ranks, "collective_backend": backend if world > 1 else None,
            "ms_allreduce": ms_allreduce,
            "allreduce_floats": int(w.bucket.flat.numel()),
        }
        # where the step's time goes outside the two solver kernels (cdeint's per-call host work, the
        # range-fault re-execution launches, the gradient reduction, read-in / read-out layers, loss, Adam -- and launch gaps)
        kf, kb = roofs[0].get("ms_per_launch"), roofs[1].get("ms_per_launch")
        if kf is not None and kb is not None:
            rec["step_breakdown"] = {"forward_kernel_ms": kf, "backward_kernel_ms": kb, "other_ms": round(dt / args.steps * 1e3 - kf - kb, 4),
                                     "note": "other = ms_per_step - the two solver kernels' own time (HIP events): everything cdeint, autograd, "
                                             "the model's linear layers, the loss, the optimizer and launch gaps add per step"}
        if z_err is not None:
            rec["z_err_vs_golden"] = z_err
            rec["z_err_note"] = ("max-abs error of z_T from the timed default forward kernel (%s) on this full workload, relative to max |z_T|, "
                                 "against the reference's own z_T (tests/golden/g5_cfg2_full.npz); north-star tolerance 1e-4" % names[0])
        if t32 is not None:
            rec["fp32_mfma_ms_per_step"] = t32 * 1e3
            rec["fp32_mfma_value"] = B_local * (T - 1) / t32
            rec["fp32_mfma_note"] = "same step with NCDE_FLAG_FP32_MFMA: every GEMM as plain fp32-input MFMA (per rank, no barrier)"
        if tf is not None:
            rec.update({"forward_only_value": B_local * (T - 1) / tf, "forward_only_ms": tf * 1e3,
                        "adjoint_false_value": B_local * (T - 1) / td, "adjoint_false_ms_per_step": td * 1e3,
                        "adjoint_false_note": "same step with NeuralCDE(adjoint=False): recording forward + exact discrete backward (per rank, no barrier)"})
        if world == 1:     # one GPU: the weak and the strong workload coincide (global batch = per-GPU batch)
            rec["strong" if args.scaling == "weak" else "weak"] = {"value": rec["value"], "ms_per_step": rec["ms_per_step"], "global_batch": B_total, "batch_per_gpu": B_local, "note": "identical to the headline at n_gpus=1"}
        elif other is not None:
            rec[other["scaling"]] = other
        if not args.no_cpu_baseline and world == 1:
            # a BOUNDED sample: the same CPU work as 1024 samples of cfg2 (10 - 30 s with the repetitions) whatever the config -- cfg5 is
            # 113 x the flops per sample (798 steps of a 128-wide field over 80 channels): 16 samples there
            per_sample = lambda cc: flops_forward_per_sample_step(cc) * ((cc["L"] * 2 - 1 if cc["interpolation"] == "rectilinear" else cc["L"]) - 1)
            cpu_B = int(min(args.cpu_sample, max(16, 16 * round(args.cpu_sample * per_sample(CONFIGS["cfg2"]) / per_sample(c) / 16))))
            rec["cpu_baseline"] = cpu_baseline(c, w.fw, w.rw, cpu_B)
            rec["gpu_over_cpu"] = rec["value"] / rec["cpu_baseline"]["value"]
        if world == 1 and args.config == "cfg2" and not args.no_extras and not args.batch:
            # the other BASELINE shapes, OUTSIDE the headline's timed region: a few steps each so that the driver's record carries
            # all three (kernel names, step time, roofline with PMC traffic where a summary of these sources is committed)
            del w, model, coeffs
            torch.cuda.empty_cache()
            rec["other_configs"] = {name: other_config(name, dev) for name in ("cfg4", "cfg5")}
        print(json.dumps(rec))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
