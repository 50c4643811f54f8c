"""One-GPU batch sweep of the cfg2 (SWEEP_CONFIG=cfg4 | cfg5: that config's) training step (forward + adjoint + Adam): B = 512 / 1024 / 2048 / 4096.
B = 4096/N on ONE GPU is the per-rank compute time of N-way STRONG scaling of the north-star workload (global batch 4096),
so the sweep bounds the strong-scaling curve without an 8-GPU node:  speedup(N) <= t(4096) / (t(4096/N) + t_allreduce).
    python tools/batch_sweep.py [out.json]
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
CFG = os.environ.get("SWEEP_CONFIG", "cfg2")
c = dict(bench.CONFIGS[CFG])
STEPS = 20 if CFG != "cfg5" else 4
rows = []
for B in (512, 1024, 2048, 4096):
    w = bench.Workload(c, B, B, 0, dev)
    dt, _ = w.timed(STEPS, 3 if CFG != "cfg5" else 1, 1, dev)
    ms_fwd, ms_adj, names = bench.time_kernels(w.model, c, w.coeffs)
    rows.append({"batch": B, "ms_per_step": dt / STEPS * 1e3, "sample_steps_per_s": B * (w.T - 1) * STEPS / dt,
                 "ms_forward_kernel": ms_fwd, "ms_adjoint_kernel": ms_adj, "kernels": names, "workgroups": (B + 15) // 16})
    print(rows[-1], flush=True)
    del w
t_full = rows[-1]["ms_per_step"]
for r in rows:
    n = 4096 // r["batch"]
    r["implied_strong_scaling_speedup_at_%d_gpus" % n] = t_full / r["ms_per_step"]
out = {"config": CFG, "rows": rows,
       "note": "implied speedup = t(B=4096) / t(B=4096/N) on one GPU, i.e. the N-GPU strong-scaling bound before the all-reduce"}
print(json.dumps(out))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
