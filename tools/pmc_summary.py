"""Summarise rocprofv3 --pmc passes (counter_collection.csv files under a directory) per kernel:
    python tools/pmc_summary.py <dir> <out.json> [--config cfg2 --batch 4096 --passes 8 --note "..."]
--passes = forward (= backward) passes the profiled command executed (bench.py --steps K --warmup W --no-extras: K + W training steps
+ the 1 + 3 launches of its HIP-event timing leg): a pass of the windowed batch-tiled backward is MANY launches of two kernels, and
`*_per_pass` = (sum over all launches seen) / passes is what bench.py reports as `roofline.traffic`.
FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B per the counter definition; FETCH_SIZE of wide coalesced reads
is doubled (MI355X_MICROARCH.md, HBM section).  GRBM_GUI_ACTIVE in the CSV is the sum over the 8 XCDs.
The summary records the fingerprint of the kernel sources it was taken on (`_meta.source_fingerprint`): bench.py reports
`roofline.traffic` from it only while the library it times is built from exactly those sources."""
import argparse, csv, glob, json, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("root"); ap.add_argument("out")
ap.add_argument("--config", default="cfg2"); ap.add_argument("--batch", type=int, default=4096); ap.add_argument("--note", default="")
ap.add_argument("--passes", type=int, default=0)
a = ap.parse_args()
# Counters are collected per full kernel name (template arguments included): the split-fp16 kernels are followed by a launch of
# the split-bf16 instantiation of the SAME kernel template that re-executes range-faulted tiles (normally none -- every workgroup
# exits at once).  Per base name the instantiation that does the work (largest average counter values) is reported under the base
# name, any other one under "<base name>|<template arguments>".
full = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(a.root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "ncde_" not in name:
            continue
        body = name.split("ncde_")[1].split("(")[0]
        full["ncde_" + body][r["Counter_Name"]].append(float(r["Counter_Value"]))
by_base = defaultdict(list)
for k in full:
    by_base[k.split("<")[0]].append(k)
acc = {}
for base, ks in by_base.items():
    weight = lambda k: sum(sum(v) / len(v) for v in full[k].values())      # noqa: E731
    ks.sort(key=weight, reverse=True)
    acc[base] = full[ks[0]]
    for k in ks[1:]:
        acc[base + "|" + k[len(base):]] = full[k]
res = {}
for k, cs in acc.items():
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    d["launches_seen"] = max(len(v) for v in cs.values())
    if "FETCH_SIZE" in d:
        d["hbm_read_MB_per_launch_corrected_x2"] = d["FETCH_SIZE"] * 1024 * 2 / 1e6
    if "WRITE_SIZE" in d:
        d["hbm_write_MB_per_launch"] = d["WRITE_SIZE"] * 1024 / 1e6
    if a.passes > 0:
        d["launches_per_pass"] = d["launches_seen"] / a.passes
        if "FETCH_SIZE" in cs:
            d["hbm_read_MB_per_pass_corrected_x2"] = sum(cs["FETCH_SIZE"]) * 1024 * 2 / 1e6 / a.passes
        if "WRITE_SIZE" in cs:
            d["hbm_write_MB_per_pass"] = sum(cs["WRITE_SIZE"]) * 1024 / 1e6 / a.passes
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        d["MfmaUtil_pct"] = 100.0 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8 * 256 * 4)
    if "SQ_ACTIVE_INST_VALU" in d and "SQ_WAVE_CYCLES" in d:
        d["VALU_active_frac_of_wave_cycles"] = d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"]
    if "SQ_WAIT_INST_ANY" in d and "SQ_WAVE_CYCLES" in d:
        d["wait_frac_of_wave_cycles"] = d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"]
    res[k] = d
from ncde_amd import _lib  # noqa: E402
res["_meta"] = {"source_fingerprint": _lib.source_fingerprint(), "config": a.config, "batch": a.batch, "passes": a.passes, "note": a.note}
json.dump(res, open(a.out, "w"), indent=1)
print(json.dumps(res, indent=1))
