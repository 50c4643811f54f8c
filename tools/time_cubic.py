"""Kernel time of the natural-cubic builder at cfg4 size under rocprofv3 (run: rocprofv3 --kernel-trace --stats ... -- python3 tools/time_cubic.py)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ncde_amd
if len(sys.argv) > 1:
    print("note: NCDE_CUBIC_LDS_KB / NCDE_CUBIC_DBG are read only by a library built with -DNCDE_DEV_KNOBS (make EXTRA=-DNCDE_DEV_KNOBS); "
          "the shipped library ignores them and every iteration below then times the same configuration", file=sys.stderr)
xc = torch.from_numpy(ncde_amd.data.synthetic_series(8192, 182, 3, missing=0.0, seed=1234)).cuda()
for kb in sys.argv[1:] or ["52"]:
    if ":" in kb:
        kb, dbg = kb.split(":"); os.environ["NCDE_CUBIC_DBG"] = dbg
    os.environ["NCDE_CUBIC_LDS_KB"] = kb
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): out = ncde_amd.natural_cubic_coeffs(xc)
    torch.cuda.synchronize(); ev0.record()
    for _ in range(20): out = ncde_amd.natural_cubic_coeffs(xc)
    ev1.record(); torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) / 20 * 1e3
    print(os.environ.get("NCDE_CUBIC_DBG"), "LDS budget %s KB: %.1f us per call (both kernels + allocs), %.0f GB/s" % (kb, us, (xc.numel() + out.numel()) * 4 / us / 1e3))
