"""development: phase-cycle breakdown of the batch-tiled FORWARD (ncde_fwd_tiled) from the instrumented build (tools/build_tlprof.sh).
usage: python tools/prof_cfg5_fwd.py [variants/tlprof.so] [cfg] [B] [L]   (run on the GPU box; the solution of such a build is garbage)"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from ncde_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "variants/tlprof.so")
import ncde_amd, bench
from ncde_amd import solver
c = dict(bench.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "cfg5"])
B = int(sys.argv[3]) if len(sys.argv) > 3 else c["B"]
c["L"] = int(sys.argv[4]) if len(sys.argv) > 4 else 60
dev = torch.device("cuda", 0)
coeffs = bench.make_inputs(c, B, 0, dev)
model, fw, rw = bench.make_model(c, "cuda")
spec = model.func.fused_spec()
interp = "cubic" if c["interpolation"] == "cubic" else "linear"
with torch.no_grad():
    z0 = model.initial_linear(coeffs[:, 0, :c["C"]]).contiguous()
lib = _lib.lib()
H = c["H"]
for flags, label in ((_lib.FLAG_FORCE_TILED, "default"), (_lib.FLAG_FORCE_TILED | _lib.FLAG_NO_COOP, "per-workgroup kernel")):
    p = solver.build_problem(coeffs, interp, z0, spec, c["solver"], _lib.OUT_INTERVAL, flags)
    out = torch.zeros(B, 2, H, device=dev)
    ws = solver._workspace(p, 0, dev)
    ms = ctypes.c_float()
    _lib.check(lib.ncde_time_kernel(ctypes.byref(p), 0, out.data_ptr(), None, None, ws.data_ptr(), ws.numel(), None, 1, ctypes.byref(ms)), "time")
    name = (lib.ncde_kernel_name(ctypes.byref(p), 0) or b"?").decode()
    per = out.view(B // 16, 16 * 2 * H)[:, :64].view(-1, 8, 8).cpu().numpy().mean(axis=0)
    T = coeffs.shape[1] + (1 if interp == "cubic" else 0)
    print("%s  %s: forward %.2f ms (%.1f us per stage); cycles per stage by wave x [dX/dt + record | hidden layers | scales + publish | wait B1 | "
          "keeper loop (per-workgroup kernel: output tiles) | arrive + wait B2 | f.dX read | bookkeeping]" % (label, name, ms.value, ms.value * 1e3 / ((T - 1) * bench.stages_of(c["solver"]))))
    print(np.array2string(per, precision=0, suppress_small=True), " total", per.sum(axis=1).round())
    if "coop" in name:
        kp = out.view(B // 16, 16 * 2 * H)[:, 64:88].view(-1, 8, 3).cpu().numpy().mean(axis=0)
        print("keeper loop per stage by wave x [compute + f.dX stores | DMA wait | barrier]:")
        print(np.array2string(kp, precision=0, suppress_small=True))
