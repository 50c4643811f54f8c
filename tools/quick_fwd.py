"""Quick check on the GPU box: cfg2 forward kernel time (plain + instrumented phases) and parity vs the golden z_T."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import ncde_amd, bench
from ncde_amd import _lib, solver
c = dict(bench.CONFIGS["cfg2"])
B = 4096
coeffs = bench.make_inputs(c, B, 0, torch.device("cuda", 0))
model, fw, rw = bench.make_model(c, "cuda")
spec = model.func.fused_spec()
with torch.no_grad():
    z0 = model.initial_linear(coeffs[:, 0]).contiguous()
lib = _lib.lib()
for flags, label in ((0, "split-bf16 plain"), (0x100, "split-bf16 instrumented")):
    p = solver.build_problem(coeffs, "linear", z0, spec, "rk4", _lib.OUT_INTERVAL, flags)
    ws = torch.zeros(int(lib.ncde_workspace_bytes(ctypes.byref(p), 0)), dtype=torch.uint8, device="cuda")
    out = torch.empty(B, 2, 32, device="cuda")
    ms = ctypes.c_float()
    _lib.check(lib.ncde_time_kernel(ctypes.byref(p), 0, out.data_ptr(), None, None, ws.data_ptr(), ws.numel(), None, 5, ctypes.byref(ms)), "time")
    print(label, "ms/launch %.4f" % ms.value)
    if flags:
        cyc = ws[: (B // 16) * 4 * 4 * 8].view(torch.int64).view(-1, 4, 4).cpu().numpy().astype(np.float64)
        per = cyc.mean(axis=0) / (398 * 4)
        print("cycles/stage by wave x phase (hidden | out+tanh | rk+exchange):"); print(np.array2string(per[:, :3], precision=0))
    else:
        f = np.load(os.path.join(ROOT, "tests", "golden", "g5_cfg2_full.npz"))
        zT = out[:, -1].cpu().numpy()
        print("z_T vs reference golden: %.2e" % (np.abs(zT - f["zT"]).max() / np.abs(f["zT"]).max()))
