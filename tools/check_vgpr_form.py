"""Bit-for-bit comparison of the shipped library with variants/novgprform.so (tools/check_vgpr_form.sh): the kernels of the two units
built with -amdgpu-mfma-vgpr-form.  Each library runs in its own child process (one ctypes handle per process)."""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import hashlib, os, sys
ROOT = sys.argv[1]; sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
from ncde_amd import _lib
if sys.argv[2] != "-": _lib.LIB_PATH = os.path.join(ROOT, sys.argv[2])
import numpy as np, torch
import golden_util as gu, gpu_util
h = hashlib.sha256()
import test_gpu_parity as T
for (C, H, HH) in ((20, 32, 32), (4, 64, 64)):
    for interp in ("linear", "cubic"):
        for method in ("rk4", "midpoint", "euler"):
            for flags in (0, _lib.FLAG_SPLIT_BF16):
                case = T._seeded_case(interp, method, True, B=100, L=9, C=C, H=H, HH=HH, nl=3, seed=77)
                h.update(gpu_util.run_case(case, flags=flags, need_grads=False)["z_out"].tobytes())
case = T._seeded_case("linear", "rk4", False, B=512, L=4, C=80, H=128, HH=128, nl=3, seed=78)
r = gpu_util.run_adjoint_direct(case, case["expect"]["z_out"])
assert "ncde_dwo_h2" in r["kernel"], r["kernel"]
for k in sorted(r["grads"]): h.update(r["grads"][k].tobytes())
print("HASH", h.hexdigest())
'''
out = []
for lib in ("-", "variants/novgprform.so"):
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, lib], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out.append([ln for ln in r.stdout.splitlines() if ln.startswith("HASH")][0])
    print(lib, out[-1])
assert out[0] == out[1], "the -amdgpu-mfma-vgpr-form units differ from their plain builds"
print("bit-identical")
