import os, time, sys, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(p, open(p).read().strip())
    except Exception as e: print(p, "n/a")
os.system("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Socket' ; free -g | head -2")
sys.path[:0] = [os.getcwd(), "oracle"]
import ncde_amd, ncde_oracle as orc
c = ncde_amd.data.make_rectilinear_coeffs(256, 25, 19, 0.3, 1234)
fw = ncde_amd.data.make_field_weights(32, 32, 20, seed=0)
f = orc.Field.original(fw, 32, 20, 3)
z0 = torch.zeros(256, 32)
for nt in (1, 4, 8, 16, 32, 64):
    torch.set_num_threads(nt)
    t = time.time(); orc.solve_forward(orc.Control(c, "linear"), f, z0, "rk4", False); print("threads", nt, "fwd B=256,T=49: %.3fs" % (time.time() - t))
