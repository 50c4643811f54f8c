#!/bin/bash
# development: libncde_hip.so with wall-clock stamps in the fused dopri5 attempt kernels (-DNCDE_DPF_PROF) -> variants/dpfprof.so (tools/prof_dpf.py)
set -e
ROOT=/root/repo; C=$ROOT/online-neural-cdes_amd/csrc; mkdir -p $ROOT/variants
hipcc -DNCDE_DPF_PROF $EXTRA --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$ROOT/include -c $C/ncde_adaptive_fast.hip -o /tmp/dpf_prof.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/variants/${NAME:-dpfprof}.so $(ls $C/*.o | grep -v ncde_adaptive_fast.o) /tmp/dpf_prof.o
echo built variants/${NAME:-dpfprof}.so
