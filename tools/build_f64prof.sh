#!/bin/bash
# development: libncde_hip.so with the s_memtime-instrumented ncde_adj_h64 (-DNCDE_F64_PROF) -> variants/f64prof.so (read by tools/prof_cfg4.py)
set -e
ROOT=/root/repo; C=$ROOT/online-neural-cdes_amd/csrc; mkdir -p $ROOT/variants
hipcc -DNCDE_F64_PROF $EXTRA --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$ROOT/include -c $C/ncde_fast64.hip -o /tmp/fast64_prof.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/variants/${NAME:-f64prof}.so $(ls $C/*.o | grep -v ncde_fast64.o) /tmp/fast64_prof.o
echo built variants/${NAME:-f64prof}.so
