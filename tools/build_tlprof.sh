#!/bin/bash
# development: libncde_hip.so with the s_memtime-instrumented batch-tiled sweep (-DNCDE_TL_PROF) -> variants/tlprof.so (tools/prof_cfg5.py)
set -e
ROOT=/root/repo; C=$ROOT/online-neural-cdes_amd/csrc; mkdir -p $ROOT/variants
hipcc -DNCDE_TL_PROF $EXTRA --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$ROOT/include -c $C/ncde_tiled.hip -o /tmp/tiled_prof_${NAME:-tlprof}.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/variants/${NAME:-tlprof}.so $(ls $C/*.o | grep -v ncde_tiled.o) /tmp/tiled_prof_${NAME:-tlprof}.o
echo built variants/${NAME:-tlprof}.so
