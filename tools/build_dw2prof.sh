#!/bin/bash
# development: libncde_hip.so with the cycle-counter build of ncde_dwo_h2 (-DNCDE_DW2_PROF) -> variants/dw2prof.so (tools/prof_dw2.py)
set -e
ROOT=/root/repo; C=$ROOT/online-neural-cdes_amd/csrc; mkdir -p $ROOT/variants
hipcc -DNCDE_DW2_PROF $EXTRA --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -I$ROOT/include -c $C/ncde_dwo2.hip -o /tmp/dwo2_prof.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/variants/${NAME:-dw2prof}.so $(ls $C/*.o | grep -v ncde_dwo2.o) /tmp/dwo2_prof.o
echo built variants/${NAME:-dw2prof}.so
