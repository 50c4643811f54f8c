#!/bin/bash
# development: build libncde_hip.so variants of ncde_fast.hip with different -D switches into variants/<name>.so
# usage: tools/build_variants.sh name1 "-DA=1 -DB=0" name2 "..." ...
set -e
ROOT=/root/repo; C=$ROOT/online-neural-cdes_amd/csrc; mkdir -p $ROOT/variants
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  ( hipcc $flags --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$ROOT/include -c $C/ncde_fast.hip -o /tmp/fast_$name.o 2>/dev/null &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/variants/$name.so $C/ncde_abi.o $C/ncde_generic.o /tmp/fast_$name.o $C/ncde_fast4.o $C/ncde_prepare.o $C/ncde_tiled.o $C/ncde_variant.o $C/ncde_timeplan.o $C/ncde_adaptive.o && echo built $name ) &
done
wait
