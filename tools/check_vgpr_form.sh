#!/bin/bash
# two translation units (ncde_fast_fwd3.hip, ncde_dwo2.hip) are built with the hidden LLVM option
# -amdgpu-mfma-vgpr-form.  Step 1 (here, CPU): build variants/novgprform.so = the same library with those units built WITHOUT it.
# Step 2 (GPU box):  python tools/check_vgpr_form.py  -- every forward instantiation of the two register-resident shape sets (split-fp16,
# split-bf16 re-execution instance) and the cooperative backward of a cfg5-shaped problem, both libraries, results compared BIT FOR BIT.
set -e
ROOT=/root/repo; C=$ROOT/online-neural-cdes_amd/csrc; mkdir -p $ROOT/variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$ROOT/include"
hipcc $F -c $C/ncde_fast_fwd3.hip -o /tmp/fwd3_plain.o
hipcc $F -c $C/ncde_dwo2.hip -o /tmp/dwo2_plain.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/variants/novgprform.so $(ls $C/*.o | grep -v -e ncde_fast_fwd3.o -e ncde_dwo2.o) /tmp/fwd3_plain.o /tmp/dwo2_plain.o
echo built variants/novgprform.so
