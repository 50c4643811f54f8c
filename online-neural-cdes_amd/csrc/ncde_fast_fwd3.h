// The three-layer (unrolled) instantiations of the register-resident forward kernel ncde_fwd_fast_bf3 (ncde_fast.hip) -- the ones the
// BASELINE configs 2 - 4 run -- in their own translation unit (ncde_fast_fwd3.hip), because that unit alone is compiled with
// `-mllvm -amdgpu-mfma-vgpr-form` (MFMA results in VGPRs; csrc/Makefile): the option takes a sixth of the instructions
// out of these kernels, and it MISCOMPILES the runtime-layer-count instantiation at H = 64 in this toolchain (z wrong by 4e-2:
// test_fast_forward_with_other_layer_counts), so nothing else is built with it.
#pragma once
#include "ncde_common.h"

typedef void (*NcdeFastFwd3Kernel)(KArgs);
// hidden = 32: the (32, 32, 20) set, 64: the (64, 64, 4) set; hp = 1: split-fp16 (default), 0: split-bf16; nullptr for anything else
NcdeFastFwd3Kernel ncde_fast_fwd3(int hidden, int interp, int method, int hp);
