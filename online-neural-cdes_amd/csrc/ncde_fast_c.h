// (H, HH) = (32, 32) with C = 4 / 8 / 12 on the register-resident kernel templates of ncde_fast.hip: one translation unit per channel
// count (ncde_fast_c.hip compiled with -DNCDE_FAST_C=...), so that the instantiations build in parallel with the rest.
#pragma once
#include "ncde_common.h"

typedef void (*NcdeFastCKernel)(KArgs);
// forward: runtime layer count (1 .. 4), hp 1 = split-fp16 (default), 0 = split-bf16 (also the re-execution instance)
// adjoint / exact discrete backward: n_layers in 1 .. 4, hp 2 (default) / 0; nullptr where the LDS plan does not fit
#define NCDE_FAST_C_DECL(C)                                                                                   \
    NcdeFastCKernel ncde_fast_c##C##_fwd(int interp, int method, int hp);                                     \
    NcdeFastCKernel ncde_fast_c##C##_adj(int n_layers, int interp, int method, int hp, bool discrete);        \
    size_t ncde_fast_c##C##_adj_lds(int n_layers, int interp, int hp);
NCDE_FAST_C_DECL(4)
NCDE_FAST_C_DECL(8)
NCDE_FAST_C_DECL(12)
