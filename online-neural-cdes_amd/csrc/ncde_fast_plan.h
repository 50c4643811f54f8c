// The general time axis (NcdeProblem.time_plan, output = NCDE_OUT_TIMES) on the shape-specialised kernels: PLAN = 1 instantiations of
// ncde_fwd_fast_bf3 (both shapes, runtime layer count) and of ncde_adj_fast3 (nl = 3), in their own translation unit.
#pragma once
#include "ncde_common.h"

typedef void (*NcdeFastPlanKernel)(KArgs);
// shape 0 = (H, HH, C) = (32, 32, 20), 1 = (64, 64, 4); hp 1 = split-fp16 (default), 0 = split-bf16 (also the re-execution instance)
NcdeFastPlanKernel ncde_fast_plan_fwd(int shape, int interp, int method, int hp);
// continuous adjoint of (32, 32, 20), nl = 3: hp 2 (default) / 0; LDS bytes of that instance
NcdeFastPlanKernel ncde_fast_plan_adj3(int n_layers, int interp, int method, int hp);
size_t ncde_fast_plan_adj3_lds(int n_layers, int interp, int method, int hp);
