// Further layer counts of the chain + gradient-wave adjoint of the (32, 32, 20) shape (ncde_adj_fast3, ncde_fast.hip): its own
// translation unit (ncde_fast_nl.hip) so that the instantiations build in parallel with the rest.
#pragma once
#include "ncde_common.h"

typedef void (*NcdeFastNlKernel)(KArgs);
// kernel for n_layers in {1, 2, 4} (4: linear control path only -- the cubic path's LDS plan is 224 bytes over the 160 KB), hp in
// {0 (all split-bf16: also the re-execution instance), 2 (default)}, or nullptr
NcdeFastNlKernel ncde_fast_adj3_nl(int n_layers, int interp, int method, int hp, bool discrete);
size_t ncde_fast_adj3_nl_lds(int n_layers, int interp, int hp);
const char* ncde_fast_adj3_nl_name(int n_layers, int hp, bool discrete);
