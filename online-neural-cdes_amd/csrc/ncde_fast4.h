// Decoupled-chain adjoint of the register-resident family (ncde_fast4.hip): host-side hooks used by ncde_fast.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "ncde_common.h"

typedef void (*NcdeFast4Kernel)(KArgs);
// kernel for (n_layers, channels) at H = HH = 32, or nullptr when that shape is not instantiated
NcdeFast4Kernel ncde_fast4_pick(int n_layers, int channels, int interp, int method, bool discrete, bool profile);
size_t ncde_fast4_lds_bytes(int n_layers, int channels, int interp);
