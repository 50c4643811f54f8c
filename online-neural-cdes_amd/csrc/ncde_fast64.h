// In-sweep adjoint of the register-resident family for H = HH = 64, C <= 4 (BASELINE cfg4): host-side hooks used by ncde_fast.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "ncde_hip.h"

// pass 1 = continuous adjoint (adjoint.py:37-145), pass 2 = exact discrete backward (ncde_backward)
bool ncde_fast64_supported(const NcdeProblem* p, int pass);
const char* ncde_fast64_kernel_name(const NcdeProblem* p, int pass);
int64_t ncde_fast64_workspace_bytes(const NcdeProblem* p, int pass);
int ncde_fast64_adjoint(const NcdeProblem* p, const float* src, const float* grad_out, const NcdeGrads* g, void* ws, size_t ws_bytes,
                        hipStream_t st, bool main_kernel_only, bool discrete);
