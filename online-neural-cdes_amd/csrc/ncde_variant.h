// Vector-field variants (gated fields, evaluate / derivative input modes): host-side hooks used by ncde_abi.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "ncde_hip.h"

bool ncde_variant_supported(const NcdeProblem* p, int pass);
int64_t ncde_variant_workspace_bytes(const NcdeProblem* p, int pass);
int ncde_variant_forward(const NcdeProblem* p, float* out, float* stages, hipStream_t st);
int ncde_variant_adjoint(const NcdeProblem* p, const float* src, const float* grad_out, const NcdeGrads* g, void* ws, hipStream_t st,
                         bool main_kernel_only, bool discrete);
