// Batch-tiled (large-hidden) kernel family: host-side hooks used by ncde_abi.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "ncde_hip.h"

bool ncde_tiled_supported(const NcdeProblem* p, int pass);
bool ncde_tiled_preferred(const NcdeProblem* p, int pass);
const char* ncde_tiled_kernel_name(const NcdeProblem* p, int pass);
int64_t ncde_tiled_workspace_bytes(const NcdeProblem* p, int pass);
int ncde_tiled_forward(const NcdeProblem* p, float* out, float* stages, void* ws, size_t ws_bytes, hipStream_t st);
int ncde_tiled_adjoint(const NcdeProblem* p, const float* src, const float* grad_out, const NcdeGrads* g, void* ws, size_t ws_bytes,
                       hipStream_t st, bool main_kernel_only, bool discrete);
// byte offset, in the workspace of pass `pass`, of the call's cooperative status word (0 = the cooperative launches ran; 1 = one gave up
// and the per-workgroup kernels re-executed the pass), or -1 when this problem / pass launches nothing cooperative
int64_t ncde_tiled_status_offset(const NcdeProblem* p, int pass);
