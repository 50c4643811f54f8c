// Fused dopri5 attempt kernels (csrc/ncde_adaptive_fast.hip); driven by ncde_dp_solve (ncde_adaptive.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

#include "ncde_hip.h"

bool ncde_dpf_supported(const NcdeProblem* p, int adj);
const char* ncde_dpf_kernel_name(const NcdeProblem* p, int adj);
size_t ncde_dpf_pack_floats(const NcdeProblem* p, int adj);      // workspace floats of the per-lane weight image (0: not supported)
size_t ncde_dpf_partial_floats(const NcdeProblem* p, int theta1);
int ncde_dpf_reduce_blocks(const NcdeProblem* p, int theta1);
int ncde_dpf_prepare(const NcdeProblem* p, const void* dp_args, size_t dp_args_bytes, int adj, hipStream_t st);
// enqueue `rounds` attempt launches (launches of a finished solve exit at once); dp_args = the caller's DpArgs block
int ncde_dpf_launch(const NcdeProblem* p, const void* dp_args, size_t dp_args_bytes, int adj, int rounds, hipStream_t st);
// reverse sweep of a taped solve (adjoint=False) on the fused stage machinery; workspace: ncde_dpf_pack_floats(p, 1) floats at DpArgs.WP
bool ncde_dpf_tape_supported(const NcdeProblem* p);
const char* ncde_dpf_tape_kernel_name(const NcdeProblem* p);
int ncde_dpf_tape_launch(const NcdeProblem* p, const void* dp_args, size_t dp_args_bytes, hipStream_t st);
