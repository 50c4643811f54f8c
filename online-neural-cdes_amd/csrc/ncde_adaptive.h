// Adaptive dopri5 (csrc/ncde_adaptive.hip); the public entry points are in include/ncde_hip.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <vector>

#include "ncde_hip.h"

int64_t ncde_dp_workspace_bytes(const NcdeProblem* p, int n_t, int adj);
bool ncde_dp_supported(const NcdeProblem* p, int adj, char* why, size_t n);
bool ncde_dp_tape_supported(const NcdeProblem* p, char* why, size_t n);      // the taped reverse sweep's own limits (hidden <= 128, LDS)
int ncde_dp_solve(const NcdeProblem* p, const NcdeTimeSpec* ts, const NcdeAdaptiveOptions* op, int adj, float* out, const float* z_out,
                  const float* grad_out, const NcdeGrads* g, void* ws, size_t ws_bytes, hipStream_t st, NcdeAdaptiveStats* stats,
                  char* err, size_t errn, void* record = nullptr, size_t record_bytes = 0, const double* replay = nullptr, int replay_n = 0);
// taped solve (adjoint=False): record size, reverse sweep (workspace: ncde_dp_workspace_bytes(p, n_t, 2))
int64_t ncde_dp_record_bytes(const NcdeProblem* p, const NcdeTimeSpec* ts, const NcdeAdaptiveOptions* op);
int ncde_dp_tape_backward_run(const NcdeProblem* p, const NcdeTimeSpec* ts, const NcdeAdaptiveOptions* op, const void* record, size_t record_bytes,
                              const float* grad_out, const NcdeGrads* g, void* ws, size_t ws_bytes, hipStream_t st, char* err, size_t errn);
