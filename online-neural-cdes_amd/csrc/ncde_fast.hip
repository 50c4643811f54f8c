// Shape-specialised kernels (placeholder: none registered yet).
#include "ncde_fast.h"

bool ncde_fast_supported(const NcdeProblem*, int) { return false; }
const char* ncde_fast_kernel_name(const NcdeProblem*, int) { return nullptr; }
int64_t ncde_fast_workspace_bytes(const NcdeProblem*, int) { return NCDE_ERR_UNSUPPORTED; }
int ncde_fast_forward(const NcdeProblem*, float*, void*, size_t, hipStream_t) { return NCDE_ERR_UNSUPPORTED; }
int ncde_fast_adjoint(const NcdeProblem*, const float*, const float*, const NcdeGrads*, void*, size_t, hipStream_t, bool) {
    return NCDE_ERR_UNSUPPORTED;
}
