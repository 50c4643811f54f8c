// ncde_dwo_h2 (ncde_dwo2.hip): output-layer gradient pass for the cooperative sweep's 2-piece fp16 records; launched by ncde_tiled.hip
#pragma once
#include <hip/hip_runtime.h>

#include "ncde_common.h"
extern "C" __global__ void ncde_dwo_h2(KArgs a, int n_sc, int n_st, float* gpartB);
