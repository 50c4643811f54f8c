// Host-side time-plan builder (csrc/ncde_timeplan.hip); the public entry points are in include/ncde_hip.h.
#pragma once
#include <cstddef>

#include "ncde_hip.h"

int ncde_time_plan_build_impl(const NcdeProblem* p, const NcdeTimeSpec* ts, void* host_buffer, size_t bytes, NcdeTimePlanInfo* info,
                              char* err, size_t errn);
