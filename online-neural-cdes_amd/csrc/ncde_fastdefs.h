// Device-side pieces shared by the register-resident kernels (ncde_fast.hip, ncde_fast4.hip).
#pragma once
#include "ncde_common.h"
#include "ncde_bf3.h"

namespace {

template <int METHOD>
struct Combine {
    // Butcher bookkeeping on the default axis (dt = 1); returns the next stage input (or the new state after the last stage).
    // Operation order of fixed_grid.py:6-29 / rk_common.py:106-114.
    static __device__ __forceinline__ float apply(int j, float k, float& y0, float& k1, float& k2) {
        if constexpr (METHOD == NCDE_RK4_38) {
            if (j == 0) { k1 = k; return y0 + k * 0.333333343267440796f; }
            if (j == 1) { k2 = k; return y0 + (k - k1 * 0.333333343267440796f); }
            if (j == 2) { const float ys = y0 + ((k1 - k2) + k); k2 = k2 + k; return ys; }
            y0 = y0 + ((k1 + 3.0f * k2) + k) * 0.125f;
            return y0;
        } else if constexpr (METHOD == NCDE_MIDPOINT) {
            if (j == 0) return y0 + k * 0.5f;
            y0 = y0 + k;
            return y0;
        } else {
            y0 = y0 + k;
            return y0;
        }
    }
};

// DS instructions of one wave are executed by the LDS in issue order, so a ds_read that follows a ds_write
// of the same wave (any lanes) sees the data without an s_waitcnt; only the COMPILER must keep the order.
__device__ __forceinline__ void wave_lds_order() { asm volatile("" ::: "memory"); }

template <int METHOD> constexpr int kStages = METHOD == NCDE_RK4_38 ? 4 : (METHOD == NCDE_MIDPOINT ? 2 : 1);

// relu on the bit pattern: max_i32(bits, 0) is 0 for every negative float (and -0.0) and the identity otherwise -- ONE
// VALU op, no canonicalisation of the MFMA result (v_max_f32 / v_med3_f32 get a v_max x,x,x in front).
__device__ __forceinline__ float relu_bits(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

}  // namespace
