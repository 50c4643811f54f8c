// Control-path coefficient construction on the GPU (the step immediately before the
// hot path).  HBM-bound per-series scans: one thread per (sample, channel) series, adjacent threads = adjacent
// channels so every time step is a coalesced row access.  Reference semantics restated (relative to
// /root/reference/modules/torchcde/torchcde):
//   forward fill + rectilinear interleave     interpolation_linear.py:85-128, misc.py:103-126
//   NaN fill of the linear knots              interpolation_linear.py:13-82, 131-180
//   natural cubic coefficients                interpolation_cubic.py:7-53 (tridiagonal solve: misc.py:13-67)
// All on the default integer time grid, fp32, operation order of the reference (build uses -ffp-contract=off).
#include <hip/hip_runtime.h>

#include <cmath>

#include "ncde_hip.h"

namespace {

// linear knots with missing values: leading gap <- first observation, trailing gap <- last observation,
// interior gaps <- linear interpolation between the neighbours, all-NaN series <- 0.
// rect >= 0: rectilinear preparation first (forward fill, every row twice, time channel advanced one slot).
__global__ __launch_bounds__(256) void ncde_linear_coeffs_kernel(const float* __restrict__ x, int B, int L, int C, int rect,
                                                                  float* __restrict__ out) {
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= (long long)B * C) return;
    const int b = (int)(tid / C), c = (int)(tid - (long long)b * C);
    const float* xs = x + (long long)b * L * C + c;
    if (rect >= 0) {
        const int T = 2 * L - 1;
        float* os = out + (long long)b * T * C + c;
        if (c == rect) {  // time channel: t_0, t_1, t_1, t_2, t_2, ... (advanced by one slot)
            for (int i = 0; i < L; ++i) {
                const float v = xs[(long long)i * C];
                if (i > 0) os[(long long)(2 * i - 1) * C] = v;
                if (2 * i < T) os[(long long)(2 * i) * C] = v;
            }
            return;
        }
        // value channels: forward fill; a leading gap takes the first observation (0 if there is none)
        float first = 0.0f;
        for (int i = 0; i < L; ++i) {
            const float v = xs[(long long)i * C];
            if (!isnan(v)) { first = v; break; }
        }
        float last = first;
        for (int i = 0; i < L; ++i) {
            const float v = xs[(long long)i * C];
            if (!isnan(v)) last = v;
            os[(long long)(2 * i) * C] = last;
            if (2 * i + 1 < T) os[(long long)(2 * i + 1) * C] = last;
        }
        return;
    }
    float* os = out + (long long)b * L * C + c;
    int prev = -1;  // index of the last observation seen
    float prev_v = 0.0f;
    for (int i = 0; i < L; ++i) {
        const float v = xs[(long long)i * C];
        if (isnan(v)) continue;
        if (prev < 0) {
            for (int k = 0; k < i; ++k) os[(long long)k * C] = v;
        } else {
            for (int k = prev + 1; k < i; ++k) {
                const float ratio = ((float)k - (float)prev) / ((float)i - (float)prev);
                os[(long long)k * C] = prev_v + ratio * (v - prev_v);
            }
        }
        os[(long long)i * C] = v;
        prev = i;
        prev_v = v;
    }
    if (prev < 0) {
        for (int k = 0; k < L; ++k) os[(long long)k * C] = 0.0f;
    } else {
        for (int k = prev + 1; k < L; ++k) os[(long long)k * C] = prev_v;
    }
}

// natural cubic spline on the integer grid.  out[b][p][4C] = a | b | 2c | 3d of piece p.
// ws[b][i][c] keeps the forward-swept right-hand side; the swept diagonal depends on i only (recomputed).
// Series with missing values (interpolation_cubic.py:77-165): ends filled from the first / last observation, spline
// through the observed knots only (non-uniform spacing, so the swept diagonal is per series: wd), then every unit
// interval [time, time+1) gets the piece that covers it, re-expanded around `time` (offset = t_knot - time).
__device__ void cubic_series_missing(const float* xs, float* os, float* wb, float* wd, int L, int C, int first, int last) {
    const long long sC = C;
    auto val = [&](int i) { return i < first ? xs[first * sC] : (i > last ? xs[last * sC] : xs[i * sC]); };
    // forward sweep over the knots: knot p is processed when the next knot q is known (r_p = 1/(t_q - t_p))
    int pp = -1, p = 0;                 // index 0 is always a knot after the fill
    float xp = val(0), r_prev = 0.0f, scaled_prev = 0.0f, nd_prev = 0.0f, nb_prev = 0.0f;
    for (int q = 1; q < L; ++q) {
        const float xq = val(q);
        if (isnan(xq)) continue;
        const float td = (float)q - (float)p;
        const float r = 1.0f / td, r2 = r * r;
        const float three = 3.0f * (xq - xp);
        const float scaled = three * r2;
        float diag, rhs;
        if (pp < 0) { diag = r * 2.0f; rhs = scaled; }
        else { diag = (r + r_prev) * 2.0f; rhs = scaled + scaled_prev; }
        float nd, nb;
        if (pp < 0) { nd = diag; nb = rhs; }
        else {
            const float w = r_prev / nd_prev;
            nd = diag - w * r_prev;
            nb = rhs - w * nb_prev;
        }
        wd[p * sC] = nd;
        wb[p * sC] = nb;
        pp = p; p = q; xp = xq; r_prev = r; scaled_prev = scaled; nd_prev = nd; nb_prev = nb;
    }
    // last knot (index L-1): diag = (0 + r_prev) * 2, rhs = 0 + scaled_prev
    float kd_next;
    {
        const float diag = (0.0f + r_prev) * 2.0f, rhs = 0.0f + scaled_prev;
        const float w = r_prev / nd_prev;
        const float nd = diag - w * r_prev, nb = rhs - w * nb_prev;
        kd_next = nb / nd;
    }
    // back substitution, coefficients per knot interval, expansion onto the unit intervals (right to left)
    int q = L - 1;
    float xq = val(q);
    for (int i = L - 2; i >= 0; --i) {
        const float xi = val(i);
        if (isnan(xi)) continue;
        const float td = (float)q - (float)i;
        const float r = 1.0f / td, r2 = r * r;
        const float kd = (wb[i * sC] - r * kd_next) / wd[i * sC];
        const float three = 3.0f * (xq - xi), six = 2.0f * three;
        const float two_c = ((six * r - 4.0f * kd) - 2.0f * kd_next) * r;
        const float three_d = (-six * r + 3.0f * (kd + kd_next)) * r2;
        for (int time = q - 1; time >= i; --time) {
            const float off = (float)i - (float)time;
            const float a_in = (0.5f * two_c - three_d * off / 3.0f) * off;
            float* o = os + (long long)time * 4 * C;
            o[0] = xi + (a_in - kd) * off;
            o[C] = kd + (three_d * off - two_c) * off;
            o[2 * C] = two_c - 2.0f * three_d * off;
            o[3 * C] = three_d;
        }
        q = i; xq = xi; kd_next = kd;
    }
}

__global__ __launch_bounds__(256) void ncde_cubic_coeffs_kernel(const float* __restrict__ x, int B, int L, int C,
                                                                 float* __restrict__ out, float* __restrict__ ws,
                                                                 const float* __restrict__ diag_swept, float* __restrict__ ws_d) {
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= (long long)B * C) return;
    const int b = (int)(tid / C), c = (int)(tid - (long long)b * C);
    const float* xs = x + (long long)b * L * C + c;
    float* os = out + (long long)b * (L - 1) * 4 * C + c;
    // missing values?  (NaN anywhere in the series)
    int first = -1, last = -1, n_obs = 0;
    for (int i = 0; i < L; ++i)
        if (!isnan(xs[(long long)i * C])) {
            if (first < 0) first = i;
            last = i;
            ++n_obs;
        }
    if (n_obs == 0) {  // no observation at all: the zero path
        for (int i = 0; i < L - 1; ++i) {
            float* o = os + (long long)i * 4 * C;
            o[0] = 0.0f; o[C] = 0.0f; o[2 * C] = 0.0f; o[3 * C] = 0.0f;
        }
        return;
    }
    if (L == 2) {
        const float x0 = xs[first * (long long)C], x1 = xs[last * (long long)C];   // ends filled from the observations
        os[0] = x0;
        os[C] = x1 - x0;
        os[2 * C] = 0.0f;
        os[3 * C] = 0.0f;
        return;
    }
    float* wb = ws + (long long)b * L * C + c;
    if (n_obs < L) {
        cubic_series_missing(xs, os, wb, ws_d + (long long)b * L * C + c, L, C, first, last);
        return;
    }
    // rhs_i = 3 (x_i - x_{i-1}) + 3 (x_{i+1} - x_i) with unit knot spacing; Thomas forward sweep
    float x_prev = xs[0], x_cur = xs[C];
    float scaled_prev = (3.0f * (x_cur - x_prev)) * 1.0f;  // three_path_diffs * reciprocal^2
    float nb_prev = scaled_prev;                            // rhs_0
    wb[0] = nb_prev;
    for (int i = 1; i < L; ++i) {
        float rhs;
        if (i < L - 1) {
            const float x_next = xs[(long long)(i + 1) * C];
            const float scaled = (3.0f * (x_next - x_cur)) * 1.0f;
            rhs = scaled + scaled_prev;
            scaled_prev = scaled;
            x_prev = x_cur;
            x_cur = x_next;
        } else {
            rhs = 0.0f + scaled_prev;
        }
        const float w = 1.0f / diag_swept[i - 1];
        const float nb = rhs - w * nb_prev;
        wb[(long long)i * C] = nb;
        nb_prev = nb;
    }
    // back substitution and coefficients, last piece first
    float kd_next = nb_prev / diag_swept[L - 1];
    for (int i = L - 2; i >= 0; --i) {
        const float kd = (wb[(long long)i * C] - 1.0f * kd_next) / diag_swept[i];
        const float xi = xs[(long long)i * C], xi1 = xs[(long long)(i + 1) * C];
        const float six = 2.0f * (3.0f * (xi1 - xi));
        float* o = os + (long long)i * 4 * C;
        o[0] = xi;
        o[C] = kd;
        o[2 * C] = (six * 1.0f - 4.0f * kd - 2.0f * kd_next) * 1.0f;
        o[3 * C] = (-six * 1.0f + 3.0f * (kd + kd_next)) * 1.0f;
        kd_next = kd;
    }
}

// swept diagonal of the natural-spline system on the unit grid: d_0 = 2, d_i = diag_i - (1/d_{i-1}) * 1
__global__ void ncde_cubic_diag_kernel(int L, float* diag_swept) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float d_prev = 0.0f;
    for (int i = 0; i < L; ++i) {
        float diag = (i < L - 1 ? 1.0f : 0.0f) + (i > 0 ? 1.0f : 0.0f);
        diag *= 2.0f;
        float d = diag;
        if (i > 0) {
            const float w = 1.0f / d_prev;
            d = diag - w * 1.0f;
        }
        diag_swept[i] = d;
        d_prev = d;
    }
}


}  // namespace

extern "C" {

int64_t ncde_prepare_workspace_bytes(int kind, int B, int L, int C) {
    if (B < 1 || L < 2 || C < 1) return NCDE_ERR_INVALID;
    if (kind == NCDE_INTERP_CUBIC) return (int64_t)sizeof(float) * (2 * (int64_t)B * L * C + L) + 256;
    return 256;
}

int ncde_prepare_linear(const float* x, int B, int L, int C, int rectilinear_time_index, float* out, void* stream) {
    if (!x || !out || B < 1 || L < 2 || C < 1 || rectilinear_time_index >= C) return NCDE_ERR_INVALID;
    const long long n = (long long)B * C;
    hipLaunchKernelGGL(ncde_linear_coeffs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, B, L, C,
                       rectilinear_time_index < 0 ? -1 : rectilinear_time_index, out);
    return hipGetLastError() == hipSuccess ? NCDE_OK : NCDE_ERR_HIP;
}

int ncde_prepare_cubic(const float* x, int B, int L, int C, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !out || !workspace || B < 1 || L < 2 || C < 1) return NCDE_ERR_INVALID;
    if ((int64_t)workspace_bytes < ncde_prepare_workspace_bytes(NCDE_INTERP_CUBIC, B, L, C)) return NCDE_ERR_WORKSPACE;
    float* ws = (float*)workspace;
    float* ws_d = ws + (size_t)B * L * C;
    float* diag = ws_d + (size_t)B * L * C;
    hipLaunchKernelGGL(ncde_cubic_diag_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, L, diag);
    const long long n = (long long)B * C;
    hipLaunchKernelGGL(ncde_cubic_coeffs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, B, L, C, out,
                       ws, diag, ws_d);
    return hipGetLastError() == hipSuccess ? NCDE_OK : NCDE_ERR_HIP;
}

}  // extern "C"
