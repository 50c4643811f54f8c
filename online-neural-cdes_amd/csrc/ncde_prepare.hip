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
// One series (one sample, one channel); xs / os are strided views (global memory or an LDS copy).
// mode 0: linear knots, NaN filled (L rows out);  1: rectilinear value channel (2L-1 rows out);  2: rectilinear time channel
__device__ void linear_series(const float* xs, long long sx, float* os, long long so, int L, int mode) {
    if (mode == 2) {  // time channel: t_0, t_1, t_1, t_2, t_2, ... (advanced by one slot)
        const int T = 2 * L - 1;
        for (int i = 0; i < L; ++i) {
            const float v = xs[i * sx];
            if (i > 0) os[(2 * i - 1) * so] = v;
            if (2 * i < T) os[(2 * i) * so] = v;
        }
        return;
    }
    if (mode == 1) {  // forward fill; a leading gap takes the first observation (0 if there is none)
        const int T = 2 * L - 1;
        float first = 0.0f;
        for (int i = 0; i < L; ++i) {
            const float v = xs[i * sx];
            if (!isnan(v)) { first = v; break; }
        }
        float last = first;
        for (int i = 0; i < L; ++i) {
            const float v = xs[i * sx];
            if (!isnan(v)) last = v;
            os[(2 * i) * so] = last;
            if (2 * i + 1 < T) os[(2 * i + 1) * so] = last;
        }
        return;
    }
    int prev = -1;  // index of the last observation seen
    float prev_v = 0.0f;
    for (int i = 0; i < L; ++i) {
        const float v = xs[i * sx];
        if (isnan(v)) continue;
        if (prev < 0) {
            for (int k = 0; k < i; ++k) os[k * so] = v;
        } else {
            for (int k = prev + 1; k < i; ++k) {
                const float ratio = ((float)k - (float)prev) / ((float)i - (float)prev);
                os[k * so] = prev_v + ratio * (v - prev_v);
            }
        }
        os[i * so] = v;
        prev = i;
        prev_v = v;
    }
    if (prev < 0) {
        for (int k = 0; k < L; ++k) os[k * so] = 0.0f;
    } else {
        for (int k = prev + 1; k < L; ++k) os[k * so] = prev_v;
    }
}

// v1: one thread per series straight on global memory (any size; coalescing limited to C*4-byte runs)
__global__ __launch_bounds__(256) void ncde_linear_coeffs_kernel(const float* __restrict__ x, int B, int L, int C, int rect,
                                                                  float* __restrict__ out) {
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= (long long)B * C) return;
    const int b = (int)(tid / C), c = (int)(tid - (long long)b * C);
    const int T = rect >= 0 ? 2 * L - 1 : L;
    linear_series(x + (long long)b * L * C + c, C, out + (long long)b * T * C + c, C, L, rect < 0 ? 0 : (c == rect ? 2 : 1));
}

// v2: one workgroup per sample, the whole [L][C] series staged in LDS: fully coalesced loads and stores, and the
// per-channel fills as WAVE SCANS along time (lane <-> time step): index of the last observation at or before i
// (inclusive prefix max) and of the next one at or after i (suffix min), 64 time steps per scan, carried across chunks.
// The arithmetic per element is exactly linear_series()'s, so the result is bit-identical to v1.
__device__ __forceinline__ int wave_prefix_max(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d, 64);
        if (lane >= d) v = max(v, o);
    }
    return v;
}
__device__ __forceinline__ int wave_suffix_min(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_down(v, d, 64);
        if (lane + d < 64) v = min(v, o);
    }
    return v;
}

__global__ __launch_bounds__(256) void ncde_linear_coeffs_lds_kernel(const float* __restrict__ x, int B, int L, int C, int rect,
                                                                     float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int T = rect >= 0 ? 2 * L - 1 : L;
    float* xin = sm;                                   // [L][C] raw
    int* prv = reinterpret_cast<int*>(sm + L * C);     // [L][C] index of the last observation <= i (-1: none)
    int* nxt = prv + L * C;                            // [L][C] index of the next observation >= i (L: none)   (linear fill only)
    const float* src = x + (long long)b * L * C;
    if (((L * C) & 3) == 0 && ((((long long)b * L * C) & 3) == 0)) {
        for (int e = tid; e < (L * C) >> 2; e += 256) reinterpret_cast<float4*>(xin)[e] = reinterpret_cast<const float4*>(src)[e];
    } else {
        for (int e = tid; e < L * C; e += 256) xin[e] = src[e];
    }
    __syncthreads();
    // scans along time, thread <-> (time chunk k, channel c): a short serial pass over the chunk, then the carry from the
    // nearest earlier (later) chunk that holds an observation
    {
        const int nch = C <= 256 ? (256 / C < 64 ? 256 / C : 64) : 1;       // chunks per channel
        const int cl = (L + nch - 1) / nch;                                 // time steps per chunk
        int* car = nxt + (rect < 0 ? L * C : 0);                            // [nch][C] chunk summaries (behind the index arrays)
        const int k = tid / C, c = tid - k * C;
        const bool act = C <= 256 && k < nch;
        const int i0 = k * cl, i1 = min(L, i0 + cl);
        for (int cc = act ? c : C; cc < C; cc += 256) {
            int last = -1;
            for (int i = i0; i < i1; ++i) {
                if (!isnan(xin[i * C + cc])) last = i;
                prv[i * C + cc] = last;
            }
            car[k * C + cc] = last;
        }
        __syncthreads();
        for (int cc = act ? c : C; cc < C; cc += 256) {
            int cin = -1;
            for (int kk = k - 1; kk >= 0 && cin < 0; --kk) cin = car[kk * C + cc];
            for (int i = i0; i < i1 && prv[i * C + cc] < 0; ++i) prv[i * C + cc] = cin;
        }
        if (rect < 0) {
            __syncthreads();
            for (int cc = act ? c : C; cc < C; cc += 256) {
                int next = L;
                for (int i = i1 - 1; i >= i0; --i) {
                    if (!isnan(xin[i * C + cc])) next = i;
                    nxt[i * C + cc] = next;
                }
                car[k * C + cc] = next;
            }
            __syncthreads();
            for (int cc = act ? c : C; cc < C; cc += 256) {
                int cin = L;
                for (int kk = k + 1; kk < nch && cin >= L; ++kk) cin = car[kk * C + cc];
                for (int i = i1 - 1; i >= i0 && nxt[i * C + cc] >= L; --i) nxt[i * C + cc] = cin;
            }
        }
    }
    __syncthreads();
    float* dst = out + (long long)b * T * C;
    // output: thread <-> (row offset, channel), rows strided by the rows one pass covers: no division in the loops,
    // consecutive threads write consecutive addresses
    const int rp = 256 / C > 0 ? 256 / C : 1;
    const int r0 = tid / C, c = tid - r0 * C;
    const bool active = C <= 256 ? r0 < rp : true;
    if (rect >= 0 && (C & 3) == 0 && C <= 1024 && ((((long long)b * T * C) & 3) == 0)) {
        // 16-byte stores: thread <-> (row offset, channel quad)
        const int CQ = C >> 2, rp4 = 256 / CQ > 0 ? 256 / CQ : 1;
        const int q0 = tid / CQ, cq = tid - q0 * CQ;
        if (q0 < rp4) {
            int first[4] = {-2, -2, -2, -2};
            for (int t = q0; t < T; t += rp4) {
                float4 v4;
                float* vv = reinterpret_cast<float*>(&v4);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int cc = 4 * cq + u;
                    if (cc == rect) {
                        vv[u] = xin[((t + 1) >> 1) * C + cc];
                    } else {
                        int p = prv[(t >> 1) * C + cc];
                        if (p < 0) {
                            if (first[u] == -2) {
                                first[u] = -1;
                                if (prv[(L - 1) * C + cc] >= 0) {
                                    int lo = 0, hi = L - 1;
                                    while (lo < hi) {
                                        const int mid = (lo + hi) >> 1;
                                        if (prv[mid * C + cc] >= 0) hi = mid; else lo = mid + 1;
                                    }
                                    first[u] = lo;
                                }
                            }
                            p = first[u];
                        }
                        vv[u] = p >= 0 ? xin[p * C + cc] : 0.0f;
                    }
                }
                *reinterpret_cast<float4*>(dst + t * C + 4 * cq) = v4;
            }
        }
    } else if (rect >= 0) {
        // value channels = forward fill of row t/2 (leading gap: the first observation, 0 if none);
        // time channel advanced by one slot: row (t+1)/2
        int first = -2;                                 // first observation of this channel, found lazily (-1: none)
        for (int cc = c; cc < C && active; cc += 256) {
            for (int t = r0; t < T; t += rp) {
                float v;
                if (cc == rect) {
                    v = xin[((t + 1) >> 1) * C + cc];
                } else {
                    int p = prv[(t >> 1) * C + cc];
                    if (p < 0) {
                        if (first == -2) {              // smallest j with prv[j] >= 0 (prv is monotone in j)
                            first = -1;
                            if (prv[(L - 1) * C + cc] >= 0) {
                                int lo = 0, hi = L - 1;
                                while (lo < hi) {
                                    const int mid = (lo + hi) >> 1;
                                    if (prv[mid * C + cc] >= 0) hi = mid; else lo = mid + 1;
                                }
                                first = lo;
                            }
                        }
                        p = first;
                    }
                    v = p >= 0 ? xin[p * C + cc] : 0.0f;
                }
                dst[t * C + cc] = v;
            }
            first = -2;
        }
    } else {
        for (int cc = c; cc < C && active; cc += 256) {
            for (int i = r0; i < L; i += rp) {
                const int e = i * C + cc;
                const int p = prv[e], q = nxt[e];
                float v;
                if (p == i) v = xin[e];
                else if (p < 0) v = q < L ? xin[q * C + cc] : 0.0f;                // leading gap / no observation
                else if (q >= L) v = xin[p * C + cc];                              // trailing gap
                else {
                    const float pv = xin[p * C + cc], qv = xin[q * C + cc];
                    const float ratio = ((float)i - (float)p) / ((float)q - (float)p);
                    v = pv + ratio * (qv - pv);
                }
                dst[e] = v;
            }
        }
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

__global__ __launch_bounds__(64) void ncde_cubic_coeffs_kernel(const float* __restrict__ x, int B, int L, int C,
                                                                 float* __restrict__ out, float* __restrict__ ws,
                                                                 const float* __restrict__ diag_swept, float* __restrict__ ws_d) {
    // one wave per workgroup: the serial recurrences are latency-bound, so many small workgroups (all CUs, many waves in
    // flight) beat few large ones -- 32768 series at cfg4 are only 128 workgroups of 256
    const long long tid = (long long)blockIdx.x * 64 + threadIdx.x;
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
    const int rect = rectilinear_time_index < 0 ? -1 : rectilinear_time_index;
    const int T = rect >= 0 ? 2 * L - 1 : L;
    (void)T;
    const size_t lds = sizeof(float) * ((size_t)L * C * (rect >= 0 ? 2 : 3) + 64 * (size_t)C);
    if (lds <= 64 * 1024 && C <= 256) {   // the series fits LDS: staged, coalesced, scan-parallel variant
        if (hipFuncSetAttribute((const void*)ncde_linear_coeffs_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return NCDE_ERR_HIP;
        hipLaunchKernelGGL(ncde_linear_coeffs_lds_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, x, B, L, C, rect, out);
    } else {
        hipLaunchKernelGGL(ncde_linear_coeffs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, B, L, C, rect, out);
    }
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
    hipLaunchKernelGGL(ncde_cubic_coeffs_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, x, B, L, C, out,
                       ws, diag, ws_d);
    return hipGetLastError() == hipSuccess ? NCDE_OK : NCDE_ERR_HIP;
}

}  // extern "C"
