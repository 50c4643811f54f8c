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
#include "ncde_host.h"

namespace {

// linear knots with missing values: leading gap <- first observation, trailing gap <- last observation,
// interior gaps <- linear interpolation between the neighbours, all-NaN series <- 0.
// rect >= 0: rectilinear preparation first (forward fill, every row twice, time channel advanced one slot).
// One series (one sample, one channel); xs / os are strided views (global memory or an LDS copy).
// mode 0: linear knots, NaN filled (L rows out);  1: rectilinear value channel (2L-1 rows out);  2: rectilinear time channel
// tg: the user time grid the observations sit on (interpolation_linear.py:131-180 with t=...), NULL = the integer grid; it only
// enters the ratio of the interior-gap fill (the rectilinear preparation leaves no interior gap).
__device__ void linear_series(const float* xs, long long sx, float* os, long long so, int L, int mode, const float* tg = nullptr) {
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
                const float ratio = tg ? (tg[k] - tg[prev]) / (tg[i] - tg[prev]) : ((float)k - (float)prev) / ((float)i - (float)prev);
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
                                                                  float* __restrict__ out, const float* __restrict__ tg) {
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= (long long)B * C) return;
    const int b = (int)(tid / C), c = (int)(tid - (long long)b * C);
    const int T = rect >= 0 ? 2 * L - 1 : L;
    linear_series(x + (long long)b * L * C + c, C, out + (long long)b * T * C + c, C, L, rect < 0 ? 0 : (c == rect ? 2 : 1), tg);
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
                                                                     float* __restrict__ out, const float* __restrict__ tg) {
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
                    const float ratio = tg ? (tg[i] - tg[p]) / (tg[q] - tg[p]) : ((float)i - (float)p) / ((float)q - (float)p);
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
// tg: user time grid (interpolation_cubic.py:56-165 with t=...), NULL = integer grid: knot spacing and the re-expansion offsets.
__device__ void cubic_series_missing(const float* xs, float* os, float* wb, float* wd, int L, int C, int first, int last,
                                     const float* tg = nullptr) {
    const long long sC = C;
    auto val = [&](int i) { return i < first ? xs[first * sC] : (i > last ? xs[last * sC] : xs[i * sC]); };
    // forward sweep over the knots: knot p is processed when the next knot q is known (r_p = 1/(t_q - t_p))
    int pp = -1, p = 0;                 // index 0 is always a knot after the fill
    float xp = val(0), r_prev = 0.0f, scaled_prev = 0.0f, nd_prev = 0.0f, nb_prev = 0.0f;
    for (int q = 1; q < L; ++q) {
        const float xq = val(q);
        if (isnan(xq)) continue;
        const float td = tg ? tg[q] - tg[p] : (float)q - (float)p;
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
        const float td = tg ? tg[q] - tg[i] : (float)q - (float)i;
        const float r = 1.0f / td, r2 = r * r;
        const float kd = (wb[i * sC] - r * kd_next) / wd[i * sC];
        const float three = 3.0f * (xq - xi), six = 2.0f * three;
        const float two_c = ((six * r - 4.0f * kd) - 2.0f * kd_next) * r;
        const float three_d = (-six * r + 3.0f * (kd + kd_next)) * r2;
        for (int time = q - 1; time >= i; --time) {
            const float off = tg ? tg[i] - tg[time] : (float)i - (float)time;
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

// one (sample, channel) series straight on global memory: any size, missing values included
__device__ void cubic_series_global(const float* __restrict__ x, int L, int C, float* __restrict__ out, float* __restrict__ ws,
                                    const float* __restrict__ diag_swept, float* __restrict__ ws_d, int b, int c,
                                    const float* __restrict__ tg = nullptr) {
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
        os[C] = tg ? (x1 - x0) / (tg[1] - tg[0]) : x1 - x0;
        os[2 * C] = 0.0f;
        os[3 * C] = 0.0f;
        return;
    }
    float* wb = ws + (long long)b * L * C + c;
    if (n_obs < L || tg) {      // a user grid is non-uniform spacing: the same sweep as for missing values (every point a knot)
        cubic_series_missing(xs, os, wb, ws_d + (long long)b * L * C + c, L, C, first, last, tg);
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

__global__ __launch_bounds__(64) void ncde_cubic_coeffs_kernel(const float* __restrict__ x, int B, int L, int C,
                                                                 float* __restrict__ out, float* __restrict__ ws,
                                                                 const float* __restrict__ diag_swept, float* __restrict__ ws_d,
                                                                 const float* __restrict__ tg) {
    // one wave per workgroup: the serial recurrences are latency-bound, so many small workgroups (all CUs, many waves in
    // flight) beat few large ones -- 32768 series at cfg4 are only 128 workgroups of 256
    const long long tid = (long long)blockIdx.x * 64 + threadIdx.x;
    if (tid >= (long long)B * C) return;
    const int b = (int)(tid / C), c = (int)(tid - (long long)b * C);
    cubic_series_global(x, L, C, out, ws, diag_swept, ws_d, b, c, tg);
}

// LDS-staged variant for complete series (no missing values): one wave = NSMP whole samples (lane <-> (sample, channel)).
//   * the samples' [L][C] blocks come in with 16-byte loads (one contiguous region per workgroup) into LDS, sample stride
//     padded so that the 64 lanes of a time step hit distinct banks;
//   * the swept diagonal d_i and 1/d_{i-1} depend on i only (constant-coefficient system on the integer grid): read from a
//     table, so the forward sweep is one multiply-subtract per step and series, its result kept in LDS (no global scratch);
//   * back substitution is the chain only (the one true division per step: bit-exactness with the reference's x / d rules
//     out a reciprocal), its result kd_i written over the swept right-hand side in LDS;
//   * the a | b | 2c | 3d rows are then an embarrassingly parallel pass over (sample, piece, section, four channels): LDS
//     reads, one 16-byte store per item, consecutive lanes on consecutive addresses.
// Arithmetic per element is exactly cubic_series_global()'s, so the output is bit-identical (golden g8).  A workgroup
// that finds a NaN among its samples falls back to cubic_series_global for all of them.
#define CUBIC_NT 256
__global__ __launch_bounds__(CUBIC_NT) void ncde_cubic_coeffs_lds_kernel(const float* __restrict__ x, int B, int L, int C, int NSMP,
                                                                     int sstride, float* __restrict__ out, float* __restrict__ ws,
                                                                     const float* __restrict__ diag_swept, float* __restrict__ ws_d) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;                                 // [NSMP][sstride]  (sample block [L][C] + padding)
    float* nbs = xs + (size_t)NSMP * sstride;        // [NSMP][sstride]  forward-swept right-hand side
    float* dtab = nbs + (size_t)NSMP * sstride;      // [L] swept diagonal d_i
    float* wtab = dtab + ((L + 3) & ~3);             // [L] 1 / d_{i-1}
    const int lane = threadIdx.x;      // all CUBIC_NT threads stage and emit; the first 64 (one wave) run the chains
    const int b0 = blockIdx.x * NSMP;
    const int nsmp = min(NSMP, B - b0);
    const int LC = L * C;
    // ---- stage the samples (contiguous region of nsmp * L * C floats), detect missing values ------------------------------
    const float* src = x + (long long)b0 * LC;
    bool bad = false;
    if ((LC & 3) == 0 && (((uintptr_t)src) & 15) == 0) {
        const int nv = LC >> 2, tot = nsmp * nv;
        int e = lane;
        for (; e + 3 * CUBIC_NT < tot; e += 4 * CUBIC_NT) {      // four 16-byte loads in flight per lane
            float4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4*>(src + 4LL * (e + CUBIC_NT * q));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ee = e + CUBIC_NT * q, sm = ee / nv, k = ee - sm * nv;
                bad = bad || isnan(v[q].x) || isnan(v[q].y) || isnan(v[q].z) || isnan(v[q].w);
                *reinterpret_cast<float4*>(xs + (size_t)sm * sstride + 4 * k) = v[q];
            }
        }
        for (; e < tot; e += CUBIC_NT) {
            const int sm = e / nv, k = e - sm * nv;
            const float4 v = *reinterpret_cast<const float4*>(src + 4LL * e);
            bad = bad || isnan(v.x) || isnan(v.y) || isnan(v.z) || isnan(v.w);
            *reinterpret_cast<float4*>(xs + (size_t)sm * sstride + 4 * k) = v;
        }
    } else {
        for (int e = lane; e < nsmp * LC; e += CUBIC_NT) {
            const int sm = e / LC, k = e - sm * LC;
            const float v = src[e];
            bad = bad || isnan(v);
            xs[(size_t)sm * sstride + k] = v;
        }
    }
    for (int i = lane; i < L; i += CUBIC_NT) {
        dtab[i] = diag_swept[i];
        wtab[i] = i > 0 ? 1.0f / diag_swept[i - 1] : 0.0f;      // the same division cubic_series_global does per step
    }
    if (__syncthreads_or(bad)) {      // missing values somewhere in this workgroup: the general per-series path
        for (int e = lane; e < nsmp * C; e += CUBIC_NT) cubic_series_global(x, L, C, out, ws, diag_swept, ws_d, b0 + e / C, e % C);
        return;
    }
    const int sm = lane / C, c = lane - sm * C;
    const bool live = sm < nsmp && lane < 64;
    const float* xl = xs + (size_t)(live ? sm : 0) * sstride + c;
    float* nl = nbs + (size_t)(live ? sm : 0) * sstride + c;
    if (L == 2) {
        if (live) {
            float* os = out + (long long)(b0 + sm) * (L - 1) * 4 * C + c;
            os[0] = xl[0]; os[C] = xl[C] - xl[0]; os[2 * C] = 0.0f; os[3 * C] = 0.0f;
        }
        return;
    }
    // The two recurrences are latency chains (forward: multiply-subtract; backward: subtract + true division, ~60 cycles), so
    // everything that is NOT on the chain -- LDS reads, the right-hand side, the coefficient rows -- is done for a block of
    // CUBIC_BLK steps at a time around a register-only chain.
    constexpr int BLK = 8;
    // ---- forward sweep (same operation order as cubic_series_global) --------------------------------------------------------
    // full blocks are branch-free (guards inside an unrolled block turn every load into its own branch + wait); the tail
    // runs step by step
    if (lane < 64) {
        float x_cur = xl[C];
        float scaled_prev = (3.0f * (x_cur - xl[0])) * 1.0f;
        float nb_prev = scaled_prev;
        nl[0] = nb_prev;
        int i0 = 1;
        for (; i0 + BLK <= L - 1; i0 += BLK) {      // steps i0 .. i0+BLK-1, all < L-1
            float xn[BLK], wv[BLK], rhs[BLK], nbv[BLK];
#pragma unroll
            for (int k = 0; k < BLK; ++k) {
                xn[k] = xl[(i0 + k + 1) * C];
                wv[k] = wtab[i0 + k];               // 1 / d_{i-1}
            }
#pragma unroll
            for (int k = 0; k < BLK; ++k) {
                const float scaled = (3.0f * (xn[k] - x_cur)) * 1.0f;
                rhs[k] = scaled + scaled_prev;
                scaled_prev = scaled;
                x_cur = xn[k];
            }
#pragma unroll
            for (int k = 0; k < BLK; ++k) {
                nbv[k] = rhs[k] - wv[k] * nb_prev;
                nb_prev = nbv[k];
            }
#pragma unroll
            for (int k = 0; k < BLK; ++k) nl[(i0 + k) * C] = nbv[k];
        }
        for (int i = i0; i < L; ++i) {
            float rhs;
            if (i < L - 1) {
                const float x_next = xl[(i + 1) * C];
                const float scaled = (3.0f * (x_next - x_cur)) * 1.0f;
                rhs = scaled + scaled_prev;
                scaled_prev = scaled;
                x_cur = x_next;
            } else {
                rhs = 0.0f + scaled_prev;
            }
            const float nb = rhs - wtab[i] * nb_prev;
            nl[i * C] = nb;
            nb_prev = nb;
        }
    }
    // ---- back substitution: the chain only; kd_i overwrites the swept right-hand side in place --------------------------------
    if (lane < 64) {
        float kd_next = nl[(L - 1) * C] / dtab[L - 1];
        nl[(L - 1) * C] = kd_next;
        int ib = L - 2;
        for (; ib - BLK + 1 >= 0; ib -= BLK) {      // pieces ib .. ib-BLK+1
            float nbv[BLK], dv[BLK], kdv[BLK + 1];
#pragma unroll
            for (int k = 0; k < BLK; ++k) {
                nbv[k] = nl[(ib - k) * C];
                dv[k] = dtab[ib - k];
            }
            kdv[0] = kd_next;
#pragma unroll
            for (int k = 0; k < BLK; ++k) kdv[k + 1] = (nbv[k] - 1.0f * kdv[k]) / dv[k];      // the chain: one true division per step
#pragma unroll
            for (int k = 0; k < BLK; ++k) nl[(ib - k) * C] = kdv[k + 1];
            kd_next = kdv[BLK];
        }
        for (int i = ib; i >= 0; --i) {
            const float kd = (nl[i * C] - 1.0f * kd_next) / dtab[i];
            nl[i * C] = kd;
            kd_next = kd;
        }
    }
    __syncthreads();
    // ---- coefficient rows a | b | 2c | 3d: embarrassingly parallel over (sample, piece, section, channels) -----------------------
    // out[b][p][4C]; one item = one 16-byte store (four channels of one section) when C % 4 == 0, else one float
    const int C4 = 4 * C;
    if ((C & 3) == 0 && ((uintptr_t)out & 15) == 0) {
        const int cq = C >> 2, per_row = 4 * cq, per_smp = (L - 1) * per_row;
        for (int e = lane; e < nsmp * per_smp; e += CUBIC_NT) {
            const int s2 = e / per_smp, r = e - s2 * per_smp;
            const int pp = r / per_row, q4 = r - pp * per_row;
            const int part = q4 / cq, q = q4 - part * cq;
            const float* xa = xs + (size_t)s2 * sstride + pp * C + 4 * q;
            const float* ka = nbs + (size_t)s2 * sstride + pp * C + 4 * q;
            const float4 xi = *reinterpret_cast<const float4*>(xa), xi1 = *reinterpret_cast<const float4*>(xa + C);
            const float4 kd = *reinterpret_cast<const float4*>(ka), kn = *reinterpret_cast<const float4*>(ka + C);
            float4 v;
            if (part == 0) v = xi;
            else if (part == 1) v = kd;
            else {
                const float xv[4] = {xi.x, xi.y, xi.z, xi.w}, x1[4] = {xi1.x, xi1.y, xi1.z, xi1.w};
                const float kv[4] = {kd.x, kd.y, kd.z, kd.w}, k1[4] = {kn.x, kn.y, kn.z, kn.w};
                float o[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float six = 2.0f * (3.0f * (x1[t] - xv[t]));
                    o[t] = part == 2 ? (six * 1.0f - 4.0f * kv[t] - 2.0f * k1[t]) * 1.0f : (-six * 1.0f + 3.0f * (kv[t] + k1[t])) * 1.0f;
                }
                v = make_float4(o[0], o[1], o[2], o[3]);
            }
            *reinterpret_cast<float4*>(out + ((long long)(b0 + s2) * (L - 1) + pp) * C4 + part * C + 4 * q) = v;
        }
    } else {
        const int per_smp = (L - 1) * C4;
        for (int e = lane; e < nsmp * per_smp; e += CUBIC_NT) {
            const int s2 = e / per_smp, r = e - s2 * per_smp;
            const int pp = r / C4, q4 = r - pp * C4;
            const int part = q4 / C, c2 = q4 - part * C;
            const float xi = xs[(size_t)s2 * sstride + pp * C + c2], xi1 = xs[(size_t)s2 * sstride + (pp + 1) * C + c2];
            const float kd = nbs[(size_t)s2 * sstride + pp * C + c2], kn = nbs[(size_t)s2 * sstride + (pp + 1) * C + c2];
            const float six = 2.0f * (3.0f * (xi1 - xi));
            float v = xi;
            if (part == 1) v = kd;
            else if (part == 2) v = (six * 1.0f - 4.0f * kd - 2.0f * kn) * 1.0f;
            else if (part == 3) v = (-six * 1.0f + 3.0f * (kd + kn)) * 1.0f;
            out[((long long)(b0 + s2) * (L - 1) + pp) * C4 + q4] = v;
        }
    }
}

// swept diagonal of the natural-spline system on the unit grid: d_0 = 2, d_i = diag_i - (1/d_{i-1}) * 1
// (one wave; the interior iteration d <- 4 - 1/d reaches its fp32 fixed point after a handful of steps, after which every
// lane fills its share of the table: the serial part is ~10 steps, not L)
__global__ void ncde_cubic_diag_kernel(int L, float* diag_swept) {
    if (blockIdx.x != 0) return;
    auto next = [](float d_prev, int i, int L_) {
        float diag = (i < L_ - 1 ? 1.0f : 0.0f) + (i > 0 ? 1.0f : 0.0f);
        diag *= 2.0f;
        if (i == 0) return diag;
        const float w = 1.0f / d_prev;
        return diag - w * 1.0f;
    };
    float d_prev = 0.0f;
    int i = 0;
    for (; i < L; ++i) {                       // every lane runs the same serial prefix (uniform)
        const float d = next(d_prev, i, L);
        if (threadIdx.x == 0) diag_swept[i] = d;
        const bool fixed = i > 0 && i < L - 2 && d == d_prev;
        d_prev = d;
        if (fixed) { ++i; break; }
    }
    // interior entries i .. L-2 all equal the fixed point; the last one sees diag = 2
    for (int k = i + (int)threadIdx.x; k < L - 1; k += (int)blockDim.x) diag_swept[k] = d_prev;
    if (threadIdx.x == 0 && i < L) diag_swept[L - 1] = next(d_prev, L - 1, L);
}


}  // namespace

extern "C" {

int64_t ncde_prepare_workspace_bytes(int kind, int B, int L, int C) {
    if (B < 1 || L < 2 || C < 1) return NCDE_ERR_INVALID;
    if (kind == NCDE_INTERP_CUBIC) return (int64_t)sizeof(float) * (2 * (int64_t)B * L * C + L) + 256;
    return 256;
}

int ncde_prepare_linear_grid(const float* x, const float* t, int B, int L, int C, int rectilinear_time_index, float* out, void* stream) {
    if (!x || !out || B < 1 || L < 2 || C < 1 || rectilinear_time_index >= C) return NCDE_ERR_INVALID;
    const float* tg = rectilinear_time_index < 0 ? t : nullptr;      // the rectilinear preparation leaves no gap the grid could enter
    const long long n = (long long)B * C;
    const int rect = rectilinear_time_index < 0 ? -1 : rectilinear_time_index;
    const int T = rect >= 0 ? 2 * L - 1 : L;
    (void)T;
    const size_t lds = sizeof(float) * ((size_t)L * C * (rect >= 0 ? 2 : 3) + 64 * (size_t)C);
    if (lds <= 64 * 1024 && C <= 256) {   // the series fits LDS: staged, coalesced, scan-parallel variant
        if (ncde_lds_optin((const void*)ncde_linear_coeffs_lds_kernel, lds) != hipSuccess)
            return NCDE_ERR_HIP;
        hipLaunchKernelGGL(ncde_linear_coeffs_lds_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, x, B, L, C, rect, out, tg);
    } else {
        hipLaunchKernelGGL(ncde_linear_coeffs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, B, L, C, rect, out, tg);
    }
    return hipGetLastError() == hipSuccess ? NCDE_OK : NCDE_ERR_HIP;
}

int ncde_prepare_linear(const float* x, int B, int L, int C, int rectilinear_time_index, float* out, void* stream) {
    return ncde_prepare_linear_grid(x, nullptr, B, L, C, rectilinear_time_index, out, stream);
}

int ncde_prepare_cubic_grid(const float* x, const float* t, int B, int L, int C, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !out || !workspace || B < 1 || L < 2 || C < 1) return NCDE_ERR_INVALID;
    if ((int64_t)workspace_bytes < ncde_prepare_workspace_bytes(NCDE_INTERP_CUBIC, B, L, C)) return NCDE_ERR_WORKSPACE;
    float* ws = (float*)workspace;
    float* ws_d = ws + (size_t)B * L * C;
    float* diag = ws_d + (size_t)B * L * C;
    hipLaunchKernelGGL(ncde_cubic_diag_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, L, diag);
    const long long n = (long long)B * C;
    // LDS-staged variant: NSMP whole samples per wave (lane <-> (sample, channel)), as many as fit 64 lanes and the LDS budget
    if (C <= 64 && !t) {      // (constant-coefficient system: the default grid only)
        const int LC = L * C;
        int sstride = LC + ((C - (LC % 32)) % 32 + 32) % 32;      // stride == C (mod 32): the lanes of one time step fall on distinct banks
        sstride = (sstride + 3) & ~3;
        int nsmp = 64 / C;
        auto lds_of = [&](int ns) { return sizeof(float) * ((size_t)2 * ns * sstride + 2 * ((L + 3) & ~3)); };
        // the recurrences are latency chains: prefer several resident workgroups per CU (<= 28 KB each: measured best at cfg4 size) over full waves
        const char* ev = ncde_dev_env("NCDE_CUBIC_LDS_KB");
        const size_t budget = (size_t)(ev && atoi(ev) > 0 ? atoi(ev) : 28) * 1024;
        while (nsmp > 1 && lds_of(nsmp) > budget) --nsmp;
        if (lds_of(nsmp) <= (size_t)150 * 1024) {
            const size_t lds = lds_of(nsmp);
            if (ncde_lds_optin((const void*)ncde_cubic_coeffs_lds_kernel, lds) != hipSuccess)
                return NCDE_ERR_HIP;
            hipLaunchKernelGGL(ncde_cubic_coeffs_lds_kernel, dim3((unsigned)((B + nsmp - 1) / nsmp)), dim3(CUBIC_NT), lds, (hipStream_t)stream, x, B, L, C,
                               nsmp, sstride, out, ws, diag, ws_d);
            return hipGetLastError() == hipSuccess ? NCDE_OK : NCDE_ERR_HIP;
        }
    }
    hipLaunchKernelGGL(ncde_cubic_coeffs_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, x, B, L, C, out,
                       ws, diag, ws_d, t);
    return hipGetLastError() == hipSuccess ? NCDE_OK : NCDE_ERR_HIP;
}

int ncde_prepare_cubic(const float* x, int B, int L, int C, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    return ncde_prepare_cubic_grid(x, nullptr, B, L, C, out, workspace, workspace_bytes, stream);
}

}  // extern "C"
