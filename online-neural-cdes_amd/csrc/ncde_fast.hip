// Shape-specialised, register-resident Neural-CDE kernels for gfx950 (the hot path of BASELINE cfg2/3/4).
//
// Workgroup = one tile of 16 samples, NW waves (one per SIMD).  v_mfma_f32_16x16x4_f32 is used in its
// "transposed" role: A = weights (16 output units x 4 k), B = activations (4 k x 16 samples), so a lane
// (s = lane&15, g = lane>>4) holds, for ONE sample s, output units chosen by how the weight rows are
// permuted into tiles.  With the permutation  tile t, D-row (g, r)  <->  unit 4*(4t+r)+g  the D registers
// of one layer ARE the B operands of the next layer (k-step 4t+r, k-sub g): the whole MLP chain runs
// register-to-register with no cross-lane movement.  The output layer uses rows (g, r) <-> (h = 4hb+g,
// c = 4cq+r) so the channel contraction sum_c tanh(.)[h,c] dX[c] is a per-lane FMA chain over r and cq.
// All weights live in VGPR/AGPRs for the whole solve (W0, W1 replicated per wave, Wo/bo split by h-block
// across the NW waves); the only per-stage traffic is the H x 16 stage state exchanged through LDS.
// dX/dt is formed on chip from ONE new coefficient row per step, prefetched a step ahead.
//
// Reference semantics: see ncde_generic.hip (same stage tables, same knot-index rule).
#include "ncde_fast.h"
#include "ncde_fast4.h"
#include "ncde_fast64.h"
#include "ncde_fast_nl.h"
#include "ncde_fast_c.h"
#include "ncde_fast_fwd3.h"
#include "ncde_fast_plan.h"
// HP = 2 (the default adjoint): hidden-layer dW/db of the previous stage behind barrier A (dL/dpre images double-buffered) / all five
// dWo blocks behind barrier A -- both shorten what the gradient waves do before barrier A, where the chain waves wait for them
#ifndef NCDE_F2_DW_LATE
#define NCDE_F2_DW_LATE 0
#endif
// HP = 2: dL/dx_L = Wo^T dP as split-bf16 (60 x 16-cycle MFMAs per stage instead of 80 x 32-cycle fp32 ones); the hi and mid pieces of
// Wo^T take the 80 registers the fp32 operand took, the lo pieces live in the LDS region the chain waves' Wo lo pieces used to occupy
#ifndef NCDE_F2_DXL_BF3
#define NCDE_F2_DXL_BF3 1
#endif
#ifndef NCDE_F2_DWO_EARLY
#define NCDE_F2_DWO_EARLY 0
#endif
#ifndef NCDE_H2_DW_LATE
#define NCDE_H2_DW_LATE 1
#endif
#ifndef NCDE_H2_DWO_EARLY
#define NCDE_H2_DWO_EARLY 0
#endif

#include <cstring>
#include <type_traits>

#include "ncde_common.h"
#include "ncde_bf3.h"
#include "ncde_fastdefs.h"
#include "ncde_host.h"

namespace {

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int H, int HH, int C, int NW, int INTERP, int METHOD, int PROF = 0>
__global__ __launch_bounds__(64 * NW, 1) void ncde_fwd_fast(KArgs a) {
    // PROF = 1: s_memtime phase counters (debug builds of the dispatcher only; see tools/profile_phases.py)
    unsigned long long prof[4] = {0, 0, 0, 0}, tlast = 0;
#define NCDE_TICK(k)                                                \
    if constexpr (PROF != 0) {                                      \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        prof[k] += now_ - tlast;                                    \
        tlast = now_;                                               \
    }
    constexpr int CP = (C + 3) & ~3, CQ = CP / 4, HB = H / 4, HT = HH / 16, KH = HH / 4, NB = HB / NW;
    constexpr int S = kStages<METHOD>;
    constexpr int NT = 64 * NW;
    constexpr int DXW = INTERP == NCDE_INTERP_LINEAR ? CP : 3 * CP;  // floats per sample per piece
    constexpr int EPT = (16 * DXW + NT - 1) / NT;                      // staged elements per thread
    static_assert(H % (4 * NW) == 0 && HH % 16 == 0, "shape not tileable");
    __shared__ __attribute__((aligned(16))) float zx[2][H * 16];
    __shared__ __attribute__((aligned(16))) float dxs[3][16 * DXW];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = lane & 15, g = lane >> 4;
    const int b0 = blockIdx.x * NCDE_TILE;
    const int bs = b0 + s;
    const bool valid = bs < a.B;

    // ---- weights -> registers ---------------------------------------------------------------------
    float w0[HT][HB], w1[HT][KH], wo[NB][CQ][KH];
    f32x4 bias0[HT], bias1[HT], biaso[NB][CQ];
    const bool has_inner = a.n_layers > 1;
#pragma unroll
    for (int t = 0; t < HT; ++t) {
        const int unitA = 4 * (4 * t + (s & 3)) + (s >> 2);  // A row i = s  <->  D row (i>>2, i&3)
#pragma unroll
        for (int ks = 0; ks < HB; ++ks) w0[t][ks] = a.W[0][unitA * H + 4 * ks + g];
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) w1[t][ks] = has_inner ? a.W[1][unitA * HH + 4 * ks + g] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int unitD = 4 * (4 * t + r) + g;
            bias0[t][r] = a.b[0][unitD];
            bias1[t][r] = has_inner ? a.b[1][unitD] : 0.0f;
        }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int hb = wave * NB + nb;
#pragma unroll
        for (int cq = 0; cq < CQ; ++cq) {
            const int hA = 4 * hb + (s >> 2), cA = 4 * cq + (s & 3);
#pragma unroll
            for (int ks = 0; ks < KH; ++ks) wo[nb][cq][ks] = cA < C ? NCDE_TANH_PRESCALE * a.Wo[(hA * C + cA) * HH + 4 * ks + g] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 4 * cq + r;
                biaso[nb][cq][r] = c < C ? NCDE_TANH_PRESCALE * a.bo[(4 * hb + g) * C + c] : 0.0f;
            }
        }
    }

    // ---- control-path staging: thread-owned elements of the [16][DXW] per-piece image -------------
    // linear: dX = row[p+1] - row[p] (one new row per step); cubic: b | 2c | 3d of piece p.
    const float* eptr[EPT];
    float eprev[EPT], enext[EPT];
    bool eok[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int e = tid + q * NT;
        const int es = e / DXW, ec = e - es * DXW;
        const int part = ec / CP, c = ec - part * CP;  // cubic: part 0..2 = b, 2c, 3d
        eok[q] = e < 16 * DXW && c < a.Cc && (b0 + es) < a.B;      // a.Cc: channels of the coefficient tensor (= C unless zero-padded)
        const long long base = (long long)(eok[q] ? b0 + es : 0) * a.cs_b;
        eptr[q] = a.coeffs + base + (INTERP == NCDE_INTERP_LINEAR ? c : (part + 1) * a.Cc + c);
        eprev[q] = 0.0f;
        enext[q] = 0.0f;
    }
    auto stage_load = [&](int piece) {  // global -> registers (piece must be < n_pieces)
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int row = INTERP == NCDE_INTERP_LINEAR ? piece + 1 : piece;
            enext[q] = eok[q] ? eptr[q][(long long)row * a.cs_t] : 0.0f;
        }
    };
    auto stage_store = [&](int piece) {  // registers -> LDS ring slot piece % 3
        float* dst = dxs[piece % 3];
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = tid + q * NT;
            if (e < 16 * DXW) dst[e] = INTERP == NCDE_INTERP_LINEAR ? enext[q] - eprev[q] : enext[q];
            eprev[q] = enext[q];
        }
    };
    if (INTERP == NCDE_INTERP_LINEAR) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) eprev[q] = eok[q] ? eptr[q][0] : 0.0f;  // row 0
    }
    stage_load(0);
    stage_store(0);

    // ---- state ----------------------------------------------------------------------------------------
    float y0[NB], k1[NB], k2[NB], zreg[HB];
#pragma unroll
    for (int ks = 0; ks < HB; ++ks) zreg[ks] = (valid && 4 * ks + g < a.Hr) ? a.z0[(long long)bs * a.Hr + 4 * ks + g] : 0.0f;      // a.Hr: row width of z0 / out (= H unless zero-padded)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int u = 4 * (wave * NB + nb) + g;
        y0[nb] = (valid && u < a.Hr) ? a.z0[(long long)bs * a.Hr + u] : 0.0f;
        k1[nb] = 0.0f;
        k2[nb] = 0.0f;
        if (valid && u < a.Hr) a.out[((long long)bs * a.n_out) * a.Hr + u] = y0[nb];
    }
    __syncthreads();

    const int n_inner = a.n_layers - 1;
    int zpar = 0;
    if constexpr (PROF != 0) tlast = __builtin_readcyclecounter();
    for (int n = 0; n < a.T - 1; ++n) {
        if (n + 1 < a.n_pieces) stage_load(n + 1);  // prefetch next piece; consumed at the end of the step
#pragma unroll
        for (int j = 0; j < S; ++j) {
            const float t = (float)n + stage_offset(METHOD, j);
            const int idx = piece_index(t, a.n_pieces);
            const float frac = t - (float)idx;
            const float* dxp = dxs[idx % 3] + s * DXW;
            if (a.stages != nullptr && wave == ((n * S + j) % NW) && valid) {  // record the stage input (exact backward)
                float* rec = a.stages + ((long long)(n * S + j) * a.B + bs) * a.Hr;
#pragma unroll
                for (int ks = 0; ks < HB; ++ks)
                    if (4 * ks + g < a.Hr) rec[4 * ks + g] = zreg[ks];
            }
            // ---- hidden layers, register to register -------------------------------------------------
            f32x4 acc[HT];
            float hB[KH];
#pragma unroll
            for (int tt = 0; tt < HT; ++tt) acc[tt] = bias0[tt];
#pragma unroll
            for (int ks = 0; ks < HB; ++ks)
#pragma unroll
                for (int tt = 0; tt < HT; ++tt) acc[tt] = mfma16(w0[tt][ks], zreg[ks], acc[tt]);
#pragma unroll
            for (int tt = 0; tt < HT; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) hB[4 * tt + r] = relu_dev(acc[tt][r]);
            for (int rep = 0; rep < n_inner; ++rep) {
#pragma unroll
                for (int tt = 0; tt < HT; ++tt) acc[tt] = bias1[tt];
#pragma unroll
                for (int ks = 0; ks < KH; ++ks)
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) acc[tt] = mfma16(w1[tt][ks], hB[ks], acc[tt]);
#pragma unroll
                for (int tt = 0; tt < HT; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) hB[4 * tt + r] = relu_dev(acc[tt][r]);
            }
            NCDE_TICK(0)
            // ---- output layer tiles owned by this wave: tanh + channel contraction -------------------
            // software pipeline over cq: the MFMA chains of tile group cq run while the VALU finishes
            // tanh + contraction of group cq-1 (one wave per SIMD: overlap exists only where VALU
            // instructions sit between MFMAs in program order, hence the explicit interleave hints)
            float kout[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) kout[nb] = 0.0f;
            f32x4 oprev[NB];
#pragma unroll
            for (int cq = 0; cq <= CQ; ++cq) {
                f32x4 o[NB];
                if (cq < CQ) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) o[nb] = biaso[nb][cq];
#pragma unroll
                    for (int ks = 0; ks < KH; ++ks)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) o[nb] = mfma16(wo[nb][cq][ks], hB[ks], o[nb]);
                }
                if (cq > 0) {
                    const int cp = cq - 1;
                    f32x4 dx;
                    if constexpr (INTERP == NCDE_INTERP_LINEAR) {
                        dx = *reinterpret_cast<const f32x4*>(dxp + 4 * cp);
                    } else {
                        const f32x4 cb = *reinterpret_cast<const f32x4*>(dxp + 4 * cp);
                        const f32x4 cc = *reinterpret_cast<const f32x4*>(dxp + CP + 4 * cp);
                        const f32x4 cd = *reinterpret_cast<const f32x4*>(dxp + 2 * CP + 4 * cp);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float inner = cc[r] + cd[r] * frac;
                            dx[r] = cb[r] + inner * frac;
                        }
                    }
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) kout[nb] = fmaf(tanh_prescaled(oprev[nb][r]), dx[r], kout[nb]);
                }
                if (cq > 0 && cq < CQ) {
                    // per MFMA of group cq: 3 VALU (+ transcendental) slots of group cq-1's epilogue
#pragma unroll
                    for (int i = 0; i < KH * NB; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // VALU
                    }
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) oprev[nb] = o[nb];
            }
            NCDE_TICK(1)
            // ---- Butcher bookkeeping for the owned state entries, then exchange the stage input ------
            float ys[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) ys[nb] = Combine<METHOD>::apply(j, kout[nb], y0[nb], k1[nb], k2[nb]);
            if (j == S - 1) {
                if (valid && (a.output == NCDE_OUT_KNOTS || n == a.T - 2)) {
                    const int row = a.output == NCDE_OUT_KNOTS ? n + 1 : 1;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        if (4 * (wave * NB + nb) + g < a.Hr) a.out[((long long)bs * a.n_out + row) * a.Hr + 4 * (wave * NB + nb) + g] = ys[nb];
                }
                if (n + 1 < a.n_pieces) stage_store(n + 1);
            }
            if constexpr (NW == 1) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) zreg[nb] = ys[nb];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            } else {
                float* zw = zx[zpar];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) zw[(4 * (wave * NB + nb) + g) * 16 + s] = ys[nb];
                __syncthreads();
#pragma unroll
                for (int ks = 0; ks < HB; ++ks) zreg[ks] = zw[(4 * ks + g) * 16 + s];
                zpar ^= 1;
            }
            NCDE_TICK(2)
        }
    }
    if constexpr (PROF != 0) {
        if (lane == 0) {
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.gpart) + ((long long)blockIdx.x * NW + wave) * 4;
            dst[0] = prof[0]; dst[1] = prof[1]; dst[2] = prof[2]; dst[3] = prof[3];
        }
    }
#undef NCDE_TICK
}

// ------------------------------------------------------------------------------------------------
// forward, split-bf16 variant: fp32-equivalent GEMMs on the bf16 matrix cores
// ------------------------------------------------------------------------------------------------
// (the split-bf16 arithmetic itself lives in ncde_bf3.h, shared with the batch-tiled family)
// NLT = number of layers known at compile time (0 = runtime): with the layer loop unrolled the whole stage is ONE basic
// block, so the scheduler can issue the hi-piece MFMAs of layer l+1 under the mid / lo split of layer l.
// PLAN = 1 (round 4): the general time axis (a.plan, ncde_timeplan.hip) -- per-step dt, per-stage (piece, t - knot, knot spacing),
// output rows picked / interpolated between the step's end points -- instead of the default integer grid with step 1.  The control
// path is then staged as dX/dt PER STAGE of the next step (evaluated from the plan's stage descriptors by all threads while the
// current step computes) instead of one new coefficient row per step: a plan may revisit or skip pieces.
template <int H, int HH, int C, int NW, int INTERP, int METHOD, int PROF = 0, int NLT = 0, int HP = 0, int PLAN = 0>
__global__ __launch_bounds__(64 * NW, (NW + 3) / 4) void ncde_fwd_fast_bf3(KArgs a) {
    unsigned long long prof[4] = {0, 0, 0, 0}, tlast = 0;
#define NCDE_TICK(k)                                                \
    if constexpr (PROF != 0) {                                      \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        prof[k] += now_ - tlast;                                    \
        tlast = now_;                                               \
    }
    constexpr int CP = (C + 3) & ~3, CQ = CP / 4, HB = H / 4, HT = HH / 16, NB = HB / NW;
    constexpr int KC0 = H / 32, KC = HH / 32;  // K chunks of layer 0 / of the HH-wide layers
    constexpr int S = kStages<METHOD>;
    constexpr int NT = 64 * NW;
    constexpr int DXW = INTERP == NCDE_INTERP_LINEAR ? CP : 3 * CP;
    constexpr int EPT = (16 * DXW + NT - 1) / NT;
    static_assert(H % (4 * NW) == 0 && HH % 32 == 0 && H % 32 == 0, "shape not tileable");
    __shared__ __attribute__((aligned(16))) float zx[2][H * 16];
    __shared__ __attribute__((aligned(16))) float dxs[PLAN ? 1 : 3][PLAN ? 4 : 16 * DXW];
    __shared__ __attribute__((aligned(16))) float dxq[PLAN ? 2 : 1][PLAN ? S * 16 * CP : 4];      // PLAN: dX/dt of every stage of a step, by step parity
    __shared__ int fault_s;
    // split-fp16 instances speculate on the fp16 range and report a fault per sample tile; the split-bf16 instance, launched behind
    // them with only_faulted set, re-executes exactly those tiles (ncde_bf3.h)
    if constexpr (HP == 0) {
        if (a.only_faulted && a.fault[blockIdx.x] == 0) return;
    }
    if constexpr (PLAN != 0) {
        if (a.plan == nullptr || !plan_header_ok(a, S)) return;      // (uniform: before the first barrier)
    }
    float mx = 0.0f;              // largest operand magnitude the split-fp16 GEMMs have seen (ncde_bf3.h)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = lane & 15, g = lane >> 4;
    const int b0 = blockIdx.x * NCDE_TILE;
    const int bs = b0 + s;
    const bool valid = bs < a.B;

    // ---- weights -> split bf16 A operands in registers ------------------------------------------------
    typedef SplitOps<HP> SO;
    typedef typename SO::T SpT;
    SpT w0[HT][KC0], w1[HT][KC], wo[NB][CQ][KC];
    f32x4 bias0[HT], bias1[HT], biaso[NB][CQ];
    const bool has_inner = a.n_layers > 1;
#pragma unroll
    for (int t = 0; t < HT; ++t) {
        // A row i = s <-> D row (g' = i>>2, r' = i&3) <-> unit 32*(t>>1) + 8g' + 4*(t&1) + r'
        const int unitA = 32 * (t >> 1) + 8 * (s >> 2) + 4 * (t & 1) + (s & 3);
        float tmp[8];
#pragma unroll
        for (int c = 0; c < KC0; ++c) {
#pragma unroll
            for (int j = 0; j < 8; ++j) tmp[j] = a.W[0][unitA * H + 32 * c + 8 * g + j];
            w0[t][c] = SO::split(tmp, mx);
        }
#pragma unroll
        for (int c = 0; c < KC; ++c) {
#pragma unroll
            for (int j = 0; j < 8; ++j) tmp[j] = has_inner ? a.W[1][unitA * HH + 32 * c + 8 * g + j] : 0.0f;
            w1[t][c] = SO::split(tmp, mx);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int unitD = 32 * (t >> 1) + 8 * g + 4 * (t & 1) + r;
            bias0[t][r] = a.b[0][unitD];
            bias1[t][r] = has_inner ? a.b[1][unitD] : 0.0f;
        }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int hb = wave * NB + nb;
#pragma unroll
        for (int cq = 0; cq < CQ; ++cq) {
            const int hA = 4 * hb + (s >> 2), cA = 4 * cq + (s & 3);
            float tmp[8];
#pragma unroll
            for (int c = 0; c < KC; ++c) {
#pragma unroll
                for (int j = 0; j < 8; ++j) tmp[j] = cA < C ? NCDE_TANH_PRESCALE * a.Wo[(hA * C + cA) * HH + 32 * c + 8 * g + j] : 0.0f;
                wo[nb][cq][c] = SO::split(tmp, mx);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cc = 4 * cq + r;
                biaso[nb][cq][r] = cc < C ? NCDE_TANH_PRESCALE * a.bo[(4 * hb + g) * C + cc] : 0.0f;
            }
        }
    }

    // ---- control-path staging (identical to ncde_fwd_fast) ------------------------------------------------
    const float* eptr[EPT];
    float eprev[EPT], enext[EPT];
    bool eok[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int e = tid + q * NT;
        const int es = e / DXW, ec = e - es * DXW;
        const int part = ec / CP, c = ec - part * CP;
        eok[q] = e < 16 * DXW && c < a.Cc && (b0 + es) < a.B;      // a.Cc: channels of the coefficient tensor (= C unless zero-padded)
        const long long base = (long long)(eok[q] ? b0 + es : 0) * a.cs_b;
        eptr[q] = a.coeffs + base + (INTERP == NCDE_INTERP_LINEAR ? c : (part + 1) * a.Cc + c);
        eprev[q] = 0.0f;
        enext[q] = 0.0f;
    }
    auto stage_load = [&](int piece) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int row = INTERP == NCDE_INTERP_LINEAR ? piece + 1 : piece;
            enext[q] = eok[q] ? eptr[q][(long long)row * a.cs_t] : 0.0f;
        }
    };
    auto stage_store = [&](int piece) {
        float* dst = dxs[piece % 3];
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = tid + q * NT;
            if (e < 16 * DXW) dst[e] = INTERP == NCDE_INTERP_LINEAR ? enext[q] - eprev[q] : enext[q];
            eprev[q] = enext[q];
        }
    };
    // PLAN: element e = (stage j, sample es, channel c) of the [S][16][CP] image of one step
    constexpr int EPQ = PLAN ? (S * 16 * CP + NT - 1) / NT : 1;
    float qn[EPQ];
    auto plan_load = [&](const int* pstep) {
#pragma unroll
        for (int q = 0; q < EPQ; ++q) {
            const int e = tid + q * NT;
            const int j = e / (16 * CP), rem = e - j * (16 * CP), es = rem / CP, c = rem - es * CP;
            float v = 0.0f;
            if (e < S * 16 * CP && c < a.Cc && b0 + es < a.B) {
                const StageDesc sd = plan_stage(pstep, j);
                const float* p = a.coeffs + (long long)(b0 + es) * a.cs_b + (long long)sd.idx * a.cs_t;
                if constexpr (INTERP == NCDE_INTERP_LINEAR) {
                    v = p[a.cs_t + c] - p[c];
                    if (sd.kdt != 1.0f) v = v / sd.kdt;      // user knot grid (interpolation_linear.py:231-234); 1 on the default grid
                } else {
                    const float bb = p[a.Cc + c], cc = p[2 * a.Cc + c], dd = p[3 * a.Cc + c];
                    const float inner = cc + dd * sd.frac;
                    v = bb + inner * sd.frac;
                }
            }
            qn[q] = v;
        }
    };
    auto plan_store = [&](int buf) {
#pragma unroll
        for (int q = 0; q < EPQ; ++q) {
            const int e = tid + q * NT;
            if (e < S * 16 * CP) dxq[buf][e] = qn[q];
        }
    };
    const int pw_ = plan_step_words(S);
    const int* pfwd = PLAN ? a.plan + plan_off_fwd() : nullptr;
    const int* pout = PLAN ? a.plan + plan_off_out(S, a.n_steps_fwd) : nullptr;
    const int n_steps = PLAN ? a.n_steps_fwd : a.T - 1;
    if constexpr (PLAN != 0) {
        plan_load(pfwd);
        plan_store(0);
    } else {
        if (INTERP == NCDE_INTERP_LINEAR) {
#pragma unroll
            for (int q = 0; q < EPT; ++q) eprev[q] = eok[q] ? eptr[q][0] : 0.0f;
        }
        stage_load(0);
        stage_store(0);
    }

    // ---- state: lane (s, g) keeps z[s][32c + 8g + j] as the layer-0 B operand ----------------------------
    float y0[NB], k1[NB], k2[NB], zreg[KC0][8];
#pragma unroll
    for (int c = 0; c < KC0; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) zreg[c][j] = (valid && 32 * c + 8 * g + j < a.Hr) ? a.z0[(long long)bs * a.Hr + 32 * c + 8 * g + j] : 0.0f;      // a.Hr: row width of z0 / out (= H unless zero-padded)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int u = 4 * (wave * NB + nb) + g;
        y0[nb] = (valid && u < a.Hr) ? a.z0[(long long)bs * a.Hr + u] : 0.0f;
        k1[nb] = 0.0f;
        k2[nb] = 0.0f;
        if (valid && u < a.Hr) a.out[((long long)bs * a.n_out) * a.Hr + u] = y0[nb];
    }
    __syncthreads();

    const int n_inner = a.n_layers - 1;
    int zpar = 0;
    if constexpr (PROF != 0) tlast = __builtin_readcyclecounter();
    for (int n = 0; n < n_steps; ++n) {
        const int* pstep = PLAN ? pfwd + n * pw_ : nullptr;
        const float dt = PLAN ? __int_as_float(pstep[0]) : 1.0f;
        if constexpr (PLAN != 0) {
            if (n + 1 < n_steps) plan_load(pstep + pw_);
        } else {
            if (n + 1 < a.n_pieces) stage_load(n + 1);
        }
#pragma unroll
        for (int j = 0; j < S; ++j) {
            const float t = (float)n + stage_offset(METHOD, j);
            const int idx = PLAN ? 0 : piece_index(t, a.n_pieces);
            const float frac = t - (float)idx;
            const float* dxp = PLAN ? dxq[n & 1] + (j * 16 + s) * CP : dxs[idx % 3] + s * DXW;
            if (a.stages != nullptr && wave == ((n * S + j) % NW) && valid) {  // record the stage input (exact backward)
                float* rec = a.stages + ((long long)(n * S + j) * a.B + bs) * a.Hr;
                if (a.Hr == H) {
#pragma unroll
                    for (int c = 0; c < KC0; ++c) {
                        *reinterpret_cast<f32x4*>(rec + 32 * c + 8 * g) = (f32x4){zreg[c][0], zreg[c][1], zreg[c][2], zreg[c][3]};
                        *reinterpret_cast<f32x4*>(rec + 32 * c + 8 * g + 4) = (f32x4){zreg[c][4], zreg[c][5], zreg[c][6], zreg[c][7]};
                    }
                } else {      // zero-padded problem: rows of the caller's record are a.Hr wide
#pragma unroll
                    for (int c = 0; c < KC0; ++c)
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj)
                            if (32 * c + 8 * g + jj < a.Hr) rec[32 * c + 8 * g + jj] = zreg[c][jj];
                }
            }
            // ---- hidden layers ---------------------------------------------------------------------------
            typename SO::Acc acc[HT];
            float hv[KC][8];
            SpT xb[KC];
            auto activate = [&]() {
#pragma unroll
                for (int tt = 0; tt < HT; ++tt) {
                    const f32x4 pre = SO::finish(acc[tt]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) hv[tt >> 1][4 * (tt & 1) + r] = relu_bits(pre[r]);
                }
#pragma unroll
                for (int c = 0; c < KC; ++c) xb[c] = SO::split(hv[c], mx);
            };
            {
                SpT zb[KC0];
#pragma unroll
                for (int c = 0; c < KC0; ++c) zb[c] = SO::split(zreg[c], mx);
#pragma unroll
                for (int tt = 0; tt < HT; ++tt) acc[tt] = SO::init(bias0[tt]);
#pragma unroll
                for (int c = 0; c < KC0; ++c)
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) SO::mac(w0[tt][c], zb[c], acc[tt]);
            }
            activate();
            auto inner_layer = [&]() {
#pragma unroll
                for (int tt = 0; tt < HT; ++tt) acc[tt] = SO::init(bias1[tt]);
#pragma unroll
                for (int c = 0; c < KC; ++c)
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) SO::mac(w1[tt][c], xb[c], acc[tt]);
                activate();
            };
            if constexpr (NLT > 0) {
#pragma unroll
                for (int rep = 0; rep < NLT - 1; ++rep) inner_layer();
            } else {
                for (int rep = 0; rep < n_inner; ++rep) inner_layer();
            }
            NCDE_TICK(0)
            // ---- output layer tiles owned by this wave: tanh + channel contraction -----------------------
            float kout[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) kout[nb] = 0.0f;
#pragma unroll
            for (int cq = 0; cq < CQ; ++cq) {
                typename SO::Acc oa[NB];
                f32x4 o[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) oa[nb] = SO::init(biaso[nb][cq]);
#pragma unroll
                for (int c = 0; c < KC; ++c)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) SO::mac(wo[nb][cq][c], xb[c], oa[nb]);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) o[nb] = SO::finish(oa[nb]);
                f32x4 dx;
                if constexpr (INTERP == NCDE_INTERP_LINEAR || PLAN != 0) {      // (PLAN: the staged values ARE dX/dt of this stage)
                    dx = *reinterpret_cast<const f32x4*>(dxp + 4 * cq);
                } else {
                    const f32x4 cb = *reinterpret_cast<const f32x4*>(dxp + 4 * cq);
                    const f32x4 cc = *reinterpret_cast<const f32x4*>(dxp + CP + 4 * cq);
                    const f32x4 cd = *reinterpret_cast<const f32x4*>(dxp + 2 * CP + 4 * cq);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float inner = cc[r] + cd[r] * frac;
                        dx[r] = cb[r] + inner * frac;
                    }
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) kout[nb] = fmaf(tanh_prescaled(o[nb][r]), dx[r], kout[nb]);
            }
            NCDE_TICK(1)
            float ys[NB], yprev[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                yprev[nb] = y0[nb];
                if constexpr (PLAN != 0) {
                    bool last;
                    ys[nb] = StageCombine::apply(METHOD, j, kout[nb], dt, y0[nb], k1[nb], k2[nb], last);
                } else {
                    ys[nb] = Combine<METHOD>::apply(j, kout[nb], y0[nb], k1[nb], k2[nb]);
                }
            }
            if (j == S - 1) {
                if constexpr (PLAN != 0) {      // output pick / interpolation between the step's end points (solvers.py:103-117, 166-172)
                    if (valid) {
                        const int q0 = pstep[1], q1 = q0 + pstep[2];
                        for (int r = q0; r < q1; ++r) {
                            const int kind = pout[2 * r];
                            const float slope = __int_as_float(pout[2 * r + 1]);
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
                                if (4 * (wave * NB + nb) + g < a.Hr)
                                    a.out[((long long)bs * a.n_out + r) * a.Hr + 4 * (wave * NB + nb) + g] =
                                        kind == 1 ? ys[nb] : (kind == 0 ? yprev[nb] : yprev[nb] + slope * (ys[nb] - yprev[nb]));
                        }
                    }
                    if (n + 1 < n_steps) plan_store((n + 1) & 1);
                } else {
                    if (valid && (a.output == NCDE_OUT_KNOTS || n == a.T - 2)) {
                        const int row = a.output == NCDE_OUT_KNOTS ? n + 1 : 1;
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            if (4 * (wave * NB + nb) + g < a.Hr) a.out[((long long)bs * a.n_out + row) * a.Hr + 4 * (wave * NB + nb) + g] = ys[nb];
                    }
                    if (n + 1 < a.n_pieces) stage_store(n + 1);
                }
            }
            {
                float* zw = zx[zpar];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) zw[(4 * (wave * NB + nb) + g) * 16 + s] = ys[nb];
                __syncthreads();
#pragma unroll
                for (int c = 0; c < KC0; ++c)
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) zreg[c][jj] = zw[(32 * c + 8 * g + jj) * 16 + s];
                zpar ^= 1;
            }
            NCDE_TICK(2)
        }
    }
    if constexpr (PROF != 0) {
        if (lane == 0) {
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.gpart) + ((long long)blockIdx.x * NW + wave) * 4;
            dst[0] = prof[0]; dst[1] = prof[1]; dst[2] = prof[2]; dst[3] = prof[3];
        }
    }
    if constexpr (HP != 0) {
        if (a.fault != nullptr) {
            if (tid == 0) fault_s = 0;
            __syncthreads();
            if (__builtin_amdgcn_ballot_w64(h2_range_fault(mx)) != 0 && lane == 0) fault_s = 1;
            __syncthreads();
            if (tid == 0) a.fault[blockIdx.x] = fault_s;
        }
    }
#undef NCDE_TICK
}

// ------------------------------------------------------------------------------------------------
// adjoint: reverse sweep of (y, a, g_theta) -- adjoint.py:37-145 as ONE persistent kernel
// ------------------------------------------------------------------------------------------------
// Per stage and wave (one 16-sample tile per workgroup, NW waves):
//   forward recompute   register-to-register as in ncde_fwd_fast (hidden layers replicated per wave)
//   own output tiles    P -> m = tanh(P); f += m.dX; dP = a (x) dX * (1 - m^2)
//   dL/dx_L partial     A = Wo^T (LDS image, ds_read_b128 = 4 k-steps), B = dP straight from the D registers
//                       (k-step <-> r, k-sub <-> lane>>4), summed over the NW waves through LDS
//   hidden backward     W1^T / W0^T chains, again register-to-register (same unit permutation)
//   weight gradients    samples are the K dimension: both operands are re-read from wave-private
//                       [unit][sample] LDS images with ONE ds_read_b128 per 4 k-steps (k <-> sample 4*kk+q)
//                       and accumulated in registers for the whole solve (dWo: own tiles; dW1/dW0: one
//                       16x16 tile per wave); bias gradients accumulate per lane and are reduced over the
//                       16 samples once, at the end.
template <int H, int HH, int C, int NL, int NW, int INTERP, int METHOD, int PROF = 0>
__global__ __launch_bounds__(64 * NW, 1) void ncde_adj_fast(KArgs a) {
    unsigned long long prof[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;
#define NCDE_TICK(k)                                                \
    if constexpr (PROF != 0) {                                      \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        prof[k] += now_ - tlast;                                    \
        tlast = now_;                                               \
    }
    constexpr int CP = (C + 3) & ~3, CQ = CP / 4, HB = H / 4, HT = HH / 16, KH = HH / 4, NB = HB / NW;
    constexpr int S = kStages<METHOD>;
    constexpr int NT = 64 * NW;
    constexpr int DXW = INTERP == NCDE_INTERP_LINEAR ? CP : 3 * CP;
    constexpr int EPT = (16 * DXW + NT - 1) / NT;
    constexpr int NTILE = NB * CQ;              // output tiles owned by a wave
    constexpr int HT0 = H / 16;                 // column tiles of dW0
    constexpr int TPW1 = HT * HT / NW, TPW0 = HT * HT0 / NW;
    constexpr int XS = 20;                      // padded sample stride of the [unit][sample] images
    constexpr int IMG = (H + NL * HH) * XS;     // z, x_1..x_NL
    constexpr int PRIV = IMG + HH * XS;            // dP tile scratch aliases the dL/dpre image (disjoint phases)
    static_assert(H % (4 * NW) == 0 && HH % 16 == 0 && H % 16 == 0 && NB <= 4, "shape not tileable");
    static_assert((HT * HT) % NW == 0 && (HT * HT0) % NW == 0, "weight-gradient tiles must split evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* zx = lds;                              // [2][H*16]   stage-state exchange
    float* dxs = zx + 2 * H * 16;                 // [3][16*DXW] control-path ring
    float* red = dxs + 3 * 16 * DXW;              // [NW][HH*16] dL/dx_L partials
    float* woT = red + NW * HH * 16;              // [NW][NTILE][HT][64][4]
    float* boL = woT + NW * NTILE * HT * 256;     // [NW][NTILE][4][4]
    float* privbase = boL + NW * NTILE * 16;      // [NW][PRIV]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = lane & 15, g = lane >> 4;
    const int b0 = blockIdx.x * NCDE_TILE;
    const int bs = b0 + s;
    const bool valid = bs < a.B;
    float* priv = privbase + wave * PRIV;
    float* img = priv;                            // rows: z [0,H), x_l [H+(l-1)*HH, H+l*HH)
    float* dpimg = priv + IMG;                    // [HH][XS]   w * dL/dpre of the current layer
    float* dptile = dpimg;                        // [16][XS]   w * dP of the current output tile (aliases dpimg)
    const float* woTw = woT + wave * NTILE * HT * 256;
    const float* boLw = boL + wave * NTILE * 16;

    // ---- weights -> registers / LDS images ------------------------------------------------------------
    float w0[HT][HB], w1[HT][KH], wo[NB][CQ][KH], w1T[HT][KH], w0T[KH];
    f32x4 bias0[HT], bias1[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) {
        const int unitA = 4 * (4 * t + (s & 3)) + (s >> 2);
#pragma unroll
        for (int ks = 0; ks < HB; ++ks) w0[t][ks] = a.W[0][unitA * H + 4 * ks + g];
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) {
            w1[t][ks] = a.W[1][unitA * HH + 4 * ks + g];
            w1T[t][ks] = a.W[1][(4 * ks + g) * HH + unitA];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int unitD = 4 * (4 * t + r) + g;
            bias0[t][r] = a.b[0][unitD];
            bias1[t][r] = a.b[1][unitD];
        }
    }
    {   // W0^T rows for the state entries this wave owns: tile row i <-> h = 4*(wave*NB + (i&3)) + (i>>2)
        const int r_own = s & 3;
        const int hrow = 4 * (wave * NB + r_own) + (s >> 2);
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) w0T[ks] = r_own < NB ? a.W[0][(4 * ks + g) * H + hrow] : 0.0f;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int hb = wave * NB + nb;
#pragma unroll
        for (int cq = 0; cq < CQ; ++cq) {
            const int hA = 4 * hb + (s >> 2), cA = 4 * cq + (s & 3);
#pragma unroll
            for (int ks = 0; ks < KH; ++ks) wo[nb][cq][ks] = cA < C ? NCDE_TANH_PRESCALE * a.Wo[(hA * C + cA) * HH + 4 * ks + g] : 0.0f;
        }
    }
    for (int e = tid; e < NW * NTILE * HT * 256; e += NT) {  // Wo^T image
        const int r = e & 3, l = (e >> 2) & 63, rest = e >> 8;
        const int tp = rest % HT, tau = (rest / HT) % NTILE, wv = rest / (HT * NTILE);
        const int nb = tau / CQ, cq = tau - nb * CQ;
        const int h = 4 * (wv * NB + nb) + (l >> 4), c = 4 * cq + r;
        const int jrow = 4 * (4 * tp + (l & 3)) + ((l & 15) >> 2);
        woT[e] = c < C ? a.Wo[(h * C + c) * HH + jrow] : 0.0f;
    }
    for (int e = tid; e < NW * NTILE * 16; e += NT) {  // bo image [wave][tile][g][r]
        const int r = e & 3, gg = (e >> 2) & 3, rest = e >> 4;
        const int tau = rest % NTILE, wv = rest / NTILE;
        const int nb = tau / CQ, cq = tau - nb * CQ;
        const int h = 4 * (wv * NB + nb) + gg, c = 4 * cq + r;
        boL[e] = c < C ? NCDE_TANH_PRESCALE * a.bo[h * C + c] : 0.0f;
    }

    // ---- control-path staging (reverse order: piece p needs rows p+1 and p) ----------------------------
    const float* eptr[EPT];
    float eprev[EPT], enext[EPT];
    bool eok[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int e = tid + q * NT;
        const int es = e / DXW, ec = e - es * DXW;
        const int part = ec / CP, c = ec - part * CP;
        eok[q] = e < 16 * DXW && c < a.Cc && (b0 + es) < a.B;      // a.Cc: channels of the coefficient tensor (= C unless zero-padded)
        const long long base = (long long)(eok[q] ? b0 + es : 0) * a.cs_b;
        eptr[q] = a.coeffs + base + (INTERP == NCDE_INTERP_LINEAR ? c : (part + 1) * a.Cc + c);
        eprev[q] = 0.0f;
        enext[q] = 0.0f;
    }
    auto stage_load = [&](int piece) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) enext[q] = eok[q] ? eptr[q][(long long)piece * a.cs_t] : 0.0f;
    };
    auto stage_store = [&](int piece) {
        float* dst = dxs + (piece % 3) * 16 * DXW;
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = tid + q * NT;
            if (e < 16 * DXW) dst[e] = INTERP == NCDE_INTERP_LINEAR ? eprev[q] - enext[q] : enext[q];
            eprev[q] = enext[q];
        }
    };
    const int p_hi = a.n_pieces - 1;
    if (INTERP == NCDE_INTERP_LINEAR) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) eprev[q] = eok[q] ? eptr[q][(long long)(p_hi + 1) * a.cs_t] : 0.0f;  // last row
    }
    stage_load(p_hi);
    stage_store(p_hi);
    if (p_hi >= 1) {
        stage_load(p_hi - 1);
        stage_store(p_hi - 1);
    }

    // ---- state ------------------------------------------------------------------------------------------
    const int last_row = a.n_out - 1;
    float y0[NB], ky1[NB], ky2[NB], a0[NB], ka1[NB], ka2[NB], as_[NB], zreg[HB];
#pragma unroll
    for (int ks = 0; ks < HB; ++ks) zreg[ks] = valid ? a.z_out[((long long)bs * a.n_out + last_row) * H + 4 * ks + g] : 0.0f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const long long o = ((long long)bs * a.n_out + last_row) * H + 4 * (wave * NB + nb) + g;
        y0[nb] = valid ? a.z_out[o] : 0.0f;
        a0[nb] = valid ? a.grad_out[o] : 0.0f;
        as_[nb] = a0[nb];
        ky1[nb] = ky2[nb] = ka1[nb] = ka2[nb] = 0.0f;
    }
    // ---- gradient accumulators ----------------------------------------------------------------------------
    f32x4 gWo[NTILE][HT], gW1[TPW1], gW0[TPW0], gbo[NTILE];
    float gb1[KH], gb0[KH];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NTILE; ++i) {
        gbo[i] = zero4;
#pragma unroll
        for (int t = 0; t < HT; ++t) gWo[i][t] = zero4;
    }
#pragma unroll
    for (int i = 0; i < TPW1; ++i) gW1[i] = zero4;
#pragma unroll
    for (int i = 0; i < TPW0; ++i) gW0[i] = zero4;
#pragma unroll
    for (int i = 0; i < KH; ++i) gb1[i] = gb0[i] = 0.0f;
    __syncthreads();

    int zpar = 0;
    if constexpr (PROF != 0) tlast = __builtin_readcyclecounter();
    for (int n = a.T - 1; n >= 1; --n) {  // reverse step: knot n -> n-1 (negated time -n -> -(n-1))
        if (n - 3 >= 0) stage_load(n - 3);  // piece needed by the NEXT-next step; stored at the end of this one
        float ynext[NB], gnext[NB], znext[HB];
        if (a.output == NCDE_OUT_KNOTS) {
#pragma unroll
            for (int ks = 0; ks < HB; ++ks) znext[ks] = valid ? a.z_out[((long long)bs * a.n_out + (n - 1)) * H + 4 * ks + g] : 0.0f;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const long long o = ((long long)bs * a.n_out + (n - 1)) * H + 4 * (wave * NB + nb) + g;
                ynext[nb] = valid ? a.z_out[o] : 0.0f;
                gnext[nb] = valid ? a.grad_out[o] : 0.0f;
            }
        }
#pragma unroll 1
        for (int j = 0; j < S; ++j) {
            const float t = -(-(float)n + stage_offset(METHOD, j));
            const int idx = piece_index(t, a.n_pieces);
            const float frac = t - (float)idx;
            const float wq = stage_weight(METHOD, j);
            const float* dxp = dxs + (idx % 3) * 16 * DXW + s * DXW;
            // ---- forward recompute; keep x_1..x_NL (registers) and their [unit][sample] images (LDS) ------
            float x[NL][KH];
            {
                f32x4 acc[HT];
#pragma unroll
                for (int tt = 0; tt < HT; ++tt) acc[tt] = bias0[tt];
#pragma unroll
                for (int ks = 0; ks < HB; ++ks)
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) acc[tt] = mfma16(w0[tt][ks], zreg[ks], acc[tt]);
#pragma unroll
                for (int tt = 0; tt < HT; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[0][4 * tt + r] = relu_dev(acc[tt][r]);
#pragma unroll
                for (int l = 1; l < NL; ++l) {
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) acc[tt] = bias1[tt];
#pragma unroll
                    for (int ks = 0; ks < KH; ++ks)
#pragma unroll
                        for (int tt = 0; tt < HT; ++tt) acc[tt] = mfma16(w1[tt][ks], x[l - 1][ks], acc[tt]);
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) x[l][4 * tt + r] = relu_dev(acc[tt][r]);
                }
            }
            if (wq != 0.0f) {
#pragma unroll
                for (int ks = 0; ks < HB; ++ks) img[(4 * ks + g) * XS + s] = zreg[ks];
#pragma unroll
                for (int l = 0; l < NL; ++l)
#pragma unroll
                    for (int ks = 0; ks < KH; ++ks) img[(H + l * HH + 4 * ks + g) * XS + s] = x[l][ks];
                wave_lds_order();
            }
            NCDE_TICK(0)
            // B operands of the dWo GEMM: x_NL[j = 16t + n][samples 4g..4g+3]
            f32x4 xB[HT];
            if (wq != 0.0f) {
#pragma unroll
                for (int tt = 0; tt < HT; ++tt) xB[tt] = *reinterpret_cast<const f32x4*>(img + (H + (NL - 1) * HH + 16 * tt + s) * XS + 4 * g);
            }
            // ---- output tiles owned by this wave ---------------------------------------------------------------
            float kout[NB];
            f32x4 accJ[HT];
#pragma unroll
            for (int tt = 0; tt < HT; ++tt) accJ[tt] = zero4;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) kout[nb] = 0.0f;
#pragma unroll
            for (int cq = 0; cq < CQ; ++cq) {
                f32x4 o[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) o[nb] = *reinterpret_cast<const f32x4*>(boLw + ((nb * CQ + cq) * 4 + g) * 4);
#pragma unroll
                for (int ks = 0; ks < KH; ++ks)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) o[nb] = mfma16(wo[nb][cq][ks], x[NL - 1][ks], o[nb]);
                f32x4 dx;
                if constexpr (INTERP == NCDE_INTERP_LINEAR) {
                    dx = *reinterpret_cast<const f32x4*>(dxp + 4 * cq);
                } else {
                    const f32x4 cb = *reinterpret_cast<const f32x4*>(dxp + 4 * cq);
                    const f32x4 cc = *reinterpret_cast<const f32x4*>(dxp + CP + 4 * cq);
                    const f32x4 cd = *reinterpret_cast<const f32x4*>(dxp + 2 * CP + 4 * cq);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float inner = cc[r] + cd[r] * frac;
                        dx[r] = cb[r] + inner * frac;
                    }
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int tau = nb * CQ + cq;
                    f32x4 dP;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float m = tanh_prescaled(o[nb][r]);
                        kout[nb] = fmaf(m, dx[r], kout[nb]);
                        dP[r] = (as_[nb] * dx[r]) * (1.0f - m * m);
                    }
                    // dL/dx_L partial: k-step <-> r
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) {
                        const f32x4 av = *reinterpret_cast<const f32x4*>(woTw + ((tau * HT + tt) * 64 + lane) * 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) accJ[tt] = mfma16(av[r], dP[r], accJ[tt]);
                    }
                    if (wq != 0.0f) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = wq * dP[r];
                            gbo[tau][r] += v;
                            dptile[(4 * g + r) * XS + s] = v;
                        }
                        wave_lds_order();
                        const f32x4 av = *reinterpret_cast<const f32x4*>(dptile + s * XS + 4 * g);
                        wave_lds_order();
#pragma unroll
                        for (int tt = 0; tt < HT; ++tt)
#pragma unroll
                            for (int q = 0; q < 4; ++q) gWo[tau][tt] = mfma16(av[q], xB[tt][q], gWo[tau][tt]);
                    }
                }
            }
            NCDE_TICK(1)
            // ---- sum the dL/dx_L partials over the waves --------------------------------------------------------
            float gpre[KH];
            if constexpr (NW == 1) {
#pragma unroll
                for (int tt = 0; tt < HT; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) gpre[4 * tt + r] = accJ[tt][r];
            } else {
#pragma unroll
                for (int tt = 0; tt < HT; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[wave * HH * 16 + (4 * (4 * tt + r) + g) * 16 + s] = accJ[tt][r];
                __syncthreads();
#pragma unroll
                for (int ks = 0; ks < KH; ++ks) {
                    float v = red[(4 * ks + g) * 16 + s];
#pragma unroll
                    for (int wv = 1; wv < NW; ++wv) v += red[wv * HH * 16 + (4 * ks + g) * 16 + s];
                    gpre[ks] = v;
                }
            }
#pragma unroll
            for (int ks = 0; ks < KH; ++ks) gpre[ks] = x[NL - 1][ks] > 0.0f ? gpre[ks] : 0.0f;
            NCDE_TICK(2)
            // ---- hidden layers backward (shared W1), then W0 ------------------------------------------------------
#pragma unroll
            for (int l = NL - 1; l >= 1; --l) {  // layer with input x_l (x[l-1]) and output x_{l+1} (x[l])
                if (wq != 0.0f) {
#pragma unroll
                    for (int ks = 0; ks < KH; ++ks) {
                        const float v = wq * gpre[ks];
                        gb1[ks] += v;
                        dpimg[(4 * ks + g) * XS + s] = v;
                    }
                    wave_lds_order();
#pragma unroll
                    for (int k = 0; k < TPW1; ++k) {
                        const int id = wave * TPW1 + k, tr = id / HT, tc = id - tr * HT;
                        const f32x4 av = *reinterpret_cast<const f32x4*>(dpimg + (16 * tr + s) * XS + 4 * g);
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(img + (H + (l - 1) * HH + 16 * tc + s) * XS + 4 * g);
#pragma unroll
                        for (int q = 0; q < 4; ++q) gW1[k] = mfma16(av[q], bv[q], gW1[k]);
                    }
                    wave_lds_order();
                }
                f32x4 acc[HT];
#pragma unroll
                for (int tt = 0; tt < HT; ++tt) acc[tt] = zero4;
#pragma unroll
                for (int ks = 0; ks < KH; ++ks)
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) acc[tt] = mfma16(w1T[tt][ks], gpre[ks], acc[tt]);
#pragma unroll
                for (int tt = 0; tt < HT; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) gpre[4 * tt + r] = x[l - 1][4 * tt + r] > 0.0f ? acc[tt][r] : 0.0f;
            }
            if (wq != 0.0f) {
#pragma unroll
                for (int ks = 0; ks < KH; ++ks) {
                    const float v = wq * gpre[ks];
                    gb0[ks] += v;
                    dpimg[(4 * ks + g) * XS + s] = v;
                }
                wave_lds_order();
#pragma unroll
                for (int k = 0; k < TPW0; ++k) {
                    const int id = wave * TPW0 + k, tr = id / HT0, tc = id - tr * HT0;
                    const f32x4 av = *reinterpret_cast<const f32x4*>(dpimg + (16 * tr + s) * XS + 4 * g);
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(img + (16 * tc + s) * XS + 4 * g);
#pragma unroll
                    for (int q = 0; q < 4; ++q) gW0[k] = mfma16(av[q], bv[q], gW0[k]);
                }
                wave_lds_order();
            }
            NCDE_TICK(3)
            f32x4 vy = zero4;  // a^T df/dy for the state entries this wave owns
#pragma unroll
            for (int ks = 0; ks < KH; ++ks) vy = mfma16(w0T[ks], gpre[ks], vy);
            // ---- Butcher bookkeeping in negated time: dy/ds = -f, da/ds = +a^T df/dy ----------------------------
            float ys[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                ys[nb] = Combine<METHOD>::apply(j, -kout[nb], y0[nb], ky1[nb], ky2[nb]);
                as_[nb] = Combine<METHOD>::apply(j, vy[nb], a0[nb], ka1[nb], ka2[nb]);
            }
            if (PROF == 2 && blockIdx.x == 0 && wave == 0) {  // debug dump: [stage][5][64]
                float* d = a.out + ((long long)(((a.T - 1 - n) * S + j)) * 5) * 64 + lane;
                d[0] = kout[0]; d[64] = vy[0]; d[128] = gpre[0]; d[192] = ys[0]; d[256] = as_[0];
            }
            if (j == S - 1) {
                if (a.output == NCDE_OUT_KNOTS) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        y0[nb] = ynext[nb];
                        ys[nb] = ynext[nb];
                        a0[nb] += gnext[nb];
                        as_[nb] = a0[nb];
                    }
                } else if (n == 1) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        a0[nb] += valid ? a.grad_out[((long long)bs * a.n_out) * H + 4 * (wave * NB + nb) + g] : 0.0f;
                        as_[nb] = a0[nb];
                    }
                }
                if (n - 3 >= 0) stage_store(n - 3);
            }
            if (j == S - 1 && a.output == NCDE_OUT_KNOTS) {
#pragma unroll
                for (int ks = 0; ks < HB; ++ks) zreg[ks] = znext[ks];
                __syncthreads();  // publishes the control-path ring slot written above
            } else if constexpr (NW == 1) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) zreg[nb] = ys[nb];
            } else {
                float* zw = zx + zpar * H * 16;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) zw[(4 * (wave * NB + nb) + g) * 16 + s] = ys[nb];
                __syncthreads();
#pragma unroll
                for (int ks = 0; ks < HB; ++ks) zreg[ks] = zw[(4 * ks + g) * 16 + s];
                zpar ^= 1;
            }
            NCDE_TICK(4)
        }
    }
    if constexpr (PROF != 0) {
        if (lane == 0) {
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.out) + ((long long)blockIdx.x * NW + wave) * 6;
            for (int k = 0; k < 6; ++k) dst[k] = prof[k];
        }
    }
#undef NCDE_TICK
    // ---- write-out: dL/dz0 and this workgroup's parameter-gradient partial ------------------------------------
    if (valid) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) a.grad_z0[(long long)bs * H + 4 * (wave * NB + nb) + g] = a0[nb];
    }
    float* gp = a.gpart + (long long)blockIdx.x * a.theta_size;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int cq = 0; cq < CQ; ++cq) {
            const int tau = nb * CQ + cq;
            const int h = 4 * (wave * NB + nb) + g;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 4 * cq + r;
                if (c < C) {
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) gp[a.gWo_off + (h * C + c) * HH + 16 * tt + s] = gWo[tau][tt][r];
                }
                const float sum = row16_sum(gbo[tau][r]);
                if (s == 0 && c < C) gp[a.gbo_off + h * C + c] = sum;
            }
        }
#pragma unroll
    for (int k = 0; k < TPW1; ++k) {
        const int id = wave * TPW1 + k, tr = id / HT, tc = id - tr * HT;
#pragma unroll
        for (int r = 0; r < 4; ++r) gp[a.gW_off[1] + (16 * tr + 4 * g + r) * HH + 16 * tc + s] = gW1[k][r];
    }
#pragma unroll
    for (int k = 0; k < TPW0; ++k) {
        const int id = wave * TPW0 + k, tr = id / HT0, tc = id - tr * HT0;
#pragma unroll
        for (int r = 0; r < 4; ++r) gp[a.gW_off[0] + (16 * tr + 4 * g + r) * H + 16 * tc + s] = gW0[k][r];
    }
#pragma unroll
    for (int ks = 0; ks < KH; ++ks) {
        const float s1 = row16_sum(gb1[ks]), s0 = row16_sum(gb0[ks]);
        if (wave == 0 && s == 0) {
            gp[a.gb_off[1] + 4 * ks + g] = s1;
            gp[a.gb_off[0] + 4 * ks + g] = s0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// adjoint, wave-specialised variant: chain waves + gradient waves (two waves per SIMD)
// ------------------------------------------------------------------------------------------------
// One wave per SIMD cannot hide its own VALU/LDS latencies and issues at most ~1 instruction per 4-5 cycles
// (profiles/r01: ncde_adj_fast spends ~20k cycles per stage on 10.9k cycles of MFMA).  Here a workgroup has
// 8 waves = 4 pairs, wave w (chain, "C") and wave w+4 (gradient, "G") sharing SIMD w and the same h-blocks:
//   C: everything ON the stage's dependency chain -- forward recompute, output tiles (P, tanh, f, dP),
//      cross-wave sum of dL/dx_L, hidden-layer backward, a^T df/dy, Butcher bookkeeping, state exchange;
//   G: every GEMM that only CONSUMES dP / dL/dpre -- the dL/dx_L partial (Wo^T resident in G's registers)
//      and ALL parameter-gradient accumulation (dWo, dbo, dW1, dW0, db*), i.e. half of the stage's MFMAs and
//      none of its VALU.  The matrix pipe of SIMD w is fed by G while C is busy in the VALU, and vice versa.
// Hand-off C -> G is through LDS images plus monotone flag words (stage counter), written after the data by
// the same wave (DS ops of a wave are performed in order) and polled by G; the dL/dx_L partials come back
// through the `red` buffer at the stage's first workgroup barrier.  Everything of stage j is consumed before
// the stage's second barrier, so all images are single-buffered.
template <int H, int HH, int C, int NL, int INTERP, int METHOD, int PROF = 0>
__global__ __launch_bounds__(512, 2) void ncde_adj_fast2(KArgs a) {
    unsigned long long prof[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;
#define NCDE_TICK(k)                                                \
    if constexpr (PROF != 0) {                                      \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        prof[k] += now_ - tlast;                                    \
        tlast = now_;                                               \
    }
    constexpr int NW = 4;  // pairs
    constexpr int CP = (C + 3) & ~3, CQ = CP / 4, HB = H / 4, HT = HH / 16, KH = HH / 4, NB = HB / NW;
    constexpr int S = kStages<METHOD>;
    constexpr int NT = 64 * NW;  // threads that stage the control path (the chain waves)
    constexpr int DXW = INTERP == NCDE_INTERP_LINEAR ? CP : 3 * CP;
    constexpr int EPT = (16 * DXW + NT - 1) / NT;
    constexpr int NTILE = NB * CQ;
    constexpr int HT0 = H / 16;
    constexpr int TPW1 = HT * HT / NW, TPW0 = HT * HT0 / NW;
    constexpr int XROWS = H + NL * HH;         // z, x_1..x_NL
    constexpr int NFLAG = 2 * NTILE;           // per pair: one flag per tile and stage parity
    constexpr int NT2 = (NTILE / 2) & ~1;      // dWo tiles done right after barrier A (whole 2-tile blocks); the rest lag one stage
    constexpr int NBLK = NTILE / 2;            // 32-row blocks (tile pairs) of this pair's dWo slice
    static_assert(NTILE % 2 == 0 && HH == 32, "dWo runs as 32x32x16 split-bf16 blocks: tile pairs x 32 hidden units");
    static_assert(H % (4 * NW) == 0 && HH % 16 == 0 && H % 16 == 0 && NB <= 4, "shape not tileable");
    static_assert((HT * HT) % NW == 0 && (HT * HT0) % NW == 0, "weight-gradient tiles must split evenly over the pairs");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* zx = lds;                                  // [2][H*16]
    float* dxs = zx + 2 * H * 16;                     // [3][16*DXW]
    float* red = dxs + 3 * 16 * DXW;                  // [NW][HH*16]
    float* boL = red + NW * HH * 16;                  // [NW][NTILE][4][4]
    float* tiles = boL + NW * NTILE * 16;             // [2][NW][NTILE][16][16]  raw dP, [row][sample], by stage parity
    float* ximg = tiles + 2 * NW * NTILE * 256;       // [2][XROWS][16]  z, x_1..x_NL (written by chain wave 0), by parity
    float* dpimg = ximg + 2 * XROWS * 16;             // [NL][HH][16]    raw dL/dpre of each hidden layer (chain wave 0)
    int* flags = reinterpret_cast<int*>(dpimg + NL * HH * 16);  // [NW][NFLAG]
    // A-operand / bias images of the small transposed weights: read per use so they do not occupy the chain
    // waves' registers for the whole solve (two waves per SIMD = 256 registers each)
    float* biasL = reinterpret_cast<float*>(flags + NW * NFLAG);  // [2][HT][4 g][4 r]
    float* w1TL = biasL + 2 * HT * 16;                // [HT][KH/4][64 lanes][4]
    float* w0TL = w1TL + HT * (KH / 4) * 256;         // [NW][KH/4][64 lanes][4]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_chain = wave < NW;
    const int pw = is_chain ? wave : wave - NW;       // pair index = SIMD = owner of h-blocks pw*NB..
    const int s = lane & 15, g = lane >> 4;
    const int b0 = blockIdx.x * NCDE_TILE;
    const int bs = b0 + s;
    const bool valid = bs < a.B;
    float* my_tiles = tiles + pw * NTILE * 256;       // + parity * NW*NTILE*256
    // explicit LDS address space: a generic volatile pointer would be lowered to (slow) flat_store/flat_load
    volatile __attribute__((address_space(3))) int* my_flags =
        (volatile __attribute__((address_space(3))) int*)(flags + pw * NFLAG);
    const float* boLw = boL + pw * NTILE * 16;

    for (int e = tid; e < NW * NFLAG; e += 512) flags[e] = 0;
    for (int e = tid; e < NW * NTILE * 16; e += 512) {
        const int r = e & 3, gg = (e >> 2) & 3, rest = e >> 4;
        const int tau = rest % NTILE, wv = rest / NTILE;
        const int nb = tau / CQ, cq = tau - nb * CQ;   // bias image is indexed nb-major by the chain waves
        const int h = 4 * (wv * NB + nb) + gg, c = 4 * cq + r;
        boL[e] = c < C ? NCDE_TANH_PRESCALE * a.bo[h * C + c] : 0.0f;
    }
    for (int e = tid; e < 2 * HT * 16; e += 512) {
        const int r = e & 3, gg = (e >> 2) & 3, t = (e >> 4) % HT, layer = e / (16 * HT);
        biasL[e] = a.b[layer][4 * (4 * t + r) + gg];
    }
    for (int e = tid; e < HT * (KH / 4) * 256; e += 512) {
        const int q = e & 3, l = (e >> 2) & 63, rest = e >> 8;
        const int k4 = rest % (KH / 4), t = rest / (KH / 4);
        const int unitA = 4 * (4 * t + (l & 3)) + ((l & 15) >> 2);
        w1TL[e] = a.W[1][(4 * (4 * k4 + q) + (l >> 4)) * HH + unitA];
    }
    for (int e = tid; e < NW * (KH / 4) * 256; e += 512) {
        const int q = e & 3, l = (e >> 2) & 63, rest = e >> 8;
        const int k4 = rest % (KH / 4), wv = rest / (KH / 4);
        const int r_own = l & 3;
        const int hrow = 4 * (wv * NB + r_own) + ((l & 15) >> 2);
        w0TL[e] = r_own < NB ? a.W[0][(4 * (4 * k4 + q) + (l >> 4)) * H + hrow] : 0.0f;
    }
    const int n_stage_total = (a.T - 1) * S;

    if (is_chain) {
        // =================================================================================================
        // chain wave
        // =================================================================================================
        float w0[HT][HB], w1[HT][KH], wo[NB][CQ][KH];
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            const int unitA = 4 * (4 * t + (s & 3)) + (s >> 2);
#pragma unroll
            for (int ks = 0; ks < HB; ++ks) w0[t][ks] = a.W[0][unitA * H + 4 * ks + g];
#pragma unroll
            for (int ks = 0; ks < KH; ++ks) w1[t][ks] = a.W[1][unitA * HH + 4 * ks + g];
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int hb = pw * NB + nb;
#pragma unroll
            for (int cq = 0; cq < CQ; ++cq) {
                const int hA = 4 * hb + (s >> 2), cA = 4 * cq + (s & 3);
#pragma unroll
                for (int ks = 0; ks < KH; ++ks) wo[nb][cq][ks] = cA < C ? NCDE_TANH_PRESCALE * a.Wo[(hA * C + cA) * HH + 4 * ks + g] : 0.0f;
            }
        }
        // control-path staging (reverse order), by the 256 chain threads
        const float* eptr[EPT];
        float eprev[EPT], enext[EPT];
        bool eok[EPT];
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = tid + q * NT;
            const int es = e / DXW, ec = e - es * DXW;
            const int part = ec / CP, c = ec - part * CP;
            eok[q] = e < 16 * DXW && c < a.Cc && (b0 + es) < a.B;      // a.Cc: channels of the coefficient tensor (= C unless zero-padded)
            const long long base = (long long)(eok[q] ? b0 + es : 0) * a.cs_b;
            eptr[q] = a.coeffs + base + (INTERP == NCDE_INTERP_LINEAR ? c : (part + 1) * a.Cc + c);
            eprev[q] = 0.0f;
            enext[q] = 0.0f;
        }
        auto stage_load = [&](int piece) {
#pragma unroll
            for (int q = 0; q < EPT; ++q) enext[q] = eok[q] ? eptr[q][(long long)piece * a.cs_t] : 0.0f;
        };
        auto stage_store = [&](int piece) {
            float* dst = dxs + (piece % 3) * 16 * DXW;
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
                const int e = tid + q * NT;
                if (e < 16 * DXW) dst[e] = INTERP == NCDE_INTERP_LINEAR ? eprev[q] - enext[q] : enext[q];
                eprev[q] = enext[q];
            }
        };
        const int p_hi = a.n_pieces - 1;
        if (INTERP == NCDE_INTERP_LINEAR) {
#pragma unroll
            for (int q = 0; q < EPT; ++q) eprev[q] = eok[q] ? eptr[q][(long long)(p_hi + 1) * a.cs_t] : 0.0f;
        }
        stage_load(p_hi);
        stage_store(p_hi);
        if (p_hi >= 1) {
            stage_load(p_hi - 1);
            stage_store(p_hi - 1);
        }
        const int last_row = a.n_out - 1;
        float y0[NB], ky1[NB], ky2[NB], a0[NB], ka1[NB], ka2[NB], as_[NB], zreg[HB];
#pragma unroll
        for (int ks = 0; ks < HB; ++ks) zreg[ks] = valid ? a.z_out[((long long)bs * a.n_out + last_row) * H + 4 * ks + g] : 0.0f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const long long o = ((long long)bs * a.n_out + last_row) * H + 4 * (pw * NB + nb) + g;
            y0[nb] = valid ? a.z_out[o] : 0.0f;
            a0[nb] = valid ? a.grad_out[o] : 0.0f;
            as_[nb] = a0[nb];
            ky1[nb] = ky2[nb] = ka1[nb] = ka2[nb] = 0.0f;
        }
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        __syncthreads();

        int zpar = 0, sc = 0;
        if constexpr (PROF != 0) tlast = __builtin_readcyclecounter();
        for (int n = a.T - 1; n >= 1; --n) {
            if (n - 3 >= 0) stage_load(n - 3);
#pragma unroll 1
            for (int j = 0; j < S; ++j) {
                ++sc;
                const float t = -(-(float)n + stage_offset(METHOD, j));
                const int idx = piece_index(t, a.n_pieces);
                const float frac = t - (float)idx;
                const float wq = stage_weight(METHOD, j);
                const float* dxp = dxs + (idx % 3) * 16 * DXW + s * DXW;
                // ---- forward recompute ------------------------------------------------------------------------
                float x[NL][KH];
                {
                    f32x4 acc[HT];
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) acc[tt] = *reinterpret_cast<const f32x4*>(biasL + (tt * 4 + g) * 4);
#pragma unroll
                    for (int ks = 0; ks < HB; ++ks)
#pragma unroll
                        for (int tt = 0; tt < HT; ++tt) acc[tt] = mfma16(w0[tt][ks], zreg[ks], acc[tt]);
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) x[0][4 * tt + r] = relu_dev(acc[tt][r]);
#pragma unroll
                    for (int l = 1; l < NL; ++l) {
#pragma unroll
                        for (int tt = 0; tt < HT; ++tt) acc[tt] = *reinterpret_cast<const f32x4*>(biasL + ((HT + tt) * 4 + g) * 4);
#pragma unroll
                        for (int ks = 0; ks < KH; ++ks)
#pragma unroll
                            for (int tt = 0; tt < HT; ++tt) acc[tt] = mfma16(w1[tt][ks], x[l - 1][ks], acc[tt]);
#pragma unroll
                        for (int tt = 0; tt < HT; ++tt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) x[l][4 * tt + r] = relu_dev(acc[tt][r]);
                    }
                }
                NCDE_TICK(0)
                const int par = sc & 1;
                if (wq != 0.0f && pw == 0) {  // [unit][sample] images for the gradient waves (identical in every pair)
                    float* xi = ximg + par * XROWS * 16;
#pragma unroll
                    for (int ks = 0; ks < HB; ++ks) xi[(4 * ks + g) * 16 + s] = zreg[ks];
#pragma unroll
                    for (int l = 0; l < NL; ++l)
#pragma unroll
                        for (int ks = 0; ks < KH; ++ks) xi[(H + l * HH + 4 * ks + g) * 16 + s] = x[l][ks];
                }
                // ---- output tiles: P, tanh, f, dP -> LDS tile + flag ----------------------------------------------
                float kout[NB];
                float sdx = 0.0f;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) kout[nb] = 0.0f;
#pragma unroll
                for (int cq = 0; cq < CQ; ++cq) {
                    f32x4 o[NB];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) o[nb] = *reinterpret_cast<const f32x4*>(boLw + ((nb * CQ + cq) * 4 + g) * 4);
#pragma unroll
                    for (int ks = 0; ks < KH; ++ks)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) o[nb] = mfma16(wo[nb][cq][ks], x[NL - 1][ks], o[nb]);
                    f32x4 dx;
                    if constexpr (INTERP == NCDE_INTERP_LINEAR) {
                        dx = *reinterpret_cast<const f32x4*>(dxp + 4 * cq);
                    } else {
                        const f32x4 cb = *reinterpret_cast<const f32x4*>(dxp + 4 * cq);
                        const f32x4 cc = *reinterpret_cast<const f32x4*>(dxp + CP + 4 * cq);
                        const f32x4 cd = *reinterpret_cast<const f32x4*>(dxp + 2 * CP + 4 * cq);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float inner = cc[r] + cd[r] * frac;
                            dx[r] = cb[r] + inner * frac;
                        }
                    }
                    // with r = 1/(exp(2P)+1):  tanh = 1 - 2r,  1 - tanh^2 = 4 r (1 - r).  Per value: v_exp, add, v_rcp,
                    // fma (r - r^2), mul (dP), fma (sum r*dx); f = sum_c dx - 2 sum_c r dx is assembled after the loop.
#pragma unroll
                    for (int r = 0; r < 4; ++r) sdx += dx[r];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const int tau = cq * NB + nb;  // tiles are numbered in publication order
                        float* tl = my_tiles + par * (NW * NTILE * 256) + tau * 256;
                        const float a4 = 4.0f * as_[nb];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float rr = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(o[nb][r]) + 1.0f);
                            kout[nb] = fmaf(rr, dx[r], kout[nb]);
                            tl[(4 * g + r) * 16 + s] = (a4 * dx[r]) * fmaf(-rr, rr, rr);
                        }
                        wave_lds_order();
                        my_flags[par * NTILE + tau] = sc;
                    }
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) kout[nb] = fmaf(-2.0f, kout[nb], sdx);
                NCDE_TICK(1)
#ifdef NCDE_V2_NOFLAGS
                __syncthreads();
#endif
                __syncthreads();  // barrier A: the gradient waves have published their dL/dx_L partials
                NCDE_TICK(2)
                float gpre[KH];
#pragma unroll
                for (int ks = 0; ks < KH; ++ks) {
                    float v = red[(4 * ks + g) * 16 + s];
#pragma unroll
                    for (int wv = 1; wv < NW; ++wv) v += red[wv * HH * 16 + (4 * ks + g) * 16 + s];
                    gpre[ks] = x[NL - 1][ks] > 0.0f ? v : 0.0f;
                }
                // ---- hidden layers backward ---------------------------------------------------------------------
#pragma unroll
                for (int l = NL - 1; l >= 1; --l) {
                    if (wq != 0.0f && pw == 0) {
#pragma unroll
                        for (int ks = 0; ks < KH; ++ks) dpimg[(l * HH + 4 * ks + g) * 16 + s] = gpre[ks];
                    }
                    f32x4 acc[HT];
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) acc[tt] = zero4;
#pragma unroll
                    for (int k4 = 0; k4 < KH / 4; ++k4)
#pragma unroll
                        for (int tt = 0; tt < HT; ++tt) {
                            const f32x4 wv = *reinterpret_cast<const f32x4*>(w1TL + ((tt * (KH / 4) + k4) * 64 + lane) * 4);
#pragma unroll
                            for (int q = 0; q < 4; ++q) acc[tt] = mfma16(wv[q], gpre[4 * k4 + q], acc[tt]);
                        }
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) gpre[4 * tt + r] = x[l - 1][4 * tt + r] > 0.0f ? acc[tt][r] : 0.0f;
                }
                if (wq != 0.0f && pw == 0) {
#pragma unroll
                    for (int ks = 0; ks < KH; ++ks) dpimg[(4 * ks + g) * 16 + s] = gpre[ks];
                }
                f32x4 vy = zero4;
#pragma unroll
                for (int k4 = 0; k4 < KH / 4; ++k4) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(w0TL + ((pw * (KH / 4) + k4) * 64 + lane) * 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) vy = mfma16(wv[q], gpre[4 * k4 + q], vy);
                }
                NCDE_TICK(3)
                float ys[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    ys[nb] = Combine<METHOD>::apply(j, -kout[nb], y0[nb], ky1[nb], ky2[nb]);
                    as_[nb] = Combine<METHOD>::apply(j, vy[nb], a0[nb], ka1[nb], ka2[nb]);
                }
                if (PROF == 2 && blockIdx.x == 0 && pw == 0) {  // debug dump: [stage][5][64]
                    float* d = a.out + ((long long)(sc - 1) * 5) * 64 + lane;
                    d[0] = kout[0]; d[64] = vy[0]; d[128] = gpre[0]; d[192] = ys[0]; d[256] = as_[0];
                }
                if (j == S - 1) {
                    if (a.output == NCDE_OUT_KNOTS) {  // reset y to the stored knot value, add dL/dz of that knot
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            const long long o = ((long long)bs * a.n_out + (n - 1)) * H + 4 * (pw * NB + nb) + g;
                            y0[nb] = valid ? a.z_out[o] : 0.0f;
                            ys[nb] = y0[nb];
                            a0[nb] += valid ? a.grad_out[o] : 0.0f;
                            as_[nb] = a0[nb];
                        }
                    } else if (n == 1) {
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            a0[nb] += valid ? a.grad_out[((long long)bs * a.n_out) * H + 4 * (pw * NB + nb) + g] : 0.0f;
                            as_[nb] = a0[nb];
                        }
                    }
                    if (n - 3 >= 0) stage_store(n - 3);
                }
                if (j == S - 1 && a.output == NCDE_OUT_KNOTS) {
#pragma unroll
                    for (int ks = 0; ks < HB; ++ks) zreg[ks] = valid ? a.z_out[((long long)bs * a.n_out + (n - 1)) * H + 4 * ks + g] : 0.0f;
                    __syncthreads();  // barrier B
                } else {
                    float* zw = zx + zpar * H * 16;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) zw[(4 * (pw * NB + nb) + g) * 16 + s] = ys[nb];
                    __syncthreads();  // barrier B
#pragma unroll
                    for (int ks = 0; ks < HB; ++ks) zreg[ks] = zw[(4 * ks + g) * 16 + s];
                    zpar ^= 1;
                }
                NCDE_TICK(4)
            }
        }
        if constexpr (PROF != 0) {
            if (lane == 0) {
                unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.out) + ((long long)blockIdx.x * 8 + wave) * 6;
                for (int k = 0; k < 6; ++k) dst[k] = prof[k];
            }
        }
        if (valid) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) a.grad_z0[(long long)bs * H + 4 * (pw * NB + nb) + g] = a0[nb];
        }
    } else {
        // =================================================================================================
        // gradient wave
        // =================================================================================================
        // (static s_setprio for this younger wave was tried: its own work gets ~40 % faster, the chain wave ~15 %
        // slower -- zero-sum on the shared SIMD, net -5 %; left at default priority)
        // (static s_setprio for this younger wave was tried: its own work gets ~40 % faster, the chain wave ~15 %
        // slower -- zero-sum on the shared SIMD, net -5 %; left at default priority)
        float woT[NTILE][HT][4];
#pragma unroll
        for (int tau = 0; tau < NTILE; ++tau) {
            const int cq = tau / NB, nb = tau - cq * NB;
            const int h = 4 * (pw * NB + nb) + g;
#pragma unroll
            for (int tp = 0; tp < HT; ++tp) {
                const int jrow = 4 * (4 * tp + (s & 3)) + (s >> 2);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 4 * cq + r;
                    woT[tau][tp][r] = c < C ? a.Wo[(h * C + c) * HH + jrow] : 0.0f;
                }
            }
        }
        f32x16 gWo[NBLK];  // dWo block (tiles 2b, 2b+1) x 32 hidden units, D layout of v_mfma_f32_32x32x16_bf16
        f32x4 gW1[TPW1], gW0[TPW0];
        float gbo[NBLK], gb1[TPW1], gb0[TPW0];
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NBLK; ++i) {
            gbo[i] = 0.0f;
#pragma unroll
            for (int q = 0; q < 16; ++q) gWo[i][q] = 0.0f;
        }
#pragma unroll
        for (int i = 0; i < TPW1; ++i) { gW1[i] = zero4; gb1[i] = 0.0f; }
#pragma unroll
        for (int i = 0; i < TPW0; ++i) { gW0[i] = zero4; gb0[i] = 0.0f; }
        __syncthreads();

        auto wait_flag = [&](int slot, int want) {
            while (__builtin_amdgcn_readfirstlane(my_flags[slot]) != want) __builtin_amdgcn_s_sleep(1);
            wave_lds_order();
        };
        // dWo of the tiles [t_lo, t_hi) of parity buffer `par` (stage weight w): samples are the K dimension, so
        // a 2-tile block is ONE 32(rows) x 32(units) x 16(samples) product = 6 split-bf16 MFMAs (fp32-equivalent)
        auto dwo_range = [&](int par, float w, auto t_lo_c, auto t_hi_c) {
            constexpr int t_lo = decltype(t_lo_c)::value, t_hi = decltype(t_hi_c)::value;
            static_assert(t_lo % 2 == 0 && t_hi % 2 == 0, "whole blocks only");
            const int i32 = lane & 31, kg = lane >> 5;
            const float* xi = ximg + par * XROWS * 16;
            float bv[8];
            {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(xi + (H + (NL - 1) * HH + i32) * 16 + 8 * kg);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(xi + (H + (NL - 1) * HH + i32) * 16 + 8 * kg + 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) { bv[q] = w * b0[q]; bv[4 + q] = w * b1[q]; }
            }
            const Split3 Bs = split8(bv);
#pragma unroll
            for (int blk = t_lo / 2; blk < t_hi / 2; ++blk) {
                const float* tl = my_tiles + par * (NW * NTILE * 256) + (2 * blk + (i32 >> 4)) * 256 + (i32 & 15) * 16 + 8 * kg;
                const f32x4 a0v = *reinterpret_cast<const f32x4*>(tl);
                const f32x4 a1v = *reinterpret_cast<const f32x4*>(tl + 4);
                float av[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) { av[q] = a0v[q]; av[4 + q] = a1v[q]; }
                gbo[blk] += w * (((av[0] + av[1]) + (av[2] + av[3])) + ((av[4] + av[5]) + (av[6] + av[7])));
                const Split3 As = split8(av);
                f32x16 c = gWo[blk];
                c = mfma_bf32(As.lo, Bs.hi, c);
                c = mfma_bf32(As.hi, Bs.lo, c);
                c = mfma_bf32(As.mid, Bs.mid, c);
                c = mfma_bf32(As.mid, Bs.hi, c);
                c = mfma_bf32(As.hi, Bs.mid, c);
                c = mfma_bf32(As.hi, Bs.hi, c);
                gWo[blk] = c;
            }
        };
        // hidden-layer weight/bias gradients of the stage whose x images have parity `par`
        auto dw_hidden = [&](int par, float w) {
            const float* xi = ximg + par * XROWS * 16;
#pragma unroll
            for (int l = NL - 1; l >= 1; --l) {  // layer with input x_l (image rows H+(l-1)*HH), dL/dpre image l
#pragma unroll
                for (int k = 0; k < TPW1; ++k) {
                    const int id = pw * TPW1 + k, tr = id / HT, tc = id - tr * HT;
                    const f32x4 av = *reinterpret_cast<const f32x4*>(dpimg + (l * HH + 16 * tr + s) * 16 + 4 * g);
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(xi + (H + (l - 1) * HH + 16 * tc + s) * 16 + 4 * g);
                    if (tc == 0) gb1[k] += w * ((av[0] + av[1]) + (av[2] + av[3]));
#pragma unroll
                    for (int q = 0; q < 4; ++q) gW1[k] = mfma16(av[q], w * bv[q], gW1[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < TPW0; ++k) {
                const int id = pw * TPW0 + k, tr = id / HT0, tc = id - tr * HT0;
                const f32x4 av = *reinterpret_cast<const f32x4*>(dpimg + (16 * tr + s) * 16 + 4 * g);
                const f32x4 bv = *reinterpret_cast<const f32x4*>(xi + (16 * tc + s) * 16 + 4 * g);
                if (tc == 0) gb0[k] += w * ((av[0] + av[1]) + (av[2] + av[3]));
#pragma unroll
                for (int q = 0; q < 4; ++q) gW0[k] = mfma16(av[q], w * bv[q], gW0[k]);
            }
        };
        using ic0 = std::integral_constant<int, 0>;
        using ic_half = std::integral_constant<int, NT2>;
        using ic_all = std::integral_constant<int, NTILE>;
        int sc = 0;
        float wprev = 0.0f;
        if constexpr (PROF != 0) tlast = __builtin_readcyclecounter();
        for (int n = a.T - 1; n >= 1; --n) {
#pragma unroll 1
            for (int j = 0; j < S; ++j) {
                ++sc;
                const int par = sc & 1;
                const float wq = stage_weight(METHOD, j);
                // Static schedule against the chain wave's timeline (tiles are published at a steady rate):
                //   (1a) lagging work of the previous stage that fits before the first tile group is complete
                //   (2a) dL/dx_L of tile group 0 (ONE flag poll, all LDS reads issued up front)
                //   (1b) the remaining lagging dWo block
                //   (2b) dL/dx_L of tile group 1 -> partial -> barrier A
                constexpr int TG = NTILE / 2;  // tiles per group
                if (wprev != 0.0f) {
                    dw_hidden(par ^ 1, wprev);
                    dwo_range(par ^ 1, wprev, ic_half{}, std::integral_constant<int, NT2 + 2 * ((NTILE - NT2) / 4)>{});
                }
                NCDE_TICK(0)
                f32x4 accJ[HT];
#pragma unroll
                for (int tt = 0; tt < HT; ++tt) accJ[tt] = zero4;
                auto dxl_group = [&](auto t_lo_c, auto t_hi_c) {
                    constexpr int t_lo = decltype(t_lo_c)::value, t_hi = decltype(t_hi_c)::value;
                    float bq[t_hi - t_lo][4];
#pragma unroll
                    for (int tau = t_lo; tau < t_hi; ++tau)
#pragma unroll
                        for (int r = 0; r < 4; ++r) bq[tau - t_lo][r] = my_tiles[par * (NW * NTILE * 256) + tau * 256 + (4 * g + r) * 16 + s];
#pragma unroll
                    for (int tau = t_lo; tau < t_hi; ++tau)
#pragma unroll
                        for (int tt = 0; tt < HT; ++tt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) accJ[tt] = mfma16(woT[tau][tt][r], bq[tau - t_lo][r], accJ[tt]);
                };
                wait_flag(par * NTILE + TG - 1, sc);   // tiles are published in order: the last one covers the group
                NCDE_TICK(1)
                dxl_group(ic0{}, std::integral_constant<int, TG>{});
                if (wprev != 0.0f) dwo_range(par ^ 1, wprev, std::integral_constant<int, NT2 + 2 * ((NTILE - NT2) / 4)>{}, ic_all{});
                NCDE_TICK(2)
                wait_flag(par * NTILE + NTILE - 1, sc);
                NCDE_TICK(3)
                dxl_group(std::integral_constant<int, TG>{}, ic_all{});
#pragma unroll
                for (int tt = 0; tt < HT; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[pw * HH * 16 + (4 * (4 * tt + r) + g) * 16 + s] = accJ[tt][r];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                NCDE_TICK(4)
                __syncthreads();  // barrier A
                // (3) first blocks of this stage's dWo: runs under the chain wave's hidden-layer backward
                if (wq != 0.0f) dwo_range(par, wq, ic0{}, ic_half{});
                __syncthreads();  // barrier B
                NCDE_TICK(5)
                wprev = wq;
            }
        }
        if (wprev != 0.0f) {
            const int par = sc & 1;
            dwo_range(par, wprev, ic_half{}, ic_all{});
            dw_hidden(par, wprev);
        }
        if constexpr (PROF != 0) {
            if (lane == 0) {
                unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.out) + ((long long)blockIdx.x * 8 + wave) * 6;
                for (int k = 0; k < 6; ++k) dst[k] = prof[k];
            }
        }
        // ---- write-out of this workgroup's parameter-gradient partial ------------------------------------------
        float* gp = a.gpart + (long long)blockIdx.x * a.theta_size;
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) {
            // D layout of the 32x32 block: col = lane&31 (hidden unit j), row = (q&3) + 8*(q>>2) + 4*(lane>>5)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int tau = 2 * blk + (q >> 3);                      // row >> 4
                const int gr = 2 * ((q >> 2) & 1) + (lane >> 5), rr = q & 3;  // in-tile row = 4*gr + rr
                const int cq = tau / NB, nb = tau - cq * NB;
                const int h = 4 * (pw * NB + nb) + gr, c = 4 * cq + rr;
                if (c < C) gp[a.gWo_off + (h * C + c) * HH + (lane & 31)] = gWo[blk][q];
            }
            // bias gradient: lane (i32, kg) holds the partial sum of block row i32 over samples 8kg..8kg+7
            float v = gbo[blk];
            v += __shfl_xor(v, 32, 64);
            const int i32 = lane & 31;
            const int tau = 2 * blk + (i32 >> 4), rowt = i32 & 15;
            const int cq = tau / NB, nb = tau - cq * NB;
            const int hrow = 4 * (pw * NB + nb) + (rowt >> 2), crow = 4 * cq + (rowt & 3);
            if (lane < 32 && crow < C) gp[a.gbo_off + hrow * C + crow] = v;
        }
#pragma unroll
        for (int k = 0; k < TPW1; ++k) {
            const int id = pw * TPW1 + k, tr = id / HT, tc = id - tr * HT;
#pragma unroll
            for (int r = 0; r < 4; ++r) gp[a.gW_off[1] + (16 * tr + 4 * g + r) * HH + 16 * tc + s] = gW1[k][r];
            float v = gb1[k];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (tc == 0 && g == 0) gp[a.gb_off[1] + 16 * tr + s] = v;
        }
#pragma unroll
        for (int k = 0; k < TPW0; ++k) {
            const int id = pw * TPW0 + k, tr = id / HT0, tc = id - tr * HT0;
#pragma unroll
            for (int r = 0; r < 4; ++r) gp[a.gW_off[0] + (16 * tr + 4 * g + r) * H + 16 * tc + s] = gW0[k][r];
            float v = gb0[k];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (tc == 0 && g == 0) gp[a.gb_off[0] + 16 * tr + s] = v;
        }
    }
    (void)n_stage_total;
#undef NCDE_TICK
}

// ------------------------------------------------------------------------------------------------
// adjoint v3: chain + gradient waves, chain GEMMs as split-bf16 MFMA (H = HH = 32)
// ------------------------------------------------------------------------------------------------
// Same role split as ncde_adj_fast2.  Differences:
//   * every GEMM on the stage's dependency chain (forward recompute, output tiles, hidden-layer backward,
//     a^T df/dy) runs as 3-way split-bf16 v_mfma_f32_16x16x32_bf16 (fp32-equivalent, see ncde_fwd_fast_bf3):
//     one K chunk = all 32 hidden units, lane (s, g) <-> k = 8g + j, layer outputs permuted as
//     tile t, D-row (g, r) <-> unit 8g + 4t + r so D registers feed the next layer's split directly;
//   * dP tiles are single-buffered and numbered in publication order; the gradient wave does the dL/dx_L
//     partial per tile group right behind the chain wave and this stage's dWo blocks in its shadow; only the
//     hidden-layer dW/db lag one stage (x images double-buffered by stage parity);
//   * the lo pieces of the output-layer weights and the split W1^T / W0^T A-operands live in LDS images.
// HP selects the split arithmetic (DESIGN.md sections 5.2b, 5.4a, 5.4c):
//   0  everything 3-way split-bf16; dL/dx_L as fp32 MFMA (round-2 kernel; also the instantiation that re-executes range-faulted
//      tiles of the other two: `only_faulted`)
//   2  the DEFAULT: forward-side GEMMs of the chain waves (recompute, output tiles) 2-way split-fp16; dL/dx_L = Wo^T dP as split-bf16
//      with the lo pieces of Wo^T in the LDS region the chain waves' Wo lo pieces no longer need; all dWo blocks behind barrier A
//   1  experimental (NCDE_FLAG_ADJOINT_SPLIT_FP16): everything split-fp16, cotangents normalised by a per-workgroup power of two
// PLAN = 1 (round 4, continuous adjoint only): the general time axis -- the reverse steps of the adjoint table of a.plan (one reverse
// solve per output interval: y reset to the stored value and dL/dz added where the table says so), per-step dt in the Butcher
// bookkeeping and in the quadrature weights of the gradient waves, dX/dt staged per stage of the next step as in ncde_fwd_fast_bf3.
template <int NL, int C, int INTERP, int METHOD, int PROF = 0, int DISC = 0, int HP = 0, int PLAN = 0>
__global__ __launch_bounds__(512, 2) void ncde_adj_fast3(KArgs a) {
    unsigned long long prof[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;
#define NCDE_TICK(k)                                                \
    if constexpr (PROF != 0) {                                      \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        prof[k] += now_ - tlast;                                    \
        tlast = now_;                                               \
    }
    constexpr int H = 32, HH = 32, NW = 4, HT = 2;
    constexpr int CP = (C + 3) & ~3, CQ = CP / 4, NB = H / 4 / NW;  // NB = 2 h-blocks per pair
    constexpr int S = kStages<METHOD>;
    constexpr int NT = 64 * NW;
    constexpr int DXW = INTERP == NCDE_INTERP_LINEAR ? CP : 3 * CP;
    constexpr int EPT = (16 * DXW + NT - 1) / NT;
    constexpr int NTILE = NB * CQ, NBLK = NTILE / 2;   // tile tau = cq*NB + nb (publication order); block = one cq
    constexpr int XROWS = H + NL * HH;
    constexpr int NFLAG = NTILE + 2;
    // HP = 1: 2-way split-fp16 GEMMs (ncde_bf3.h) -- the forward-side operands as they are, the cotangent-side ones in units of a
    // per-workgroup power of two `sig` that follows max |a| over the tile from step to step; the workgroup reports a range fault
    // (a.fault) when any operand left the fp16 range, and the HP = 0 instance re-executes exactly those workgroups (only_faulted).
    // HP = 2 (default): only the FORWARD-side GEMMs of the chain waves (stage recompute, output tiles: operands z, x_l, W -- O(1)
    // magnitudes) are split-fp16; everything that carries the cotangent, and the whole gradient wave, stays as in HP = 0.
    // Nothing changes in what the two roles exchange.  HP = 1 (experimental, NCDE_FLAG_ADJOINT_SPLIT_FP16): everything split-fp16.
    constexpr int HPF = HP != 0 ? 1 : 0, HPC = HP == 1 ? 1 : 0;
    typedef SplitOps<HPF> SF;       // forward-side operands
    typedef SplitOps<HPC> SO;       // cotangent-side operands, gradient waves
    typedef typename SF::T SpF;
    typedef typename SO::T SpT;
    constexpr int NPF = SF::NP, NP = SO::NP;
    if constexpr (HP == 0) {
        if (a.only_faulted && a.fault[blockIdx.x] == 0) return;
    }
    static_assert(NB == 2 && NTILE % 2 == 0, "one 32-row dWo block per cq");
    static_assert(PLAN == 0 || DISC == 0, "the planned discrete backward runs on the batch-tiled family");
    if constexpr (PLAN != 0) {
        if (a.plan == nullptr || !plan_header_ok(a, S)) return;      // (uniform: before the first barrier)
    }
    // PLAN: the dxs region holds [2][S][16][CP] (dX/dt of every stage of a reverse step, by step parity) instead of the ring of pieces
    const int pw_ = plan_step_words(S);
    const int* padj = PLAN ? a.plan + plan_off_adj(S, a.n_steps_fwd, a.n_out) : nullptr;
    const int n_rsteps = PLAN ? a.n_steps_adj : a.T - 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* zx = lds;                                   // [2][H*16]
    float* dxs = zx + 2 * H * 16;                      // [3][16*DXW]  (PLAN: [2][S][16][CP])
    constexpr int DXR = PLAN ? 2 * S * 16 * CP : 3 * 16 * DXW;
    float* red = dxs + DXR;                            // [NW][HH*16]
    float* boL = red + NW * HH * 16;                   // [NW][NB*CQ][4 g][4 r]   (nb-major)
    float* tiles = boL + NW * NTILE * 16;              // [NW][NTILE][16][16]     raw dP of the current stage
    float* ximg = tiles + NW * NTILE * 256;            // [2][XROWS][16]          by stage parity (chain wave 0)
    float* dpimg = ximg + 2 * XROWS * 16;              // [NDP][NL][HH][16]       (chain wave 0); NDP = 2 (by stage parity) when the
                                                       // gradient waves consume it behind barrier A of the NEXT stage (HP = 1)
    // (HP = 2: only where the second image fits -- the cubic control path stages three coefficient rows per piece and is at the LDS limit)
    constexpr int NDP = ((HP == 1 && NCDE_H2_DW_LATE != 0) || (HP == 2 && NCDE_F2_DW_LATE != 0 && INTERP == NCDE_INTERP_LINEAR)) ? 2 : 1;
    constexpr bool DXL3 = HP == 2 && NCDE_F2_DXL_BF3 != 0;
    constexpr bool DWO0_LATE = (HP == 1 && NCDE_H2_DWO_EARLY == 0) || (HP == 2 && NCDE_F2_DWO_EARLY == 0);   // block 0 behind barrier A too
    int* flags = reinterpret_cast<int*>(dpimg + NDP * NL * HH * 16);  // [NW][NFLAG]
    float* biasL = reinterpret_cast<float*>(flags + NW * NFLAG);      // [2][HT][4 g][4 r]: b[8g + 4t + r]
    unsigned* w1T3 = reinterpret_cast<unsigned*>(biasL + 2 * HT * 16);  // [HT][NP][64][4]  split W1^T A operands
    unsigned* w0T3 = w1T3 + HT * NP * 256;                            // [NW][NP][64][4]   split W0^T (own state rows)
    unsigned* woLo = w0T3 + NW * NP * 256;                            // [NW][NB][CQ][64][4] lo pieces: of the chain waves' Wo A operands
                                                                      // (HP = 0) / of the gradient waves' Wo^T A operands (HP = 1)
    unsigned* w1S3 = woLo + ((HP == 2 && !DXL3) ? 0 : NW * NB * CQ * 256);       // [2 layers][HT][NPF][64][4]  split W0 / W1 (forward) A operands
                                                                      // (HP = 2 has no lo-piece image: nobody reads one)
    float* amax = reinterpret_cast<float*>(w1S3 + 2 * HT * NPF * 256); // [NW] max |a| of each chain wave's state rows; [NW] = fault word
    int* fault_s = reinterpret_cast<int*>(amax + NW);
    float mx = 0.0f;              // largest operand magnitude the split-fp16 GEMMs have seen (ncde_bf3.h)
    float sig = 1.0f;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_chain = wave < NW;
    const int pw = is_chain ? wave : wave - NW;
    const int s = lane & 15, g = lane >> 4;
    const int b0 = blockIdx.x * NCDE_TILE;
    const int bs = b0 + s;
    const bool valid = bs < a.B;
    float* my_tiles = tiles + pw * NTILE * 256;
    volatile __attribute__((address_space(3))) int* my_flags =
        (volatile __attribute__((address_space(3))) int*)(flags + pw * NFLAG);
    volatile __attribute__((address_space(3))) int* xflag = (volatile __attribute__((address_space(3))) int*)(flags + NTILE);
    const float* boLw = boL + pw * NTILE * 16;

    for (int e = tid; e < NW * NFLAG; e += 512) flags[e] = 0;
    if (tid == 0) *fault_s = 0;
    for (int e = tid; e < NW * NTILE * 16; e += 512) {
        const int r = e & 3, gg = (e >> 2) & 3, rest = e >> 4;
        const int t2 = rest % NTILE, wv = rest / NTILE;
        const int nb = t2 / CQ, cq = t2 - nb * CQ;
        const int h = 4 * (wv * NB + nb) + gg, c = 4 * cq + r;
        boL[e] = c < C ? NCDE_TANH_PRESCALE * a.bo[h * C + c] : 0.0f;
    }
    for (int e = tid; e < 2 * HT * 16; e += 512) {
        const int r = e & 3, gg = (e >> 2) & 3, t = (e >> 4) % HT, layer = e / (16 * HT);
        biasL[e] = (layer == 0 || NL > 1) ? a.b[layer][8 * gg + 4 * t + r] : 0.0f;      // NL = 1: no inner layer, no second (W, b)
    }
    if (tid < 64 * HT) {  // split W1^T: A row i <-> output unit 8(i>>2)+4t+(i&3), k = 8kg + jj
        const int l = tid & 63, t = tid >> 6;
        const int unit_out = 8 * ((l & 15) >> 2) + 4 * t + (l & 3);
        float tmp[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) tmp[jj] = NL > 1 ? a.W[1][(8 * (l >> 4) + jj) * HH + unit_out] : 0.0f;
        SO::store(w1T3 + t * NP * 256, l, SO::split(tmp, mx));
    } else if (tid >= 64 * HT + 64 * NW && tid < 64 * HT + 64 * NW + 64 * HT) {  // split W0 and W1 (forward), shared by all chain waves
        const int l = tid & 63, t = (tid >> 6) - HT - NW;
        const int unitA = 8 * ((l & 15) >> 2) + 4 * t + (l & 3);
#pragma unroll
        for (int layer = 0; layer < 2; ++layer) {
            float tmp[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) tmp[jj] = (layer == 0 || NL > 1) ? a.W[layer][unitA * HH + 8 * (l >> 4) + jj] : 0.0f;   // H == HH
            SF::store(w1S3 + (layer * HT + t) * NPF * 256, l, SF::split(tmp, mx));
        }
    } else if (tid < 64 * HT + 64 * NW) {  // split W0^T rows of the state entries pair wv owns
        const int l = tid & 63, wv = (tid >> 6) - HT;
        const int r_own = l & 3;
        const int hrow = 4 * (wv * NB + r_own) + ((l & 15) >> 2);
        float tmp[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) tmp[jj] = r_own < NB ? a.W[0][(8 * (l >> 4) + jj) * H + hrow] : 0.0f;
        SO::store(w0T3 + wv * NP * 256, l, SO::split(tmp, mx));
    }

    if (is_chain) {
        // =================================================================================================
        // chain wave
        // =================================================================================================
        u32x4 woHi[NB][CQ], woMid[NB][CQ];      // HP = 1: (hi, lo), both in registers
        unsigned* my_woLo = woLo + pw * NB * CQ * 256;
        auto fwd_weights = [&](int layer, int tt) { return SF::load(w1S3 + (layer * HT + tt) * NPF * 256, lane); };
        auto wo_operand = [&](int nb, int cq) {
            SpF As;
            As.hi = woHi[nb][cq];
            if constexpr (HPF != 0) {
                As.lo = woMid[nb][cq];
            } else {
                As.mid = woMid[nb][cq];
                As.lo = *reinterpret_cast<const u32x4*>(my_woLo + ((nb * CQ + cq) * 64 + lane) * 4);
            }
            return As;
        };
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int cq = 0; cq < CQ; ++cq) {
                const int hA = 4 * (pw * NB + nb) + (s >> 2), cA = 4 * cq + (s & 3);
                float tmp[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) tmp[jj] = cA < C ? NCDE_TANH_PRESCALE * a.Wo[(hA * C + cA) * HH + 8 * g + jj] : 0.0f;
                const SpF sp = SF::split(tmp, mx);
                woHi[nb][cq] = sp.hi;
                if constexpr (HPF != 0) {
                    woMid[nb][cq] = sp.lo;
                } else {
                    woMid[nb][cq] = sp.mid;
                    *reinterpret_cast<u32x4*>(my_woLo + ((nb * CQ + cq) * 64 + lane) * 4) = sp.lo;
                }
            }
        // control-path staging (reverse order), by the 256 chain threads
        const float* eptr[EPT];
        float eprev[EPT], enext[EPT];
        bool eok[EPT];
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = tid + q * NT;
            const int es = e / DXW, ec = e - es * DXW;
            const int part = ec / CP, c = ec - part * CP;
            eok[q] = e < 16 * DXW && c < a.Cc && (b0 + es) < a.B;      // a.Cc: channels of the coefficient tensor (= C unless zero-padded)
            const long long base = (long long)(eok[q] ? b0 + es : 0) * a.cs_b;
            eptr[q] = a.coeffs + base + (INTERP == NCDE_INTERP_LINEAR ? c : (part + 1) * a.Cc + c);
            eprev[q] = 0.0f;
            enext[q] = 0.0f;
        }
        auto stage_load = [&](int piece) {
#pragma unroll
            for (int q = 0; q < EPT; ++q) enext[q] = eok[q] ? eptr[q][(long long)piece * a.cs_t] : 0.0f;
        };
        auto stage_store = [&](int piece) {
            float* dst = dxs + (piece % 3) * 16 * DXW;
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
                const int e = tid + q * NT;
                if (e < 16 * DXW) dst[e] = INTERP == NCDE_INTERP_LINEAR ? eprev[q] - enext[q] : enext[q];
                eprev[q] = enext[q];
            }
        };
        constexpr int EPQ = PLAN ? (S * 16 * CP + NT - 1) / NT : 1;
        float qn[EPQ];
        auto plan_load = [&](const int* pstep) {      // element e = (stage j, sample es, channel c)
#pragma unroll
            for (int q = 0; q < EPQ; ++q) {
                const int e = tid + q * NT;
                const int j = e / (16 * CP), rem = e - j * (16 * CP), es = rem / CP, c = rem - es * CP;
                float v = 0.0f;
                if (e < S * 16 * CP && c < a.Cc && b0 + es < a.B) {
                    const StageDesc sd = plan_stage(pstep, j);
                    const float* p = a.coeffs + (long long)(b0 + es) * a.cs_b + (long long)sd.idx * a.cs_t;
                    if constexpr (INTERP == NCDE_INTERP_LINEAR) {
                        v = p[a.cs_t + c] - p[c];
                        if (sd.kdt != 1.0f) v = v / sd.kdt;
                    } else {
                        const float bb = p[a.Cc + c], cc = p[2 * a.Cc + c], dd = p[3 * a.Cc + c];
                        const float inner = cc + dd * sd.frac;
                        v = bb + inner * sd.frac;
                    }
                }
                qn[q] = v;
            }
        };
        auto plan_store = [&](int buf) {
#pragma unroll
            for (int q = 0; q < EPQ; ++q) {
                const int e = tid + q * NT;
                if (e < S * 16 * CP) dxs[buf * (S * 16 * CP) + e] = qn[q];
            }
        };
        const int p_hi = a.n_pieces - 1;
        if constexpr (PLAN != 0) {
            plan_load(padj);
            plan_store(0);
        } else {
            if (INTERP == NCDE_INTERP_LINEAR) {
#pragma unroll
                for (int q = 0; q < EPT; ++q) eprev[q] = eok[q] ? eptr[q][(long long)(p_hi + 1) * a.cs_t] : 0.0f;
            }
            stage_load(p_hi);
            stage_store(p_hi);
            if (p_hi >= 1) {
                stage_load(p_hi - 1);
                stage_store(p_hi - 1);
            }
        }
        const int last_row = a.n_out - 1;
        float y0[NB], ky1[NB], ky2[NB], a0[NB], ka1[NB], ka2[NB], as_[NB], zreg[8];
        // DISC (exact discrete backward): the stage inputs come from the forward's stage record, [(n*S + j)][B][H],
        // walked backwards (linear index `lin`), fetched one stage ahead; ka1/ka2/ky1 hold dL/dY of stages 4/3/2.
        f32x4 znext[2];
        auto rec_fetch = [&](int lin) {
            const float* rp = a.stages + ((long long)lin * a.B + (valid ? bs : 0)) * a.Hr + 8 * g;
            if (a.Hr == H) {
                znext[0] = *reinterpret_cast<const f32x4*>(rp);
                znext[1] = *reinterpret_cast<const f32x4*>(rp + 4);
            } else {      // zero-padded problem: rows of the caller's record are a.Hr wide (units >= a.Hr are 0)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) znext[jj >> 2][jj & 3] = 8 * g + jj < a.Hr ? rp[jj] : 0.0f;
            }
        };
        if constexpr (DISC != 0) {
            rec_fetch((a.T - 1) * S - 1);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) zreg[jj] = valid ? znext[jj >> 2][jj & 3] : 0.0f;
        } else {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) zreg[jj] = (valid && 8 * g + jj < a.Hr) ? a.z_out[((long long)bs * a.n_out + last_row) * a.Hr + 8 * g + jj] : 0.0f;
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const long long o = ((long long)bs * a.n_out + last_row) * a.Hr + 4 * (pw * NB + nb) + g;
            const bool live = valid && 4 * (pw * NB + nb) + g < a.Hr;      // a.Hr: row width of z_out / grad_out / grad_z0 (= H unless zero-padded)
            y0[nb] = (DISC == 0 && live) ? a.z_out[o] : 0.0f;
            a0[nb] = live ? a.grad_out[o] : 0.0f;
            as_[nb] = (DISC != 0 && METHOD == NCDE_RK4_38) ? a0[nb] * 0.125f : a0[nb];
            ky1[nb] = ky2[nb] = ka1[nb] = ka2[nb] = 0.0f;
        }
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        // max |a| over this wave's state rows -> amax[pw] (read by every wave behind the next barrier)
        auto publish_amax = [&](const float* av) {
            if constexpr (HPC != 0) {
                float m = __builtin_fmaxf(__builtin_fabsf(av[0]), __builtin_fabsf(av[1]));
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) m = __builtin_fmaxf(m, __shfl_xor(m, off, 64));
                if (lane == 0) amax[pw] = m;
            }
        };
        static_assert(NB == 2, "publish_amax reads two entries");
        publish_amax(a0);
        __syncthreads();
        if constexpr (HPC != 0) sig = h2_pick_scale(__builtin_fmaxf(__builtin_fmaxf(amax[0], amax[1]), __builtin_fmaxf(amax[2], amax[3])), 1.0f);
        float isig = h2_inv_scale(sig);

        int zpar = 0, sc = 0;
        if constexpr (PROF != 0) tlast = __builtin_readcyclecounter();
        for (int n = n_rsteps; n >= 1; --n) {
            const int rs = n_rsteps - n;      // PLAN: index of this reverse step in the adjoint table
            const int* pstep = PLAN ? padj + rs * pw_ : nullptr;
            const float dt = PLAN ? __int_as_float(pstep[0]) : 1.0f;
            const int reset_row = PLAN ? pstep[1] : -1;
            if constexpr (PLAN != 0) {
                if (n > 1) plan_load(pstep + pw_);
            } else {
                if (n - 3 >= 0) stage_load(n - 3);
            }
#pragma unroll 1
            for (int j = 0; j < S; ++j) {
                ++sc;
                const int par = sc & 1;
                const float t = DISC != 0 ? (float)(n - 1) + stage_offset(METHOD, S - 1 - j) : -(-(float)n + stage_offset(METHOD, j));
                const int idx = PLAN ? 0 : piece_index(t, a.n_pieces);
                const float frac = t - (float)idx;
                const float wq = DISC != 0 ? 1.0f : stage_weight(METHOD, j);
                if constexpr (DISC != 0) {
                    const int lin = (n - 1) * S + (S - 1 - j);
                    if (lin >= 1) rec_fetch(lin - 1);
                }
                const float* dxp = PLAN ? dxs + (rs & 1) * (S * 16 * CP) + (j * 16 + s) * CP : dxs + (idx % 3) * 16 * DXW + s * DXW;
                // ---- forward recompute (split-bf16); x[l][4t+r] <-> unit 8g + 4t + r -----------------------------------
                float x[NL][8];
                SpF xb;
                {
                    typename SF::Acc acc[HT];
                    xb = SF::split(zreg, mx);
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) {
                        acc[tt] = SF::init(*reinterpret_cast<const f32x4*>(biasL + (tt * 4 + g) * 4));
                        SF::mac(fwd_weights(0, tt), xb, acc[tt]);
                    }
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) {
                        const f32x4 pre = SF::finish(acc[tt]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) x[0][4 * tt + r] = relu_bits(pre[r]);
                    }
#pragma unroll
                    for (int l = 1; l < NL; ++l) {
                        xb = SF::split(x[l - 1], mx);
#pragma unroll
                        for (int tt = 0; tt < HT; ++tt) {
                            acc[tt] = SF::init(*reinterpret_cast<const f32x4*>(biasL + ((HT + tt) * 4 + g) * 4));
                            SF::mac(fwd_weights(1, tt), xb, acc[tt]);
                        }
#pragma unroll
                        for (int tt = 0; tt < HT; ++tt) {
                            const f32x4 pre = SF::finish(acc[tt]);
#pragma unroll
                            for (int r = 0; r < 4; ++r) x[l][4 * tt + r] = relu_bits(pre[r]);
                        }
                    }
                    xb = SF::split(x[NL - 1], mx);
                }
                NCDE_TICK(0)
                if (wq != 0.0f) {  // [unit][sample] images for the gradient waves: the chain waves hold identical copies, wave pw
                                   // writes image pw (stage input, x_1 .. x_NL) -- NL + 1 <= 4 images, one per wave
                    float* xi = ximg + par * XROWS * 16;
                    if (pw == 0) {
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) xi[(8 * g + jj) * 16 + s] = zreg[jj];
                    }
#pragma unroll
                    for (int l = 0; l < NL; ++l)
                        if (pw == (l + 1) % NW) {
#pragma unroll
                            for (int jj = 0; jj < 8; ++jj) xi[(H + l * HH + 8 * g + jj) * 16 + s] = x[l][jj];
                            if (l == NL - 1) {      // x_L is what the gradient waves' dWo blocks of THIS stage wait for
                                wave_lds_order();
                                *xflag = sc;
                            }
                        }
                }
                // From here on the activations are needed only as ReLU masks (x_L's B operand `xb` is already split): one bit each
                // instead of 8 NL registers carried across the output tiles (round 4: the kernel's last 32 B / lane of scratch)
                unsigned relu_mask = 0;
#pragma unroll
                for (int l = 0; l < NL; ++l)
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) relu_mask |= (x[l][jj] > 0.0f ? 1u : 0u) << (8 * l + jj);
                // ---- output tiles: P, r = 1/(exp(2P)+1), f, dP -> LDS tile + flag -----------------------------------
                float kout[NB];
                float sdx = 0.0f;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) kout[nb] = 0.0f;
#pragma unroll
                for (int cq = 0; cq < CQ; ++cq) {
                    f32x4 o[NB];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        typename SF::Acc oa = SF::init(*reinterpret_cast<const f32x4*>(boLw + ((nb * CQ + cq) * 4 + g) * 4));
                        SF::mac(wo_operand(nb, cq), xb, oa);
                        o[nb] = SF::finish(oa);
                    }
                    f32x4 dx;
                    if constexpr (INTERP == NCDE_INTERP_LINEAR || PLAN != 0) {
                        dx = *reinterpret_cast<const f32x4*>(dxp + 4 * cq);
                    } else {
                        const f32x4 cb = *reinterpret_cast<const f32x4*>(dxp + 4 * cq);
                        const f32x4 cc = *reinterpret_cast<const f32x4*>(dxp + CP + 4 * cq);
                        const f32x4 cd = *reinterpret_cast<const f32x4*>(dxp + 2 * CP + 4 * cq);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float inner = cc[r] + cd[r] * frac;
                            dx[r] = cb[r] + inner * frac;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) sdx += dx[r];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const int tau = cq * NB + nb;
                        float* tl = my_tiles + tau * 256;
                        const float a4 = (4.0f * sig) * as_[nb];     // dP, and everything downstream of it, in units of sig
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float rr = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(o[nb][r]) + 1.0f);
                            if constexpr (DISC == 0) kout[nb] = fmaf(rr, dx[r], kout[nb]);
                            tl[(4 * g + r) * 16 + s] = (a4 * dx[r]) * fmaf(-rr, rr, rr);
                        }
                        if (nb == NB - 1) {      // one publication per block (= cq, both tiles): the gradient wave polls odd tiles only
                            wave_lds_order();
                            my_flags[tau] = sc;
                        }
                    }
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) kout[nb] = fmaf(-2.0f, kout[nb], sdx);
                NCDE_TICK(1)
                __syncthreads();  // barrier A: the gradient waves have published their dL/dx_L partials
                NCDE_TICK(2)
                float gpre[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    float v = red[(8 * g + jj) * 16 + s];
#pragma unroll
                    for (int wv = 1; wv < NW; ++wv) v += red[wv * HH * 16 + (8 * g + jj) * 16 + s];
                    gpre[jj] = ((relu_mask >> (8 * (NL - 1) + jj)) & 1u) ? v : 0.0f;
                }
                // ---- hidden layers backward (split-bf16) -----------------------------------------------------------------
#pragma unroll
                for (int l = NL - 1; l >= 1; --l) {
                    if (wq != 0.0f && pw == l % NW) {      // one image per chain wave (all hold the same gpre)
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) dpimg[(NDP - 1) * par * NL * HH * 16 + (l * HH + 8 * g + jj) * 16 + s] = gpre[jj];
                    }
                    const SpT gb = SO::split(gpre, mx);
                    typename SO::Acc acc[HT];
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) {
                        acc[tt] = SO::init(zero4);
                        SO::mac(SO::load(w1T3 + tt * NP * 256, lane), gb, acc[tt]);
                    }
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) {
                        const f32x4 gq = SO::finish(acc[tt]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) gpre[4 * tt + r] = ((relu_mask >> (8 * (l - 1) + 4 * tt + r)) & 1u) ? gq[r] : 0.0f;
                    }
                }
                if (wq != 0.0f && pw == 0) {
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) dpimg[(NDP - 1) * par * NL * HH * 16 + (8 * g + jj) * 16 + s] = gpre[jj];
                }
                f32x4 vy;
                {
                    const SpT gb = SO::split(gpre, mx);
                    typename SO::Acc va = SO::init(zero4);
                    SO::mac(SO::load(w0T3 + pw * NP * 256, lane), gb, va);
                    vy = SO::finish(va);
                    if constexpr (HPC != 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) vy[r] *= isig;
                    }
                }
                NCDE_TICK(3)
                float ys[NB];
                if constexpr (DISC != 0) {
                    // transpose of the Butcher step (RK4 3/8: c4 = a/8; c3 = 3c4 + d4; c2 = 3c4 - d4 + d3;
                    // c1 = c4 + d4 - d3/3 + d2/3; a += d4 + d3 + d2 + d1), d = vy = dL/dY of this stage
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const float d = vy[nb];
                        ys[nb] = 0.0f;
                        if constexpr (METHOD == NCDE_RK4_38) {
                            const float c4 = a0[nb] * 0.125f;
                            if (j == 0) { ka1[nb] = d; as_[nb] = 3.0f * c4 + d; }
                            else if (j == 1) { ka2[nb] = d; as_[nb] = (3.0f * c4 - ka1[nb]) + d; }
                            else if (j == 2) { ky1[nb] = d; as_[nb] = ((c4 + ka1[nb]) - 0.333333343267440796f * ka2[nb]) + 0.333333343267440796f * d; }
                            else { a0[nb] = (((a0[nb] + ka1[nb]) + ka2[nb]) + ky1[nb]) + d; }
                        } else if constexpr (METHOD == NCDE_MIDPOINT) {
                            if (j == 0) { ka1[nb] = d; as_[nb] = 0.5f * d; }
                            else { a0[nb] = (a0[nb] + ka1[nb]) + d; }
                        } else {
                            a0[nb] = a0[nb] + d;
                        }
                    }
                    if (j == S - 1) {
                        if (a.output == NCDE_OUT_KNOTS || n == 1) {
                            const int row = a.output == NCDE_OUT_KNOTS ? n - 1 : 0;
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
                                a0[nb] += (valid && 4 * (pw * NB + nb) + g < a.Hr) ? a.grad_out[((long long)bs * a.n_out + row) * a.Hr + 4 * (pw * NB + nb) + g] : 0.0f;
                        }
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) as_[nb] = METHOD == NCDE_RK4_38 ? a0[nb] * 0.125f : a0[nb];
                        if (n - 3 >= 0) stage_store(n - 3);
                        publish_amax(a0);
                    }
                    __syncthreads();  // barrier B
                    if constexpr (HPC != 0) {
                        if (j == S - 1) {
                            sig = h2_pick_scale(__builtin_fmaxf(__builtin_fmaxf(amax[0], amax[1]), __builtin_fmaxf(amax[2], amax[3])), sig);
                            isig = h2_inv_scale(sig);
                        }
                    }
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) zreg[jj] = valid ? znext[jj >> 2][jj & 3] : 0.0f;
                    NCDE_TICK(4)
                    continue;
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    if constexpr (PLAN != 0) {
                        bool last;
                        ys[nb] = StageCombine::apply(METHOD, j, -kout[nb], dt, y0[nb], ky1[nb], ky2[nb], last);
                        as_[nb] = StageCombine::apply(METHOD, j, vy[nb], dt, a0[nb], ka1[nb], ka2[nb], last);
                    } else {
                        ys[nb] = Combine<METHOD>::apply(j, -kout[nb], y0[nb], ky1[nb], ky2[nb]);
                        as_[nb] = Combine<METHOD>::apply(j, vy[nb], a0[nb], ka1[nb], ka2[nb]);
                    }
                }
                const bool plan_reset = PLAN != 0 && j == S - 1 && reset_row >= 0;      // end of an output interval (adjoint.py:116-133)
                if constexpr (PLAN != 0) {
                    if (plan_reset) {
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            const long long o = ((long long)bs * a.n_out + reset_row) * a.Hr + 4 * (pw * NB + nb) + g;
                            const bool live = valid && 4 * (pw * NB + nb) + g < a.Hr;
                            y0[nb] = live ? a.z_out[o] : 0.0f;
                            ys[nb] = y0[nb];
                            a0[nb] += live ? a.grad_out[o] : 0.0f;
                            as_[nb] = a0[nb];
                        }
                    }
                    if (j == S - 1) {
                        if (n > 1) plan_store((rs + 1) & 1);
                        publish_amax(as_);
                    }
                }
                if (PLAN == 0 && j == S - 1) {
                    if (a.output == NCDE_OUT_KNOTS) {  // reset y to the stored knot value, add dL/dz of that knot
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            const long long o = ((long long)bs * a.n_out + (n - 1)) * a.Hr + 4 * (pw * NB + nb) + g;
                            const bool live = valid && 4 * (pw * NB + nb) + g < a.Hr;
                            y0[nb] = live ? a.z_out[o] : 0.0f;
                            ys[nb] = y0[nb];
                            a0[nb] += live ? a.grad_out[o] : 0.0f;
                            as_[nb] = a0[nb];
                        }
                    } else if (n == 1) {
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            a0[nb] += (valid && 4 * (pw * NB + nb) + g < a.Hr) ? a.grad_out[((long long)bs * a.n_out) * a.Hr + 4 * (pw * NB + nb) + g] : 0.0f;
                            as_[nb] = a0[nb];
                        }
                    }
                    if (n - 3 >= 0) stage_store(n - 3);
                    publish_amax(as_);      // as_ = a at the step's lower end (+ dL/dz of that knot)
                }
                if (plan_reset || (PLAN == 0 && j == S - 1 && a.output == NCDE_OUT_KNOTS)) {
                    const int zrow = PLAN ? reset_row : n - 1;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) zreg[jj] = (valid && 8 * g + jj < a.Hr) ? a.z_out[((long long)bs * a.n_out + zrow) * a.Hr + 8 * g + jj] : 0.0f;
                    __syncthreads();  // barrier B
                } else {
                    float* zw = zx + zpar * H * 16;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) zw[(4 * (pw * NB + nb) + g) * 16 + s] = ys[nb];
                    __syncthreads();  // barrier B
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) zreg[jj] = zw[(8 * g + jj) * 16 + s];
                    zpar ^= 1;
                }
                if constexpr (HPC != 0) {
                    if (j == S - 1) {
                        sig = h2_pick_scale(__builtin_fmaxf(__builtin_fmaxf(amax[0], amax[1]), __builtin_fmaxf(amax[2], amax[3])), sig);
                        isig = h2_inv_scale(sig);
                    }
                }
                NCDE_TICK(4)
            }
        }
        if constexpr (PROF != 0) {
            if (lane == 0) {
                unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.out) + ((long long)blockIdx.x * 8 + wave) * 6;
                for (int k = 0; k < 6; ++k) dst[k] = prof[k];
            }
        }
        if (valid) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                if (4 * (pw * NB + nb) + g < a.Hr) a.grad_z0[(long long)bs * a.Hr + 4 * (pw * NB + nb) + g] = a0[nb];
        }
    } else {
        // =================================================================================================
        // gradient wave
        // =================================================================================================
        // A operands of the dL/dx_L GEMM, output row i <-> unit 8(i>>2)+4t'+(i&3).  HP = 0: fp32 (v_mfma_f32_16x16x4_f32), one
        // value per (tile, t', r).  HP = 1: split-fp16, K = the 32 rows of a block (= one cq): k = 8 kg + jj <-> tile nb = k >> 4,
        // row 4 g' + r = k & 15 of that tile; hi pieces in 40 registers, lo pieces in this pair's LDS image.
        float woT[(HPC || DXL3) ? 1 : NTILE][HT][4];
        u32x4 woT2h[(HPC || DXL3) ? NBLK : 1][HT];
        u32x4 woT3m[DXL3 ? NBLK : 1][HT];      // DXL3: (hi, mid) of the 3-way bf16 split in registers, lo in LDS
        unsigned* my_woTlo = woLo + pw * NBLK * HT * 256;
        static_assert(NBLK * HT == NB * CQ, "the lo-piece image reuses the chain waves' region");
#pragma unroll
        for (int tau = 0; tau < NTILE; ++tau) {
            const int cq = tau / NB, nb = tau - cq * NB;
            const int h = 4 * (pw * NB + nb) + g;
#pragma unroll
            for (int tp = 0; tp < HT; ++tp) {
                const int jrow = 8 * (s >> 2) + 4 * tp + (s & 3);
                if constexpr (HPC == 0 && !DXL3) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = 4 * cq + r;
                        woT[tau][tp][r] = c < C ? a.Wo[(h * C + c) * HH + jrow] : 0.0f;
                    }
                } else if (nb == 0) {
                    float tmp[8];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const int kk = 8 * g + jj;
                        const int hh = 4 * (pw * NB + (kk >> 4)) + ((kk & 15) >> 2), c = 4 * cq + (kk & 3);
                        tmp[jj] = c < C ? a.Wo[(hh * C + c) * HH + jrow] : 0.0f;
                    }
                    if constexpr (DXL3) {
                        const Split3 sp = split8(tmp);
                        woT2h[cq][tp] = sp.hi;
                        woT3m[cq][tp] = sp.mid;
                        *reinterpret_cast<u32x4*>(my_woTlo + ((cq * HT + tp) * 64 + lane) * 4) = sp.lo;
                    } else {
                        const Split2h sp = split8h(tmp, mx);
                        woT2h[cq][tp] = sp.hi;
                        *reinterpret_cast<u32x4*>(my_woTlo + ((cq * HT + tp) * 64 + lane) * 4) = sp.lo;
                    }
                }
            }
        }
        f32x16 gWo[NBLK];
        f32x4 gW1, gW0;      // one 16x16 tile of dW1 / dW0 per pair: tile (tr, tc) = (pw >> 1, pw & 1)
        float gbo[NBLK], gb1 = 0.0f, gb0 = 0.0f;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        gW1 = zero4;
        gW0 = zero4;
#pragma unroll
        for (int i = 0; i < NBLK; ++i) {
            gbo[i] = 0.0f;
#pragma unroll
            for (int q = 0; q < 16; ++q) gWo[i][q] = 0.0f;
        }
        __syncthreads();
        if constexpr (HPC != 0) sig = h2_pick_scale(__builtin_fmaxf(__builtin_fmaxf(amax[0], amax[1]), __builtin_fmaxf(amax[2], amax[3])), 1.0f);
        float acc_sig = sig;        // the units the gradient accumulators are in

        const int tr = pw >> 1, tc = pw & 1;
        auto wait_flag = [&](int slot, int want) {
            while (__builtin_amdgcn_readfirstlane(my_flags[slot]) != want) __builtin_amdgcn_s_sleep(1);
            wave_lds_order();
        };
        // dWo of 2-tile block `blk` (= one cq): 32 rows x 32 units x 16 samples = 6 split-bf16 32x32x16 MFMAs
        auto dwo_block = [&](const SpT& Bs, float w, int blk) {
            const int i32 = lane & 31, kg = lane >> 5;
            const float* tl = my_tiles + (2 * blk + (i32 >> 4)) * 256 + (i32 & 15) * 16 + 8 * kg;
            const f32x4 a0v = *reinterpret_cast<const f32x4*>(tl);
            const f32x4 a1v = *reinterpret_cast<const f32x4*>(tl + 4);
            float av[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) { av[q] = a0v[q]; av[4 + q] = a1v[q]; }
            gbo[blk] += w * (((av[0] + av[1]) + (av[2] + av[3])) + ((av[4] + av[5]) + (av[6] + av[7])));
            const SpT As = SO::split(av, mx);
            f32x16 c = gWo[blk];
            if constexpr (HPC == 0) {
                c = mfma_bf32(As.lo, Bs.hi, c);
                c = mfma_bf32(As.hi, Bs.lo, c);
                c = mfma_bf32(As.mid, Bs.mid, c);
                c = mfma_bf32(As.mid, Bs.hi, c);
                c = mfma_bf32(As.hi, Bs.mid, c);
                c = mfma_bf32(As.hi, Bs.hi, c);
            } else {
                // the cross products are folded into the accumulator block by block: a second set of 5 x 16 accumulator
                // registers does not fit beside gWo and the W_o^T operands
                f32x16 cx;
#pragma unroll
                for (int q = 0; q < 16; ++q) cx[q] = 0.0f;
                cx = mfma_h32(As.lo, Bs.hi, cx);
                cx = mfma_h32(As.hi, Bs.lo, cx);
                c = mfma_h32(As.hi, Bs.hi, c);
#pragma unroll
                for (int q = 0; q < 16; q += 2) {
                    const f32x2 f = __builtin_elementwise_fma((f32x2){cx[q], cx[q + 1]}, (f32x2){NCDE_H2_INV, NCDE_H2_INV}, (f32x2){c[q], c[q + 1]});
                    c[q] = f[0];
                    c[q + 1] = f[1];
                }
            }
            gWo[blk] = c;
        };
        auto x3_split = [&](int par, float w) {
            const int i32 = lane & 31, kg = lane >> 5;
            const float* xi = ximg + par * XROWS * 16 + (H + (NL - 1) * HH + i32) * 16 + 8 * kg;
            const f32x4 b0v = *reinterpret_cast<const f32x4*>(xi);
            const f32x4 b1v = *reinterpret_cast<const f32x4*>(xi + 4);
            float bv[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) { bv[q] = w * b0v[q]; bv[4 + q] = w * b1v[q]; }
            return SO::split(bv, mx);
        };
        // hidden-layer dW/db of the stage whose x images have parity `par` (fp32 MFMA, samples are K)
        const float* dpimg_all = dpimg;
        auto dw_hidden = [&](int par, float w) {
            const float* xi = ximg + par * XROWS * 16;
            const float* dpimg = dpimg_all + (NDP - 1) * par * NL * HH * 16;
#pragma unroll
            for (int l = NL - 1; l >= 1; --l) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(dpimg + (l * HH + 16 * tr + s) * 16 + 4 * g);
                const f32x4 bv = *reinterpret_cast<const f32x4*>(xi + (H + (l - 1) * HH + 16 * tc + s) * 16 + 4 * g);
                if (tc == 0) gb1 += w * ((av[0] + av[1]) + (av[2] + av[3]));
#pragma unroll
                for (int q = 0; q < 4; ++q) gW1 = mfma16(av[q], w * bv[q], gW1);
            }
            const f32x4 av = *reinterpret_cast<const f32x4*>(dpimg + (16 * tr + s) * 16 + 4 * g);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(xi + (16 * tc + s) * 16 + 4 * g);
            if (tc == 0) gb0 += w * ((av[0] + av[1]) + (av[2] + av[3]));
#pragma unroll
            for (int q = 0; q < 4; ++q) gW0 = mfma16(av[q], w * bv[q], gW0);
        };
        f32x4 accJ[HT], accJx[HT];
        auto dxl_tiles = [&](auto t_lo_c, auto t_hi_c) {
            constexpr int t_lo = decltype(t_lo_c)::value, t_hi = decltype(t_hi_c)::value;
            if constexpr (HPC != 0 || DXL3) {        // one block: B = dP of the 32 rows of this block for sample s, k = 8g + jj
                static_assert(t_hi - t_lo == 2, "a block is two tiles");
                float bv[8];
                const float* tl = my_tiles + (t_lo + (g >> 1)) * 256 + (8 * (g & 1)) * 16 + s;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) bv[jj] = tl[jj * 16];
                if constexpr (DXL3) {
                    const Split3 Bq = split8(bv);
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) {
                        Split3 Aw;
                        Aw.hi = woT2h[t_lo / 2][tt];
                        Aw.mid = woT3m[t_lo / 2][tt];
                        Aw.lo = *reinterpret_cast<const u32x4*>(my_woTlo + (((t_lo / 2) * HT + tt) * 64 + lane) * 4);
                        accJ[tt] = mfma_split(Aw, Bq, accJ[tt]);
                    }
                } else {
                    const Split2h Bq = split8h(bv, mx);
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) {
                        Split2h Aw;
                        Aw.hi = woT2h[t_lo / 2][tt];
                        Aw.lo = *reinterpret_cast<const u32x4*>(my_woTlo + (((t_lo / 2) * HT + tt) * 64 + lane) * 4);
                        mfma_split2(Aw, Bq, accJ[tt], accJx[tt]);
                    }
                }
            } else {
                float bq[t_hi - t_lo][4];
#pragma unroll
                for (int tau = t_lo; tau < t_hi; ++tau)
#pragma unroll
                    for (int r = 0; r < 4; ++r) bq[tau - t_lo][r] = my_tiles[tau * 256 + (4 * g + r) * 16 + s];
#pragma unroll
                for (int tau = t_lo; tau < t_hi; ++tau)
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) accJ[tt] = mfma16(woT[tau][tt][r], bq[tau - t_lo][r], accJ[tt]);
            }
        };
        // the cotangent scale moved at the last step boundary: bring the accumulators along (after the last contribution in the
        // old units -- dw_hidden of the previous stage -- and before the first in the new ones)
        auto rescale_acc = [&]() {
            if constexpr (HPC != 0) {
                if (sig != acc_sig) {
                    const float ratio = sig * h2_inv_scale(acc_sig);
#pragma unroll
                    for (int i = 0; i < NBLK; ++i) {
                        gbo[i] *= ratio;
#pragma unroll
                        for (int q = 0; q < 16; ++q) gWo[i][q] *= ratio;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) { gW1[r] *= ratio; gW0[r] *= ratio; }
                    gb1 *= ratio;
                    gb0 *= ratio;
                    acc_sig = sig;
                }
            }
        };
        int sc = 0;
        float wprev = 0.0f;
        if constexpr (PROF != 0) tlast = __builtin_readcyclecounter();
        for (int n = n_rsteps; n >= 1; --n) {
            const float dtw = PLAN ? __int_as_float(padj[(n_rsteps - n) * pw_]) : 1.0f;      // the step's dt scales its quadrature weights
#pragma unroll 1
            for (int j = 0; j < S; ++j) {
                ++sc;
                const int par = sc & 1;
                const float wq = DISC != 0 ? 1.0f : stage_weight(METHOD, j) * dtw;
                // hidden-layer dW/db of the previous stage, under the chain wave's forward recompute
                if (NDP == 1 && wprev != 0.0f) dw_hidden(par ^ 1, wprev);
                NCDE_TICK(0)
#pragma unroll
                for (int tt = 0; tt < HT; ++tt) accJ[tt] = accJx[tt] = zero4;
                if constexpr (NDP == 1) rescale_acc();
                // Per block (= one cq, two tiles, published together): poll, dL/dx_L (16 fp32 MFMAs, on the stage's
                // critical path).  One dWo block of this stage is slotted in behind block 1 (more would make the wave
                // fall behind the chain wave); only the LAST block's dL/dx_L trails the chain wave into barrier A.
                SpT Bs;
                bool have_bs = false;
#pragma unroll
                for (int blk = 0; blk < NBLK; ++blk) {
                    NCDE_TICK(2)
                    wait_flag(2 * blk + 1, sc);
                    NCDE_TICK(1)
                    if (blk == 0) dxl_tiles(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
                    else if (blk == 1) dxl_tiles(std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
                    else if (blk == 2) { if constexpr (NBLK > 2) dxl_tiles(std::integral_constant<int, 4>{}, std::integral_constant<int, 6>{}); }
                    else if (blk == 3) { if constexpr (NBLK > 3) dxl_tiles(std::integral_constant<int, 6>{}, std::integral_constant<int, 8>{}); }
                    else if (blk == 4) { if constexpr (NBLK > 4) dxl_tiles(std::integral_constant<int, 8>{}, std::integral_constant<int, 10>{}); }
                    static_assert(NBLK <= 5, "extend the block dispatch");
                    if (wq != 0.0f && blk == 1 && !DWO0_LATE) {
                        if (!have_bs) {
                            while (__builtin_amdgcn_readfirstlane(*xflag) != sc) __builtin_amdgcn_s_sleep(1);
                            wave_lds_order();
                            Bs = x3_split(par, wq);
                            have_bs = true;
                        }
                        dwo_block(Bs, wq, blk >> 1);
                    }
                }
                NCDE_TICK(3)
#pragma unroll
                for (int tt = 0; tt < HT; ++tt) {
                    if constexpr (HPC != 0) accJ[tt] = h2_combine(accJ[tt], accJx[tt]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[pw * HH * 16 + (8 * g + 4 * tt + r) * 16 + s] = accJ[tt][r];
                }
                NCDE_TICK(4)
                __syncthreads();  // barrier A
                if constexpr (NDP == 2) {   // under the chain waves' hidden-layer backward
                    if (wprev != 0.0f) dw_hidden(par ^ 1, wprev);
                    rescale_acc();
                }
                if (wq != 0.0f) {
                    if (!have_bs) {
                        while (__builtin_amdgcn_readfirstlane(*xflag) != sc) __builtin_amdgcn_s_sleep(1);
                        wave_lds_order();
                        Bs = x3_split(par, wq);
                    }
#pragma unroll
                    for (int blk = 0; blk < NBLK; ++blk)
                        if (blk >= 1 || DWO0_LATE) dwo_block(Bs, wq, blk);

                }
                __syncthreads();  // barrier B
                if constexpr (HPC != 0) {
                    if (j == S - 1) sig = h2_pick_scale(__builtin_fmaxf(__builtin_fmaxf(amax[0], amax[1]), __builtin_fmaxf(amax[2], amax[3])), sig);
                }
                NCDE_TICK(5)
                wprev = wq;
            }
        }
        if (wprev != 0.0f) dw_hidden(sc & 1, wprev);
        if constexpr (PROF != 0) {
            if (lane == 0) {
                unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.out) + ((long long)blockIdx.x * 8 + wave) * 6;
                for (int k = 0; k < 6; ++k) dst[k] = prof[k];
            }
        }
        // ---- write-out of this workgroup's parameter-gradient partial ------------------------------------------
        float* gp = a.gpart + (long long)blockIdx.x * a.theta_size;
        if constexpr (HPC != 0) {
            const float un = h2_inv_scale(acc_sig);
#pragma unroll
            for (int i = 0; i < NBLK; ++i) {
                gbo[i] *= un;
#pragma unroll
                for (int q = 0; q < 16; ++q) gWo[i][q] *= un;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { gW1[r] *= un; gW0[r] *= un; }
            gb1 *= un;
            gb0 *= un;
        }
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int nb = q >> 3;                                   // block row >> 4 = tile within the block = nb
                const int gr = 2 * ((q >> 2) & 1) + (lane >> 5), rr = q & 3;
                const int h = 4 * (pw * NB + nb) + gr, c = 4 * blk + rr;
                if (c < C) gp[a.gWo_off + (h * C + c) * HH + (lane & 31)] = gWo[blk][q];
            }
            float v = gbo[blk];
            v += __shfl_xor(v, 32, 64);
            const int i32 = lane & 31;
            const int nb = i32 >> 4, rowt = i32 & 15;
            const int hrow = 4 * (pw * NB + nb) + (rowt >> 2), crow = 4 * blk + (rowt & 3);
            if (lane < 32 && crow < C) gp[a.gbo_off + hrow * C + crow] = v;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (NL > 1) gp[a.gW_off[1] + (16 * tr + 4 * g + r) * HH + 16 * tc + s] = gW1[r];
            gp[a.gW_off[0] + (16 * tr + 4 * g + r) * H + 16 * tc + s] = gW0[r];
        }
        {
            float v1 = gb1, v0 = gb0;
            v1 += __shfl_xor(v1, 16, 64); v1 += __shfl_xor(v1, 32, 64);
            v0 += __shfl_xor(v0, 16, 64); v0 += __shfl_xor(v0, 32, 64);
            if (tc == 0 && g == 0) {
                if constexpr (NL > 1) gp[a.gb_off[1] + 16 * tr + s] = v1;
                gp[a.gb_off[0] + 16 * tr + s] = v0;
            }
        }
    }
    if constexpr (HP != 0) {
        if (a.fault != nullptr) {
            if (__builtin_amdgcn_ballot_w64(h2_range_fault(mx)) != 0 && lane == 0) *fault_s = 1;
            __syncthreads();
            if (tid == 0) a.fault[blockIdx.x] = *fault_s;
        }
    }
#undef NCDE_TICK
}

#ifndef NCDE_FAST_KERNELS_ONLY      // (ncde_fast_nl.hip includes this file for the kernel templates alone)
// ------------------------------------------------------------------------------------------------
// dispatch tables
// ------------------------------------------------------------------------------------------------
struct Shape {
    int H, HH, C;
};

using FwdFn = void (*)(KArgs);

template <int H, int HH, int C, int NW>
FwdFn pick_fwd(int interp, int method) {
#define NCDE_PICK(I, M) \
    if (interp == I && method == M) return ncde_fwd_fast<H, HH, C, NW, I, M>;
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_RK4_38)
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_MIDPOINT)
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_EULER)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_RK4_38)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_MIDPOINT)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_EULER)
#undef NCDE_PICK
    return nullptr;
}

template <int H, int HH, int C, int NW, int HP>
FwdFn pick_fwd_split(int interp, int method, int n_layers) {
    // (the unrolled three-layer instantiations live in ncde_fast_fwd3.hip: the one unit built with MFMA results in VGPRs)
    if (n_layers == 3 && NW == 4 && ((H == 32 && HH == 32 && C == 20) || (H == 64 && HH == 64 && C == 4))) return ncde_fast_fwd3(H, interp, method, HP);
#define NCDE_PICK(I, M) \
    if (interp == I && method == M) return ncde_fwd_fast_bf3<H, HH, C, NW, I, M, 0, 0, HP>;
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_RK4_38)
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_MIDPOINT)
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_EULER)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_RK4_38)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_MIDPOINT)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_EULER)
#undef NCDE_PICK
    return nullptr;
}
template <int H, int HH, int C, int NW>
FwdFn pick_fwd_bf3(int interp, int method, int n_layers, int hp) {
    return hp ? pick_fwd_split<H, HH, C, NW, 1>(interp, method, n_layers) : pick_fwd_split<H, HH, C, NW, 0>(interp, method, n_layers);
}

template <int H, int HH, int C, int NL, int NW>
FwdFn pick_adj(int interp, int method) {
#define NCDE_PICK(I, M) \
    if (interp == I && method == M) return ncde_adj_fast<H, HH, C, NL, NW, I, M>;
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_RK4_38)
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_MIDPOINT)
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_EULER)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_RK4_38)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_MIDPOINT)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_EULER)
#undef NCDE_PICK
    return nullptr;
}

template <int H, int HH, int C, int NL>
FwdFn pick_adj2(int interp, int method) {
#define NCDE_PICK(I, M) \
    if (interp == I && method == M) return ncde_adj_fast2<H, HH, C, NL, I, M>;
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_RK4_38)
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_MIDPOINT)
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_EULER)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_RK4_38)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_MIDPOINT)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_EULER)
#undef NCDE_PICK
    return nullptr;
}

template <int H, int HH, int C, int NL>
size_t adj2_lds_bytes(int interp) {
    constexpr int NW = 4, CP = (C + 3) & ~3, CQ = CP / 4, NB = H / 4 / NW, NTILE = NB * CQ;
    const int DXW = interp == NCDE_INTERP_LINEAR ? CP : 3 * CP;
    return sizeof(float) * (size_t)(2 * H * 16 + 3 * 16 * DXW + NW * HH * 16 + NW * NTILE * 16 + 2 * NW * NTILE * 256 +
                                    2 * (H + NL * HH) * 16 + NL * HH * 16 + NW * 2 * NTILE + 2 * (HH / 16) * 16 +
                                    (HH / 16) * (HH / 16) * 256 + NW * (HH / 16) * 256);
}

// The all-split-fp16 instantiations (HP = 1) exist in development builds only (-DNCDE_DEV_KNOBS): the shipped
// library neither contains them nor honours NCDE_FLAG_ADJOINT_SPLIT_FP16.
#ifdef NCDE_DEV_KNOBS
#define NCDE_ADJ3_HP1(NL_, C_, I_, M_, P_, D_) ncde_adj_fast3<NL_, C_, I_, M_, P_, D_, 1>
#define NCDE_ADJ3_HP1_AVAILABLE 1
#else
#define NCDE_ADJ3_HP1(NL_, C_, I_, M_, P_, D_) static_cast<FwdFn>(nullptr)
#define NCDE_ADJ3_HP1_AVAILABLE 0
#endif

template <int NL, int C>
FwdFn pick_adj3(int interp, int method, int hp) {
#define NCDE_PICK(I, M) \
    if (interp == I && method == M) return hp == 2 ? ncde_adj_fast3<NL, C, I, M, 0, 0, 2> : (hp == 1 ? NCDE_ADJ3_HP1(NL, C, I, M, 0, 0) : ncde_adj_fast3<NL, C, I, M>);
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_RK4_38)
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_MIDPOINT)
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_EULER)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_RK4_38)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_MIDPOINT)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_EULER)
#undef NCDE_PICK
    return nullptr;
}

template <int NL, int C>
FwdFn pick_adj3_disc(int interp, int method, int hp) {
#define NCDE_PICK(I, M) \
    if (interp == I && method == M) return hp == 2 ? ncde_adj_fast3<NL, C, I, M, 0, 1, 2> : (hp == 1 ? NCDE_ADJ3_HP1(NL, C, I, M, 0, 1) : ncde_adj_fast3<NL, C, I, M, 0, 1>);
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_RK4_38)
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_MIDPOINT)
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_EULER)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_RK4_38)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_MIDPOINT)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_EULER)
#undef NCDE_PICK
    return nullptr;
}

#endif  // NCDE_FAST_KERNELS_ONLY
template <int NL, int C>
size_t adj3_lds_bytes(int interp, int hp, int plan_stages);
template <int NL, int C>
size_t adj3_lds_bytes(int interp, int hp) { return adj3_lds_bytes<NL, C>(interp, hp, 0); }
template <int NL, int C>
size_t adj3_lds_bytes(int interp, int hp, int plan_stages) {   // hp: the kernel's HP template argument (0, 1, 2); plan_stages: S of a PLAN = 1 instance
    constexpr int H = 32, HH = 32, NW = 4, HT = 2, CP = (C + 3) & ~3, CQ = CP / 4, NB = 2, NTILE = NB * CQ;
    const int DXW = interp == NCDE_INTERP_LINEAR ? CP : 3 * CP;
    const int DXR = plan_stages ? 2 * plan_stages * 16 * CP : 3 * 16 * DXW;
    const int NPF = hp ? 2 : 3, NP = hp == 1 ? 2 : 3, NDP = ((hp == 1 && NCDE_H2_DW_LATE) || (hp == 2 && NCDE_F2_DW_LATE && interp == NCDE_INTERP_LINEAR)) ? 2 : 1;
    return sizeof(float) * (size_t)(2 * H * 16 + DXR + NW * HH * 16 + NW * NTILE * 16 + NW * NTILE * 256 +
                                    2 * (H + NL * HH) * 16 + NDP * NL * HH * 16 + NW * (NTILE + 2) + 2 * HT * 16 +
                                    HT * NP * 256 + NW * NP * 256 + ((hp == 2 && !NCDE_F2_DXL_BF3) ? 0 : NW * NB * CQ * 256) + 2 * HT * NPF * 256 + NW + 4);
}

#ifndef NCDE_FAST_KERNELS_ONLY
template <int H, int HH, int C, int NL, int NW>
size_t adj_lds_bytes(int interp) {
    constexpr int CP = (C + 3) & ~3, CQ = CP / 4, HT = HH / 16, NB = H / 4 / NW, NTILE = NB * CQ;
    const int DXW = interp == NCDE_INTERP_LINEAR ? CP : 3 * CP;
    constexpr int PRIV = (H + NL * HH) * 20 + HH * 20;
    return sizeof(float) * (size_t)(2 * H * 16 + 3 * 16 * DXW + NW * HH * 16 + NW * NTILE * HT * 256 + NW * NTILE * 16 + NW * PRIV);
}

struct FastEntry {
    Shape shape;
    int nw;
    FwdFn (*fwd)(int, int);
    const char* fwd_name;
    FwdFn (*fwd_bf3)(int, int, int, int);  // split-GEMM variants (default: split-fp16; NCDE_FLAG_SPLIT_BF16: split-bf16); NCDE_FLAG_FP32_MFMA selects `fwd`
    const char* fwd_bf3_name;
    const char* fwd_h2_name;
    int nw_bf3;
    int adj_layers;                 // n_layers the adjoint instantiation is built for (0 = none)
    FwdFn (*adj)(int, int);
    size_t (*adj_lds)(int);
    const char* adj_name;
    FwdFn (*adj2)(int, int);        // wave-specialised variant, fp32 chain (NCDE_FLAG_ADJOINT_V2)
    size_t (*adj2_lds)(int);
    const char* adj2_name;
    FwdFn (*adj3)(int, int, int);   // wave-specialised variant, split GEMMs (default); last argument: 1 = split-fp16, 0 = split-bf16
    size_t (*adj3_lds)(int, int);
    const char* adj3_name;
    const char* adj3_h2_name;       // HP = 1
    const char* adj3_f2_name;       // HP = 2
    FwdFn (*adj3_disc)(int, int, int);   // same kernel transposing the discretised solve (ncde_backward)
    const char* adj3_disc_name;
    const char* adj3_disc_h2_name;
    const char* adj3_disc_f2_name;
};

const FastEntry kFast[] = {
    // BASELINE cfg2 / cfg3
    {{32, 32, 20}, 4, pick_fwd<32, 32, 20, 4>, "ncde_fwd_fast<H32,HH32,C20,NW4>",
     pick_fwd_bf3<32, 32, 20, 4>, "ncde_fwd_fast_bf3<H32,HH32,C20,NW4,bf16x3>", "ncde_fwd_fast_bf3<H32,HH32,C20,NW4,fp16x2>", 4,
     3, pick_adj<32, 32, 20, 3, 4>, adj_lds_bytes<32, 32, 20, 3, 4>, "ncde_adj_fast<H32,HH32,C20,NL3,NW4>",
     pick_adj2<32, 32, 20, 3>, adj2_lds_bytes<32, 32, 20, 3>, "ncde_adj_fast2<H32,HH32,C20,NL3,chain+grad>",
     pick_adj3<3, 20>, adj3_lds_bytes<3, 20>, "ncde_adj_fast3<H32,HH32,C20,NL3,chain+grad,bf16x3>", "ncde_adj_fast3<H32,HH32,C20,NL3,chain+grad,fp16x2>",
     "ncde_adj_fast3<H32,HH32,C20,NL3,chain+grad,fwd-side fp16x2 + bf16x3>",
     pick_adj3_disc<3, 20>, "ncde_adj_fast3<H32,HH32,C20,NL3,chain+grad,bf16x3,discrete>", "ncde_adj_fast3<H32,HH32,C20,NL3,chain+grad,fp16x2,discrete>",
     "ncde_adj_fast3<H32,HH32,C20,NL3,chain+grad,fwd-side fp16x2 + bf16x3,discrete>"},
    // BASELINE cfg4 (adjoint: generic family for now -- the per-wave LDS images do not fit at HH=64)
    {{64, 64, 4}, 4, pick_fwd<64, 64, 4, 4>, "ncde_fwd_fast<H64,HH64,C4,NW4>",
     pick_fwd_bf3<64, 64, 4, 4>, "ncde_fwd_fast_bf3<H64,HH64,C4,NW4,bf16x3>", "ncde_fwd_fast_bf3<H64,HH64,C4,NW4,fp16x2>", 4, 0, nullptr, nullptr, nullptr,
     nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr},
    // round 5: the forward alone for up to 40 channels at H, HH <= 32 (its 160 + 80 registers of output-layer operands fit the one
    // wave per SIMD the forward runs at; the adjoint's chain / gradient waves, two per SIMD, have no room for them): inference and the
    // forward half of a training step of such shapes leave the batch-tiled family.  Default time axis only.
    {{32, 32, 40}, 4, pick_fwd<32, 32, 40, 4>, "ncde_fwd_fast<H32,HH32,C40,NW4>",
     pick_fwd_bf3<32, 32, 40, 4>, "ncde_fwd_fast_bf3<H32,HH32,C40,NW4,bf16x3>", "ncde_fwd_fast_bf3<H32,HH32,C40,NW4,fp16x2>", 4, 0, nullptr, nullptr, nullptr,
     nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr},
};

const FastEntry* find_entry(const NcdeProblem* p) {
    // structure the fast family understands: layer 0 is H->HH, every further layer is ONE shared HH->HH
    if (p->n_layers < 1) return nullptr;
    const int HH = p->layer_out[0];
    for (int l = 1; l < p->n_layers; ++l)
        if (p->layer_out[l] != HH || p->layer_in[l] != HH || p->layer_W[l] != p->layer_W[1] || p->layer_b[l] != p->layer_b[1]) return nullptr;
    for (const FastEntry& e : kFast)
        if (e.shape.H == p->hidden && e.shape.HH == HH && e.shape.C == p->channels) return &e;
    return nullptr;
}

// the decoupled-chain adjoint (ncde_fast4.hip): selectable (NCDE_FLAG_ADJOINT_V4), not the default -- measured slower
bool use_v4(const NcdeProblem* p, const FastEntry* e, bool discrete) {
    if (e->shape.H != 32 || e->shape.HH != 32) return false;
    if (!(p->flags & NCDE_FLAG_ADJOINT_V4) || (p->flags & (NCDE_FLAG_ADJOINT_V1 | NCDE_FLAG_ADJOINT_V2 | 0x200u))) return false;
    if (discrete && (p->flags & NCDE_FLAG_DEBUG_PROFILE)) return false;
    return ncde_fast4_pick(p->n_layers, p->channels, p->interp, p->method, discrete, (p->flags & NCDE_FLAG_DEBUG_PROFILE) != 0) != nullptr;
}

}  // namespace

// (32, 32, 20) with a layer count besides the entry's own: the further instantiations of ncde_adj_fast3 in ncde_fast_nl.hip
static int nl_hp(const NcdeProblem* p) { return (p->flags & (NCDE_FLAG_SPLIT_BF16 | NCDE_FLAG_FP32_MFMA)) ? 0 : 2; }
static bool use_nl(const NcdeProblem* p, const FastEntry* e, int pass) {
    if (pass < 1 || !e || e->adj3 == nullptr || e->adj_layers == p->n_layers) return false;
    if (p->flags & (NCDE_FLAG_ADJOINT_V1 | NCDE_FLAG_ADJOINT_V2 | NCDE_FLAG_ADJOINT_V4 | NCDE_FLAG_DEBUG_PROFILE | 0x200u)) return false;
    return ncde_fast_adj3_nl(p->n_layers, p->interp, p->method, nl_hp(p), pass == 2) != nullptr;
}
// (H, HH) = (32, 32) with C = 4 / 8 / 12 (round 6, ncde_fast_c.hip): the same kernel templates instantiated for few channels, every
// layer count 1 .. 4, forward + continuous adjoint + exact discrete backward; default time axis, default / split-bf16 arithmetic
static int c_set(const NcdeProblem* p) {
    if (p->hidden != 32 || p->n_layers < 1 || p->n_layers > 4 || p->layer_out[0] != 32 || p->layer_in[0] != 32) return 0;
    if (p->channels != 4 && p->channels != 8 && p->channels != 12) return 0;
    if (p->field_kind != NCDE_FIELD_ORIGINAL || p->field_input != NCDE_INPUT_MATMUL || p->output == NCDE_OUT_TIMES) return 0;
    for (int l = 1; l < p->n_layers; ++l)
        if (p->layer_out[l] != 32 || p->layer_in[l] != 32 || p->layer_W[l] != p->layer_W[1] || p->layer_b[l] != p->layer_b[1]) return 0;
    if (p->n_layers > 1 && (p->layer_W[1] == p->layer_W[0] || p->layer_b[1] == p->layer_b[0])) return 0;
    if (p->flags & (NCDE_FLAG_FP32_MFMA | NCDE_FLAG_ADJOINT_V1 | NCDE_FLAG_ADJOINT_V2 | NCDE_FLAG_ADJOINT_V4 | NCDE_FLAG_ADJOINT_SPLIT_FP16 |
                    NCDE_FLAG_DEBUG_PROFILE | 0x200u)) return 0;
    return p->channels;
}
static NcdeFastCKernel c_fwd(int C, int interp, int method, int hp) {
    return C == 4 ? ncde_fast_c4_fwd(interp, method, hp) : (C == 8 ? ncde_fast_c8_fwd(interp, method, hp) : ncde_fast_c12_fwd(interp, method, hp));
}
static NcdeFastCKernel c_adj(int C, int nl, int interp, int method, int hp, bool disc) {
    return C == 4 ? ncde_fast_c4_adj(nl, interp, method, hp, disc) : (C == 8 ? ncde_fast_c8_adj(nl, interp, method, hp, disc) : ncde_fast_c12_adj(nl, interp, method, hp, disc));
}
static size_t c_adj_lds(int C, int nl, int interp, int hp) {
    return C == 4 ? ncde_fast_c4_adj_lds(nl, interp, hp) : (C == 8 ? ncde_fast_c8_adj_lds(nl, interp, hp) : ncde_fast_c12_adj_lds(nl, interp, hp));
}
static bool use_c(const NcdeProblem* p, int pass) {
    const int C = c_set(p);
    if (!C) return false;
    if (pass == 0) return c_fwd(C, p->interp, p->method, 1) != nullptr;
    return c_adj(C, p->n_layers, p->interp, p->method, nl_hp(p), pass == 2) != nullptr && c_adj(C, p->n_layers, p->interp, p->method, 0, pass == 2) != nullptr;
}
static const char* c_name(const NcdeProblem* p, int pass) {
    static char buf[3][5][3][2][96];      // [channel set][layers][pass][split]: filled on first use (idempotent: racing threads write the same bytes)
    const int C = p->channels, ci = C == 4 ? 0 : (C == 8 ? 1 : 2), bf = (p->flags & NCDE_FLAG_SPLIT_BF16) ? 1 : 0;
    char* b = buf[ci][p->n_layers][pass][bf];
    if (!b[0]) {
        char tmp[96];
        if (pass == 0) snprintf(tmp, sizeof(tmp), "ncde_fwd_fast_bf3<H32,HH32,C%d,NW4,%s>", C, bf ? "bf16x3" : "fp16x2");
        else snprintf(tmp, sizeof(tmp), "ncde_adj_fast3<H32,HH32,C%d,NL%d,chain+grad,%s%s>", C, p->n_layers, bf ? "bf16x3" : "fwd-side fp16x2 + bf16x3", pass == 2 ? ",discrete" : "");
        memcpy(b, tmp, sizeof(tmp));
    }
    return b;
}
// H = HH = 64, C <= 4: the in-sweep adjoint of ncde_fast64.hip (the entry's own adjoint slots are empty; C < 4 has no entry at all)
static bool use_h64(const NcdeProblem* p, int pass) { return pass >= 1 && ncde_fast64_supported(p, pass); }

// general time axis (NcdeProblem.time_plan): the PLAN = 1 instances of ncde_fast_plan.hip -- forward of both shapes (split GEMMs,
// runtime layer count), continuous adjoint of (32, 32, 20) with nl = 3; (64, 64, <= 4): ncde_fast64.hip's own planned instances
static bool planned_ok(const NcdeProblem* p, const FastEntry* e, int pass) {
    if (p->flags & (NCDE_FLAG_FP32_MFMA | NCDE_FLAG_ADJOINT_V1 | NCDE_FLAG_ADJOINT_V2 | NCDE_FLAG_ADJOINT_V4 | NCDE_FLAG_DEBUG_PROFILE | 0x200u)) return false;
    if (e != nullptr && e->shape.C == 40) return false;      // (the forward-only C <= 40 set has no plan-walking instances)
    if (pass == 0) return e != nullptr && ncde_fast_plan_fwd(e->shape.H == 32 ? 0 : 1, p->interp, p->method, 1) != nullptr;
    if (pass == 1) return e != nullptr && e->shape.H == 32 && ncde_fast_plan_adj3(p->n_layers, p->interp, p->method, 2) != nullptr;
    return false;
}

bool ncde_fast_supported(const NcdeProblem* p, int pass) {
    if (use_h64(p, pass)) return true;
    if (use_c(p, pass)) return true;
    const FastEntry* e = find_entry(p);
    if (p->output == NCDE_OUT_TIMES) return planned_ok(p, e, pass);
    if (!e) return false;
    if (pass == 0) return true;
    if (use_nl(p, e, pass)) return true;
    if (pass == 2) return e->adj3_disc != nullptr && e->adj_layers == p->n_layers;
    return e->adj != nullptr && e->adj_layers == p->n_layers;
}

const char* ncde_fast_kernel_name(const NcdeProblem* p, int pass) {
    if (use_h64(p, pass)) return ncde_fast64_kernel_name(p, pass);
    if (use_c(p, pass)) return c_name(p, pass);
    if (!ncde_fast_supported(p, pass)) return nullptr;
    const FastEntry* e = find_entry(p);
    if (p->output == NCDE_OUT_TIMES) {
        const bool bf = (p->flags & NCDE_FLAG_SPLIT_BF16) != 0;
        if (pass == 0) return e->shape.H == 32 ? (bf ? "ncde_fwd_fast_bf3<H32,HH32,C20,NW4,bf16x3,time plan>" : "ncde_fwd_fast_bf3<H32,HH32,C20,NW4,fp16x2,time plan>")
                                               : (bf ? "ncde_fwd_fast_bf3<H64,HH64,C4,NW4,bf16x3,time plan>" : "ncde_fwd_fast_bf3<H64,HH64,C4,NW4,fp16x2,time plan>");
        return bf ? "ncde_adj_fast3<H32,HH32,C20,NL3,chain+grad,bf16x3,time plan>" : "ncde_adj_fast3<H32,HH32,C20,NL3,chain+grad,fwd-side fp16x2 + bf16x3,time plan>";
    }
    if (use_nl(p, e, pass)) return ncde_fast_adj3_nl_name(p->n_layers, nl_hp(p), pass == 2);
    const bool h2f = !(p->flags & NCDE_FLAG_SPLIT_BF16);
    const bool h2 = h2f && NCDE_ADJ3_HP1_AVAILABLE != 0 && (p->flags & NCDE_FLAG_ADJOINT_SPLIT_FP16);
    const bool f2 = h2f && !h2 && !(p->flags & 0x200u);
    if (pass == 0) return ((p->flags & NCDE_FLAG_FP32_MFMA) == 0 && e->fwd_bf3) ? (h2f ? e->fwd_h2_name : e->fwd_bf3_name) : e->fwd_name;
    if (use_v4(p, e, pass == 2)) return pass == 2 ? "ncde_adj_fast4<H32,HH32,C20,NL3,y-waves+cotangent-waves(bf16x3),discrete>"
                                                  : "ncde_adj_fast4<H32,HH32,C20,NL3,y-waves+cotangent-waves(bf16x3)>";
    if (pass == 2) return h2 ? e->adj3_disc_h2_name : (f2 ? e->adj3_disc_f2_name : e->adj3_disc_name);
    if (p->flags & NCDE_FLAG_ADJOINT_V1) return e->adj_name;
    if ((p->flags & NCDE_FLAG_ADJOINT_V2) && e->adj2) return e->adj2_name;
    return e->adj3 ? (h2 ? e->adj3_h2_name : (f2 ? e->adj3_f2_name : e->adj3_name)) : (e->adj2 ? e->adj2_name : e->adj_name);
}

// range-fault words of the split-fp16 kernels, one per workgroup, at the tail of the workspace
static int64_t fault_bytes(const Layout& y) { return ((int64_t)y.n_wg * 4 + 255) & ~(int64_t)255; }

int64_t ncde_fast_workspace_bytes(const NcdeProblem* p, int pass) {
    if (use_h64(p, pass)) return ncde_fast64_workspace_bytes(p, pass);
    // (the few-channel instances of ncde_fast_c.hip use the generic layout below: partials + fault words)
    if (!ncde_fast_supported(p, pass)) return NCDE_ERR_UNSUPPORTED;
    const Layout y = make_layout(p);
    if (pass == 0) return ((p->flags & NCDE_FLAG_DEBUG_PROFILE) ? 256 + (int64_t)y.n_wg * 8 * 4 * 8 : 256) + fault_bytes(y);
    return (int64_t)sizeof(float) * (int64_t)y.n_wg * (int64_t)y.theta_size + 256 +
           ((p->flags & NCDE_FLAG_DEBUG_PROFILE) ? (int64_t)y.n_wg * 8 * 6 * 8 + 256 : 0) +
           ((p->flags & 0x200u) ? (int64_t)p->n_knots * 4 * 5 * 64 * 4 + 256 : 0) + fault_bytes(y);
}

int ncde_fast_forward(const NcdeProblem* p, float* out, float* stages, void* ws, size_t ws_bytes, hipStream_t st) {
    (void)ws_bytes;
    if (use_c(p, 0)) {      // few channels: ncde_fast_c.hip's instances, the launch protocol of the split-fp16 forward below
        const Layout y = make_layout(p);
        KArgs a;
        fill_kargs(p, y, &a);
        a.out = out;
        a.stages = stages;
        const int hp = (p->flags & NCDE_FLAG_SPLIT_BF16) ? 0 : 1;
        a.fault = hp ? reinterpret_cast<int*>(static_cast<char*>(ws) + ncde_fast_workspace_bytes(p, 0) - fault_bytes(y)) : nullptr;
        hipLaunchKernelGGL(c_fwd(p->channels, p->interp, p->method, hp), dim3(y.n_wg), dim3(256), 0, st, a);
        if (hp) {      // re-execution of range-faulted tiles in split-bf16 (normally none: every workgroup exits at once)
            a.only_faulted = 1;
            hipLaunchKernelGGL(c_fwd(p->channels, p->interp, p->method, 0), dim3(y.n_wg), dim3(256), 0, st, a);
        }
        return hipGetLastError() == hipSuccess ? NCDE_OK : NCDE_ERR_HIP;
    }
    const FastEntry* e = find_entry(p);
    if (!e) return NCDE_ERR_UNSUPPORTED;
    FwdFn fn = e->fwd(p->interp, p->method);
    if (!fn) return NCDE_ERR_UNSUPPORTED;
    const Layout y = make_layout(p);
    KArgs a;
    fill_kargs(p, y, &a);
    a.out = out;
    a.stages = stages;
    const bool bf3 = (p->flags & NCDE_FLAG_FP32_MFMA) == 0 && e->fwd_bf3 != nullptr;
    const int hp = (bf3 && !(p->flags & NCDE_FLAG_SPLIT_BF16)) ? 1 : 0;
    if (bf3) fn = e->fwd_bf3(p->interp, p->method, p->n_layers, hp);
    const bool planned = p->output == NCDE_OUT_TIMES;
    if (planned) {
        if (!planned_ok(p, e, 0)) return NCDE_ERR_UNSUPPORTED;
        fn = ncde_fast_plan_fwd(e->shape.H == 32 ? 0 : 1, p->interp, p->method, hp);
    }
    a.fault = hp ? reinterpret_cast<int*>(static_cast<char*>(ws) + ncde_fast_workspace_bytes(p, 0) - fault_bytes(y)) : nullptr;
    if (p->flags & NCDE_FLAG_DEBUG_PROFILE) {  // phase-cycle counters -> workspace [n_wg][NW][4] u64
        if (!(e->shape.H == 32 && e->shape.C == 20 && p->interp == NCDE_INTERP_LINEAR && p->method == NCDE_RK4_38)) return NCDE_ERR_UNSUPPORTED;
        fn = bf3 ? (p->n_layers == 3 ? (hp ? ncde_fwd_fast_bf3<32, 32, 20, 4, NCDE_INTERP_LINEAR, NCDE_RK4_38, 1, 3, 1>
                                           : ncde_fwd_fast_bf3<32, 32, 20, 4, NCDE_INTERP_LINEAR, NCDE_RK4_38, 1, 3>)
                                     : ncde_fwd_fast_bf3<32, 32, 20, 4, NCDE_INTERP_LINEAR, NCDE_RK4_38, 1>)
                 : ncde_fwd_fast<32, 32, 20, 4, NCDE_INTERP_LINEAR, NCDE_RK4_38, 1>;
        a.gpart = (float*)ws;
    }
    hipLaunchKernelGGL(fn, dim3(y.n_wg), dim3(64 * (bf3 ? e->nw_bf3 : e->nw)), 0, st, a);
    if (hp && !(p->flags & NCDE_FLAG_DEBUG_PROFILE)) {  // re-execution of range-faulted tiles (normally none: every workgroup exits at once)
        a.only_faulted = 1;
        FwdFn fx = planned ? ncde_fast_plan_fwd(e->shape.H == 32 ? 0 : 1, p->interp, p->method, 0) : e->fwd_bf3(p->interp, p->method, p->n_layers, 0);
        hipLaunchKernelGGL(fx, dim3(y.n_wg), dim3(64 * e->nw_bf3), 0, st, a);
    }
    return hipGetLastError() == hipSuccess ? NCDE_OK : NCDE_ERR_HIP;
}

int ncde_fast_adjoint(const NcdeProblem* p, const float* z_out, const float* grad_out, const NcdeGrads* g, void* ws,
                      size_t ws_bytes, hipStream_t st, bool main_kernel_only, bool discrete) {
    if (use_h64(p, discrete ? 2 : 1)) return ncde_fast64_adjoint(p, z_out, grad_out, g, ws, ws_bytes, st, main_kernel_only, discrete);
    if (!ncde_fast_supported(p, discrete ? 2 : 1)) return NCDE_ERR_UNSUPPORTED;
    if (use_c(p, discrete ? 2 : 1)) {      // few channels: the same kernel template and launch protocol as the other layer counts below
        const int hp = nl_hp(p);
        const Layout y = make_layout(p);
        KArgs a;
        fill_kargs(p, y, &a);
        a.grad_out = grad_out; a.grad_z0 = g->grad_z0;
        if (discrete) { a.stages = const_cast<float*>(z_out); a.discrete = 1; }
        else a.z_out = z_out;
        a.gpart = (float*)ws;
        a.fault = hp ? reinterpret_cast<int*>(static_cast<char*>(ws) + ncde_fast_workspace_bytes(p, discrete ? 2 : 1) - fault_bytes(y)) : nullptr;
        for (int pass_hp : {hp, 0}) {      // main launch, then (hp = 2) the split-bf16 instance on range-faulted tiles only
            NcdeFastCKernel fn = c_adj(p->channels, p->n_layers, p->interp, p->method, pass_hp, discrete);
            const size_t lds = c_adj_lds(p->channels, p->n_layers, p->interp, pass_hp);
            if (!fn || ncde_lds_optin((const void*)fn, lds) != hipSuccess) return NCDE_ERR_HIP;
            hipLaunchKernelGGL(fn, dim3(y.n_wg), dim3(512), lds, st, a);
            if (hipGetLastError() != hipSuccess) return NCDE_ERR_HIP;
            if (hp == 0) break;
            a.only_faulted = 1;
        }
        if (main_kernel_only) return NCDE_OK;
        return launch_reduce_partials(p, y, g, (const float*)ws, y.n_wg, st);
    }
    const FastEntry* e = find_entry(p);
    if (p->output == NCDE_OUT_TIMES) {      // general time axis: the PLAN = 1 instances of the same kernel template, same launch protocol
        if (discrete || !planned_ok(p, e, 1)) return NCDE_ERR_UNSUPPORTED;
        const int hp = (p->flags & NCDE_FLAG_SPLIT_BF16) ? 0 : 2;
        const Layout y = make_layout(p);
        KArgs a;
        fill_kargs(p, y, &a);
        a.grad_out = grad_out; a.grad_z0 = g->grad_z0; a.z_out = z_out;
        a.gpart = (float*)ws;
        a.fault = hp ? reinterpret_cast<int*>(static_cast<char*>(ws) + ncde_fast_workspace_bytes(p, 1) - fault_bytes(y)) : nullptr;
        for (int pass_hp : {hp, 0}) {
            NcdeFastPlanKernel fn = ncde_fast_plan_adj3(p->n_layers, p->interp, p->method, pass_hp);
            const size_t lds = ncde_fast_plan_adj3_lds(p->n_layers, p->interp, p->method, pass_hp);
            if (!fn || ncde_lds_optin((const void*)fn, lds) != hipSuccess) return NCDE_ERR_HIP;
            hipLaunchKernelGGL(fn, dim3(y.n_wg), dim3(512), lds, st, a);
            if (hipGetLastError() != hipSuccess) return NCDE_ERR_HIP;
            if (hp == 0) break;
            a.only_faulted = 1;
        }
        if (main_kernel_only) return NCDE_OK;
        return launch_reduce_partials(p, y, g, (const float*)ws, y.n_wg, st);
    }
    if (use_nl(p, e, discrete ? 2 : 1)) {      // another layer count of the same kernel template: same launch protocol as below
        const int hp = nl_hp(p);
        const Layout y = make_layout(p);
        KArgs a;
        fill_kargs(p, y, &a);
        a.grad_out = grad_out; a.grad_z0 = g->grad_z0;
        if (discrete) { a.stages = const_cast<float*>(z_out); a.discrete = 1; }
        else a.z_out = z_out;
        a.gpart = (float*)ws;
        a.fault = hp ? reinterpret_cast<int*>(static_cast<char*>(ws) + ncde_fast_workspace_bytes(p, discrete ? 2 : 1) - fault_bytes(y)) : nullptr;
        for (int pass_hp : {hp, 0}) {      // main launch, then (hp = 2) the split-bf16 instance on range-faulted tiles only
            NcdeFastNlKernel fn = ncde_fast_adj3_nl(p->n_layers, p->interp, p->method, pass_hp, discrete);
            const size_t lds = ncde_fast_adj3_nl_lds(p->n_layers, p->interp, pass_hp);
            if (!fn || ncde_lds_optin((const void*)fn, lds) != hipSuccess) return NCDE_ERR_HIP;
            hipLaunchKernelGGL(fn, dim3(y.n_wg), dim3(512), lds, st, a);
            if (hipGetLastError() != hipSuccess) return NCDE_ERR_HIP;
            if (hp == 0) break;
            a.only_faulted = 1;
        }
        if (main_kernel_only) return NCDE_OK;
        return launch_reduce_partials(p, y, g, (const float*)ws, y.n_wg, st);
    }
    const bool v1 = !discrete && ((p->flags & NCDE_FLAG_ADJOINT_V1) != 0 || (e->adj2 == nullptr && e->adj3 == nullptr));
    const bool v3 = discrete || (!v1 && e->adj3 != nullptr && !(p->flags & NCDE_FLAG_ADJOINT_V2));
    const bool v2 = !v1 && !v3;
    // hp = the kernel's HP: 2 (default) forward-side GEMMs of the chain waves split-fp16, 0 (NCDE_FLAG_SPLIT_BF16) all split-bf16,
    // 1 (NCDE_FLAG_ADJOINT_SPLIT_FP16, experimental: DESIGN.md section 5.4c) everything split-fp16
    const int hp = (!v3 || (p->flags & (NCDE_FLAG_SPLIT_BF16 | 0x200u))) ? 0
                   : ((NCDE_ADJ3_HP1_AVAILABLE != 0 && (p->flags & NCDE_FLAG_ADJOINT_SPLIT_FP16)) ? 1 : 2);
    FwdFn fn = discrete ? e->adj3_disc(p->interp, p->method, hp)
                        : (v1 ? e->adj(p->interp, p->method) : (v3 ? e->adj3(p->interp, p->method, hp) : e->adj2(p->interp, p->method)));
    if (!fn) return NCDE_ERR_UNSUPPORTED;
    if (discrete && (p->flags & (NCDE_FLAG_DEBUG_PROFILE | 0x200u))) return NCDE_ERR_UNSUPPORTED;
    const Layout y = make_layout(p);
    KArgs a;
    fill_kargs(p, y, &a);
    a.grad_out = grad_out; a.grad_z0 = g->grad_z0;
    if (discrete) { a.stages = const_cast<float*>(z_out); a.discrete = 1; }
    else a.z_out = z_out;
    a.gpart = (float*)ws;
    if (use_v4(p, e, discrete)) {
        const bool prof = (p->flags & NCDE_FLAG_DEBUG_PROFILE) != 0;
        NcdeFast4Kernel k4 = ncde_fast4_pick(p->n_layers, p->channels, p->interp, p->method, discrete, prof);
        if (prof) a.out = (float*)ws + (size_t)y.n_wg * y.theta_size + 64;
        const size_t lds4 = ncde_fast4_lds_bytes(p->n_layers, p->channels, p->interp);
        if (ncde_lds_optin((const void*)k4, lds4) != hipSuccess) return NCDE_ERR_HIP;
        hipLaunchKernelGGL(k4, dim3(y.n_wg), dim3(512), lds4, st, a);
        if (hipGetLastError() != hipSuccess) return NCDE_ERR_HIP;
        if (main_kernel_only) return NCDE_OK;
        return launch_reduce_partials(p, y, g, (const float*)ws, y.n_wg, st);
    }
    if (p->flags & 0x200u) {  // development: per-stage chain values of workgroup 0 -> tail of the workspace
        if (!(e->shape.H == 32 && e->shape.C == 20 && p->interp == NCDE_INTERP_CUBIC && p->method == NCDE_MIDPOINT)) return NCDE_ERR_UNSUPPORTED;
        fn = !v1 ? ncde_adj_fast2<32, 32, 20, 3, NCDE_INTERP_CUBIC, NCDE_MIDPOINT, 2> : ncde_adj_fast<32, 32, 20, 3, 4, NCDE_INTERP_CUBIC, NCDE_MIDPOINT, 2>;
        a.out = (float*)ws + (size_t)y.n_wg * y.theta_size + 64;
    }
    if (p->flags & NCDE_FLAG_DEBUG_PROFILE) {  // phase-cycle counters -> tail of the workspace [n_wg][NW][6] u64
        if (!(e->shape.H == 32 && e->shape.C == 20 && p->interp == NCDE_INTERP_LINEAR && p->method == NCDE_RK4_38)) return NCDE_ERR_UNSUPPORTED;
        fn = v3 ? (hp == 2 ? ncde_adj_fast3<3, 20, NCDE_INTERP_LINEAR, NCDE_RK4_38, 1, 0, 2>
                           : (hp == 1 ? NCDE_ADJ3_HP1(3, 20, NCDE_INTERP_LINEAR, NCDE_RK4_38, 1, 0) : ncde_adj_fast3<3, 20, NCDE_INTERP_LINEAR, NCDE_RK4_38, 1>))
                : (v2 ? ncde_adj_fast2<32, 32, 20, 3, NCDE_INTERP_LINEAR, NCDE_RK4_38, 1> : ncde_adj_fast<32, 32, 20, 3, 4, NCDE_INTERP_LINEAR, NCDE_RK4_38, 1>);
        a.out = (float*)ws + (size_t)y.n_wg * y.theta_size + 64;
    }
    const size_t lds = v1 ? e->adj_lds(p->interp) : (v3 ? e->adj3_lds(p->interp, hp) : e->adj2_lds(p->interp));
    if (ncde_lds_optin((const void*)fn, lds) != hipSuccess) return NCDE_ERR_HIP;
    a.fault = hp ? reinterpret_cast<int*>(static_cast<char*>(ws) + ncde_fast_workspace_bytes(p, discrete ? 2 : 1) - fault_bytes(y)) : nullptr;
    hipLaunchKernelGGL(fn, dim3(y.n_wg), dim3(v1 ? 64 * e->nw : 512), lds, st, a);
    if (hipGetLastError() != hipSuccess) return NCDE_ERR_HIP;
    if (hp && !(p->flags & NCDE_FLAG_DEBUG_PROFILE)) {  // re-execution of range-faulted tiles in split-bf16 (normally none: every workgroup exits at once)
        FwdFn fx = discrete ? e->adj3_disc(p->interp, p->method, 0) : e->adj3(p->interp, p->method, 0);
        const size_t ldx = e->adj3_lds(p->interp, 0);
        if (ncde_lds_optin((const void*)fx, ldx) != hipSuccess) return NCDE_ERR_HIP;
        a.only_faulted = 1;
        hipLaunchKernelGGL(fx, dim3(y.n_wg), dim3(512), ldx, st, a);
        if (hipGetLastError() != hipSuccess) return NCDE_ERR_HIP;
    }
    if (main_kernel_only) return NCDE_OK;
    return launch_reduce_partials(p, y, g, (const float*)ws, y.n_wg, st);
}

#endif  // NCDE_FAST_KERNELS_ONLY
