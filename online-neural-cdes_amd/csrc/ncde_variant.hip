// Vector-field VARIANTS of the reference on the generic 16-sample-tile structure:
//   gated fields   src/ncde/vector_fields/gating.py:7-61      minimal: M = sigmoid(Wg hh + bg) * tanh(Wo hh + bo)
//                                                             gru:     M = sigmoid(Wg net(u) + bg) * tanh(Wo net(rg * u) + bo),
//                                                                      rg = sigmoid(Wr u + br)
//   input modes    modules/torchcde/torchcde/solver.py:112-137 matmul:     u = z,             dz/dt = M[H,C] . dX/dt
//                                                             evaluate:   u = [z, X(t)],     dz/dt = M[H]
//                                                             derivative: u = [z, dX/dt(t)], dz/dt = M[H]
// Same design as ncde_generic.hip (activations in LDS as [unit][sample], fp32 MFMA 16x16x4, weights streamed from L2,
// the whole time loop inside the kernel, per-workgroup parameter-gradient partials); the head loop carries two
// accumulators (tanh and sigmoid head), the inner net can run twice per stage (GRU), the field input has H + C rows
// in the evaluate / derivative modes.  The continuous adjoint and the exact discrete backward share one kernel.
#include "ncde_common.h"
#include "ncde_host.h"
#include "ncde_variant.h"

#define VR_NW 4
#define VR_THREADS (64 * VR_NW)
#define VR_MAXJT 8  // last hidden width <= 128

namespace {

__device__ __forceinline__ float sigmoid_dev(float x) {
    const float e = __builtin_amdgcn_exp2f(x * -1.4426950408889634f);  // exp(-x)
    return __builtin_amdgcn_rcpf(1.0f + e);
}

// control input of the stage for the 16 samples of the tile -> CIN[c*16 + s]:
// value = false: dX/dt(t) (interpolation_linear.py:231-234, interpolation_cubic.py:331-336)
// value = true : X(t)     (interpolation_linear.py:221-229, interpolation_cubic.py:324-329)
__device__ void vr_load_cin(const KArgs& a, int b0, const StageDesc& sd, bool value, float* CIN, int Cp, int tid) {
    const int idx = sd.idx;
    const float frac = sd.frac, kdt = sd.kdt;
    for (int e = tid; e < 16 * Cp; e += VR_THREADS) {
        const int s = e / Cp, c = e - s * Cp;
        const int b = b0 + s;
        float v = 0.0f;
        if (c < a.C && b < a.B) {
            const float* p = a.coeffs + (long long)b * a.cs_b + (long long)idx * a.cs_t;
            if (a.interp == NCDE_INTERP_LINEAR) {
                const float d = p[a.cs_t + c] - p[c];
                v = value ? p[c] + (frac * d) / kdt : (kdt != 1.0f ? d / kdt : d);
            } else {
                const float aa = p[c], bb = p[a.C + c], cc = p[2 * a.C + c], dd = p[3 * a.C + c];
                if (value) {
                    float inner = 0.5f * cc + (dd * frac) / 3.0f;
                    inner = bb + inner * frac;
                    v = aa + inner * frac;
                } else {
                    const float inner = cc + dd * frac;
                    v = bb + inner * frac;
                }
            }
        }
        CIN[c * 16 + s] = v;
    }
}

// out[n][s] = act(sum_k W[n][k] in[k][s] + bias[n]), n < ru16(N) (rows >= N come out as act(0) masked to 0)
// ACT: 0 = relu, 1 = sigmoid
// wl != NULL: the matrix is resident in LDS, zero-padded to [ru16(N)][ru16(K)] with row stride ru16(K) + 1 and the bias in
// column ru16(K): at the model sizes of the reference's vector-field grids the first weight loads of every phase of a stage
// are otherwise an exposed L2 round trip, and with the padding the inner loops carry no guards (a guarded load compiles to
// an exec-mask branch per element).
template <int ACT>
__device__ void vr_dense(const float* __restrict__ W, const float* __restrict__ bias, int N, int K, const float* in, float* out,
                         int wave, int lane, const float* wl = nullptr) {
    const int li = lane & 15, lk = lane >> 4;
    const int ntiles = (N + 15) >> 4, nks = (K + 3) >> 2;
    for (int t = wave; t < ntiles; t += VR_NW) {
        const int rowA = 16 * t + li;
        const bool rv = rowA < N;
        const float* wrow = W + (long long)(rv ? rowA : 0) * K;
        f32x4 acc;
        if (wl) {
            const int Kp = ru16(K);
            const float* lrow = wl + rowA * (Kp + 1) + lk;
            const float* brow = in + lk * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = wl[(16 * t + 4 * lk + r) * (Kp + 1) + Kp];
            f32x4 acc2 = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int kb = 0; kb < Kp; kb += 16) {
                float av[4], bv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { av[e] = lrow[kb + 4 * e]; bv[e] = brow[(kb + 4 * e) * 16]; }
                acc = mfma16(av[0], bv[0], acc);
                acc2 = mfma16(av[1], bv[1], acc2);
                acc = mfma16(av[2], bv[2], acc);
                acc2 = mfma16(av[3], bv[3], acc2);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += acc2[r];
        } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * t + 4 * lk + r;
            acc[r] = row < N ? bias[row] : 0.0f;
        }
#pragma unroll 4
        for (int ks = 0; ks < nks; ++ks) {
            const int k = 4 * ks + lk;
            const float av = (rv && k < K) ? wrow[k] : 0.0f;
            acc = mfma16(av, in[k * 16 + li], acc);
        }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * t + 4 * lk + r;
            const float v = ACT == 0 ? fmaxf(acc[r], 0.0f) : sigmoid_dev(acc[r]);
            out[row * 16 + li] = row < N ? v : 0.0f;
        }
    }
}

// x_1 .. x_L of the inner net for input `in`; all layers kept when X_all (adjoint), else ping-pong into A0/A1.
// Returns the buffer holding x_L.
__device__ const float* vr_net(const KArgs& a, const float* in, float* X, int DS, bool keep_all, int wave, int lane, const float* lds) {
    for (int l = 0; l < a.n_layers; ++l) {
        float* outb = keep_all ? X + l * DS : X + (l & 1) * DS;
        vr_dense<0>(a.W[l], a.b[l], a.dout[l], a.din[l], in, outb, wave, lane, a.wres[l] >= 0 ? lds + a.wres[l] : nullptr);
        __syncthreads();
        in = outb;
    }
    return in;
}

// copy the matrices the host marked resident into LDS: [N][K] -> zero-padded [ru16(N)][ru16(K) + 1], bias in the last column
__device__ void vr_fill_resident(const KArgs& a, float* lds, int tid) {
    auto fill = [&](const float* W, const float* b, int N, int K, int off) {
        const int Kp = ru16(K), Np = ru16(N), ld = Kp + 1;
        for (int e = tid; e < Np * ld; e += VR_THREADS) {
            const int r = e / ld, c = e - r * ld;
            float v = 0.0f;
            if (r < N) v = c < K ? W[(long long)r * K + c] : (c == Kp ? b[r] : 0.0f);
            lds[off + e] = v;
        }
    };
    for (int l = 0; l < a.n_layers; ++l) {
        bool first = a.wres[l] >= 0;
        for (int q = 0; q < l; ++q) first = first && a.wres[q] != a.wres[l];
        if (first) fill(a.W[l], a.b[l], a.dout[l], a.din[l], a.wres[l]);
    }
    const int dlast = a.n_layers ? a.dout[a.n_layers - 1] : a.d0;
    if (a.wres_o >= 0) fill(a.Wo, a.bo, a.rows, dlast, a.wres_o);
    if (a.wres_g >= 0) fill(a.Wg, a.bg, a.rows, dlast, a.wres_g);
    if (a.wres_r >= 0) fill(a.Wr, a.br, a.d0, a.d0, a.wres_r);
}

struct HeadTile {
    int rowA;       // weight row this lane streams (A operand), -1 = none
    int rowD[4];    // row of D register r, -1 = none
    int hD[4];      // state index h of D register r
};

// tile q of the head: matmul -> (hb, cq) with rows (h = 4hb+g, c = 4cq+r); direct -> rows 16q .. 16q+15 = h
__device__ __forceinline__ HeadTile vr_head_tile(const KArgs& a, int q, int ncq, int li, int lk, int& cq_out) {
    HeadTile t;
    if (a.field_input == NCDE_INPUT_MATMUL) {
        const int hb = q / ncq, cq = q - hb * ncq;
        cq_out = cq;
        const int hA = 4 * hb + (li >> 2), cA = 4 * cq + (li & 3), hD = 4 * hb + lk;
        t.rowA = (hA < a.H && cA < a.C) ? hA * a.C + cA : -1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 4 * cq + r;
            t.rowD[r] = (hD < a.H && c < a.C) ? hD * a.C + c : -1;
            t.hD[r] = hD;
        }
    } else {
        cq_out = 0;
        t.rowA = 16 * q + li < a.H ? 16 * q + li : -1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int h = 16 * q + 4 * lk + r;
            t.rowD[r] = h < a.H ? h : -1;
            t.hD[r] = h;
        }
    }
    return t;
}

__device__ __forceinline__ f32x4 vr_head_gemm(const float* __restrict__ W, const float* __restrict__ bias, const HeadTile& t, int K,
                                              const float* in, int li, int lk, const float* wl = nullptr) {
    f32x4 acc;
    if (wl) {      // an absent row (rowA < 0) reads row 0 and multiplies by zero
        const int Kp = ru16(K);
        const float* lrow = wl + (t.rowA >= 0 ? t.rowA : 0) * (Kp + 1) + lk;
        const float* brow = in + lk * 16 + li;
        const float am = t.rowA >= 0 ? 1.0f : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float bvv = wl[(t.rowD[r] >= 0 ? t.rowD[r] : 0) * (Kp + 1) + Kp];
            acc[r] = t.rowD[r] >= 0 ? bvv : 0.0f;
        }
        f32x4 acc2 = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kb = 0; kb < Kp; kb += 16) {
            float av[4], bv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { av[e] = lrow[kb + 4 * e] * am; bv[e] = brow[(kb + 4 * e) * 16]; }
            acc = mfma16(av[0], bv[0], acc);
            acc2 = mfma16(av[1], bv[1], acc2);
            acc = mfma16(av[2], bv[2], acc);
            acc2 = mfma16(av[3], bv[3], acc2);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += acc2[r];
        return acc;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = t.rowD[r] >= 0 ? bias[t.rowD[r]] : 0.0f;
    const float* wrow = W + (long long)(t.rowA >= 0 ? t.rowA : 0) * K;
    const int nks = (K + 3) >> 2;
#pragma unroll 4
    for (int ks = 0; ks < nks; ++ks) {
        const int k = 4 * ks + lk;
        const float av = (t.rowA >= 0 && k < K) ? wrow[k] : 0.0f;
        acc = mfma16(av, in[k * 16 + li], acc);
    }
    return acc;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(VR_THREADS) void ncde_fwd_variant(KArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int b0 = blockIdx.x * NCDE_TILE;
    const int H = a.H, C = a.C, Hp = ru16(H), Cp = ru4(C), d0 = a.d0, L = a.n_layers;
    int Dp = max(Hp, ru16(d0));
    for (int l = 0; l < L; ++l) Dp = max(Dp, ru16(a.dout[l]));
    const int HS = Hp * 16, DS = Dp * 16;
    const bool matmul = a.field_input == NCDE_INPUT_MATMUL, gru = a.field_kind == NCDE_FIELD_GRU, gated = a.field_kind != NCDE_FIELD_ORIGINAL;
    float* U = lds;              // field input [d0][16]: rows < H = stage state, rows H.. = control input
    float* RU = U + DS;          // rg * u (gru)
    float* RG = RU + DS;         // reset gate (gru)
    float* XA = RG + DS;         // ping-pong activations, inner pass (2 buffers)
    float* XB = XA + 2 * DS;     // ping-pong activations, reset pass (2 buffers)
    float* Y0 = XB + 2 * DS;
    float* K1 = Y0 + HS;
    float* K2 = K1 + HS;
    float* KO = K2 + HS;
    float* CIN = KO + HS;        // [Cp][16]
    const int total = 7 * DS + 4 * HS + Cp * 16;
    for (int e = tid; e < total; e += VR_THREADS) lds[e] = 0.0f;
    vr_fill_resident(a, lds, tid);
    const float* wl_o = a.wres_o >= 0 ? lds + a.wres_o : nullptr;
    const float* wl_g = a.wres_g >= 0 ? lds + a.wres_g : nullptr;
    const float* wl_r = a.wres_r >= 0 ? lds + a.wres_r : nullptr;
    __syncthreads();
    for (int e = tid; e < HS; e += VR_THREADS) {
        const int h = e >> 4, s = e & 15, b = b0 + s;
        if (h < H && b < a.B) {
            const float v = a.z0[(long long)b * H + h];
            Y0[e] = v;
            U[e] = v;
            a.out[((long long)b * a.n_out) * H + h] = v;
        }
    }
    const int S = n_stages(a.method);
    const int dlast = L ? a.dout[L - 1] : d0;
    const int ncq = Cp >> 2, ngrp = matmul ? (Hp >> 2) : (Hp >> 4), per_grp = matmul ? ncq : 1;
    const bool planned = a.plan != nullptr;
    if (planned && !plan_header_ok(a, S)) return;      // (uniform: before the first barrier)
    const int n_steps = planned ? a.n_steps_fwd : a.T - 1;
    const int* pout = planned ? a.plan + plan_off_out(S, a.n_steps_fwd) : nullptr;
    for (int n = 0; n < n_steps; ++n) {
        const int* pstep = planned ? a.plan + plan_off_fwd() + n * plan_step_words(S) : nullptr;
        const float dt = planned ? __int_as_float(pstep[0]) : 1.0f;
        for (int j = 0; j < S; ++j) {
            const StageDesc sd = planned ? plan_stage(pstep, j) : default_stage(a.method, (float)n + stage_offset(a.method, j), a.n_pieces);
            vr_load_cin(a, b0, sd, a.field_input == NCDE_INPUT_EVALUATE, CIN, Cp, tid);
            __syncthreads();
            if (!matmul)
                for (int e = tid; e < C * 16; e += VR_THREADS) U[H * 16 + e] = CIN[e];
            if (a.stages) {
                float* rec = a.stages + ((long long)(n * S + j) * a.B + b0) * H;
                for (int e = tid; e < 16 * H; e += VR_THREADS) {
                    const int s = e / H, h = e - s * H;
                    if (b0 + s < a.B) rec[e] = U[h * 16 + s];
                }
            }
            __syncthreads();
            if (gru) {
                vr_dense<1>(a.Wr, a.br, d0, d0, U, RG, wave, lane, wl_r);
                __syncthreads();
                for (int e = tid; e < ru16(d0) * 16; e += VR_THREADS) RU[e] = RG[e] * U[e];
                __syncthreads();
            }
            const float* xi = vr_net(a, U, XA, DS, false, wave, lane, lds);
            const float* xr = gru ? vr_net(a, RU, XB, DS, false, wave, lane, lds) : xi;
            // groups of head tiles: matmul -> one h-block (all its channel quads, contraction accumulated in a
            // register); evaluate / derivative -> one 16-row tile
            for (int grp = wave; grp < ngrp; grp += VR_NW) {
                float ksum = 0.0f;
                for (int qi = 0; qi < per_grp; ++qi) {
                    int cq;
                    const HeadTile ht = vr_head_tile(a, grp * per_grp + qi, ncq, li, lk, cq);
                    const f32x4 pt = vr_head_gemm(a.Wo, a.bo, ht, dlast, xr, li, lk, wl_o);
                    f32x4 ps = pt;
                    if (gated) ps = vr_head_gemm(a.Wg, a.bg, ht, dlast, xi, li, lk, wl_g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float m = tanh_dev(pt[r]);
                        if (gated) m = sigmoid_dev(ps[r]) * m;
                        if (matmul) ksum = fmaf(m, CIN[(4 * cq + r) * 16 + li], ksum);
                        else if (ht.hD[r] < Hp) KO[ht.hD[r] * 16 + li] = ht.rowD[r] >= 0 ? m : 0.0f;
                    }
                }
                if (matmul && 4 * grp + lk < Hp) KO[(4 * grp + lk) * 16 + li] = ksum;
            }
            __syncthreads();
            for (int e = tid; e < HS; e += VR_THREADS) {
                float y0 = Y0[e], k1 = K1[e], k2 = K2[e];
                const float yprev = y0;
                bool last;
                const float ys = StageCombine::apply(a.method, j, KO[e], dt, y0, k1, k2, last);
                U[e] = ys;
                K1[e] = k1;
                K2[e] = k2;
                if (last) {
                    Y0[e] = y0;
                    const int h = e >> 4, s = e & 15, b = b0 + s;
                    if (h < H && b < a.B) {
                        if (planned) {
                            const int q0 = pstep[1], q1 = q0 + pstep[2];
                            for (int q = q0; q < q1; ++q) {
                                const int kind = pout[2 * q];
                                const float slope = __int_as_float(pout[2 * q + 1]);
                                a.out[((long long)b * a.n_out + q) * H + h] = kind == 1 ? y0 : (kind == 0 ? yprev : yprev + slope * (y0 - yprev));
                            }
                        } else if (a.output == NCDE_OUT_KNOTS) a.out[((long long)b * a.n_out + (n + 1)) * H + h] = y0;
                        else if (n == a.T - 2) a.out[((long long)b * a.n_out + 1) * H + h] = y0;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// adjoint / exact discrete backward
// ------------------------------------------------------------------------------------------------
namespace {

// *p += v on this workgroup's PRIVATE gradient partial.  In global memory (in_lds == 0; matmul heads: 41k floats per stage at
// cfg2 dims): a hardware float add without return instead of load / add / store, so nothing waits for the old value -- the
// read-modify-write round trips were the longest thing in a stage (gru/matmul 268 -> 205 ms).  Every address is only ever
// touched by one wave, in program order, so the sum order -- and the result -- is the same on every run.  In LDS a plain
// read-modify-write is the faster one (flat atomics into the LDS aperture: evaluate 32 -> 38 ms).
__device__ __forceinline__ void vr_accum(float* p, float v, int in_lds) {
    if (in_lds) *p += v;
    else unsafeAtomicAdd(p, v);
}

// gW[j][i] += w sum_s gpre[j][s] xin[i][s];  gb[j] += w sum_s gpre[j][s]      (samples are the K dim of the MFMA)
__device__ void vr_dw_acc(const float* gpre, const float* xin, int N, int K, float w, float* gW, float* gb, int tid, int wave, int lane, int in_lds) {
    const int li = lane & 15, lk = lane >> 4;
    for (int jj = tid; jj < N; jj += VR_THREADS) {
        float sum = 0.0f;
#pragma unroll
        for (int s = 0; s < 16; ++s) sum += gpre[jj * 16 + s];
        vr_accum(gb + jj, w * sum, in_lds);
    }
    const int njt = (N + 15) >> 4, nit = (K + 15) >> 4;
    for (int tt = wave; tt < njt * nit; tt += VR_NW) {
        const int jt = tt / nit, it = tt - jt * nit;
        f32x4 g = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) g = mfma16(gpre[(16 * jt + li) * 16 + 4 * ks + lk], xin[(16 * it + li) * 16 + 4 * ks + lk], g);
        const int col = 16 * it + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * jt + 4 * lk + r;
            if (row < N && col < K) vr_accum(gW + (long long)row * K + col, w * g[r], in_lds);
        }
    }
}

// The same weight gradient kept in this wave's REGISTERS for the whole sweep (tiles tt = wave + 4 q, q < VR_DWT): no
// read-modify-write of the partial and no guards per stage -- the partial is touched once, at the end (vr_dw_flush).  Used
// for the hidden matrices and the reset gate when there are at most two distinct hidden matrices of <= 4 VR_DWT tiles each
// (the reference's stack: one input layer + ONE shared inner layer); the bias gradient stays a plain accumulate.
#define VR_DWT 4
__device__ __forceinline__ void vr_dw_acc_reg(const float* gpre, const float* xin, int N, int K, float w, f32x4 (&acc)[VR_DWT], float* gb,
                                              int tid, int wave, int lane, int in_lds) {
    const int li = lane & 15, lk = lane >> 4;
    for (int jj = tid; jj < N; jj += VR_THREADS) {
        float sum = 0.0f;
#pragma unroll
        for (int s = 0; s < 16; ++s) sum += gpre[jj * 16 + s];
        vr_accum(gb + jj, w * sum, in_lds);
    }
    const int nit = (K + 15) >> 4, ntile = ((N + 15) >> 4) * nit;
#pragma unroll
    for (int q = 0; q < VR_DWT; ++q) {
        const int tt = wave + VR_NW * q;
        if (tt < ntile) {
            const int jt = tt / nit, it = tt - jt * nit;
            const float* ap = gpre + (16 * jt + li) * 16 + lk;      // rows >= N / columns >= K of the padded activations are zero
            const float* bp = xin + (16 * it + li) * 16 + lk;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc[q] = mfma16(w * ap[4 * ks], bp[4 * ks], acc[q]);
        }
    }
}
__device__ __forceinline__ void vr_dw_flush(const f32x4 (&acc)[VR_DWT], int N, int K, float* gW, int wave, int lane) {
    const int li = lane & 15, lk = lane >> 4;
    const int nit = (K + 15) >> 4, ntile = ((N + 15) >> 4) * nit;
#pragma unroll
    for (int q = 0; q < VR_DWT; ++q) {
        const int tt = wave + VR_NW * q;
        if (tt < ntile) {
            const int jt = tt / nit, it = tt - jt * nit, col = 16 * it + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * jt + 4 * lk + r;
                if (row < N && col < K) gW[(long long)row * K + col] += acc[q][r];
            }
        }
    }
}

// out[i][s] (+)= sum_j W[j][i] gpre[j][s], i < ru16(K); optionally x relu'(mask[i][s])
__device__ void vr_bwd_data(const float* __restrict__ W, int N, int K, const float* gpre, const float* mask, float* out, bool accumulate,
                            int wave, int lane, const float* wl = nullptr) {
    const int li = lane & 15, lk = lane >> 4;
    const int nit = (K + 15) >> 4, nks = (N + 3) >> 2;
    for (int it = wave; it < nit; it += VR_NW) {
        const int col = 16 * it + li;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (wl) {
            const int Kp = ru16(K), Np = ru16(N);
            const float* lcol = wl + lk * (Kp + 1) + col;      // col < ru16(K): a zero column of the resident copy when >= K
            const float* brow = gpre + lk * 16 + li;
            f32x4 acc2 = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int kb = 0; kb < Np; kb += 16) {
                float av[4], bv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { av[e] = lcol[(kb + 4 * e) * (Kp + 1)]; bv[e] = brow[(kb + 4 * e) * 16]; }
                acc = mfma16(av[0], bv[0], acc);
                acc2 = mfma16(av[1], bv[1], acc2);
                acc = mfma16(av[2], bv[2], acc);
                acc2 = mfma16(av[3], bv[3], acc2);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += acc2[r];
        } else
#pragma unroll 4
        for (int ks = 0; ks < nks; ++ks) {
            const int k = 4 * ks + lk;
            const float av = (k < N && col < K) ? W[(long long)k * K + col] : 0.0f;
            acc = mfma16(av, gpre[k * 16 + li], acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * it + 4 * lk + r;
            float v = acc[r];
            if (mask) v = mask[row * 16 + li] > 0.0f ? v : 0.0f;
            if (accumulate) v += out[row * 16 + li];
            out[row * 16 + li] = v;
        }
    }
}

}  // namespace

extern "C" __global__ __launch_bounds__(VR_THREADS) void ncde_adj_variant(KArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int b0 = blockIdx.x * NCDE_TILE;
    const int H = a.H, C = a.C, Hp = ru16(H), Cp = ru4(C), d0 = a.d0, L = a.n_layers;
    int Dp = max(Hp, ru16(d0));
    for (int l = 0; l < L; ++l) Dp = max(Dp, ru16(a.dout[l]));
    const int HS = Hp * 16, DS = Dp * 16;
    const bool matmul = a.field_input == NCDE_INPUT_MATMUL, gru = a.field_kind == NCDE_FIELD_GRU, gated = a.field_kind != NCDE_FIELD_ORIGINAL;
    float* U = lds;                  // field input (rows < H: y stage input)
    float* RG = U + DS;
    float* RU = RG + DS;
    float* XI = RU + DS;             // [L] inner pass activations
    float* XR = XI + L * DS;         // [L] reset pass activations (gru)
    float* GA = XR + L * DS;         // dL/dpre ping
    float* GB = GA + DS;             // pong
    float* PW = GB + DS;             // [8] per-wave partials of dL/dx_L: 0..3 inner pass, 4..7 reset pass
    float* DUI = PW + 8 * DS;        // cotangent of u
    float* DUR = DUI + DS;           // cotangent of rg * u
    float* AS = DUR + DS;            // stage cotangent [Hp][16]
    float* Y0 = AS + HS;
    float* A0 = Y0 + HS;
    float* KY1 = A0 + HS;
    float* KY2 = KY1 + HS;
    float* KA1 = KY2 + HS;
    float* KA2 = KA1 + HS;
    float* KOY = KA2 + HS;
    float* KOA = KOY + HS;
    float* CIN = KOA + HS;
    float* SC = CIN + Cp * 16;       // per-wave 16x17 transpose scratch
    float* GL = SC + VR_NW * 16 * 17;
    const int total = (15 + 2 * L) * DS + 9 * HS + Cp * 16 + VR_NW * 16 * 17 + (a.gacc_in_lds ? a.theta_size : 0);
    for (int e = tid; e < total; e += VR_THREADS) lds[e] = 0.0f;
    vr_fill_resident(a, lds, tid);
    const float* wl_o = a.wres_o >= 0 ? lds + a.wres_o : nullptr;
    const float* wl_g = a.wres_g >= 0 ? lds + a.wres_g : nullptr;
    const float* wl_r = a.wres_r >= 0 ? lds + a.wres_r : nullptr;
    float* gacc = a.gacc_in_lds ? GL : a.gpart + (long long)blockIdx.x * a.theta_size;
    if (!a.gacc_in_lds)
        for (int e = tid; e < a.theta_size; e += VR_THREADS) gacc[e] = 0.0f;
    __syncthreads();
    const int last_row = a.n_out - 1;
    const int S = n_stages(a.method);
    const bool disc = a.discrete != 0;
    const bool planned = a.plan != nullptr;
    if (planned && !plan_header_ok(a, S)) return;      // (uniform: before the first barrier)
    const int pw_ = plan_step_words(S);
    const int* pfwd = planned ? a.plan + plan_off_fwd() : nullptr;
    const int* pout = planned ? a.plan + plan_off_out(S, a.n_steps_fwd) : nullptr;
    const int* padj = planned ? a.plan + plan_off_adj(S, a.n_steps_fwd, a.n_out) : nullptr;
    const int n_rsteps = planned ? (disc ? a.n_steps_fwd : a.n_steps_adj) : a.T - 1;
    for (int e = tid; e < HS; e += VR_THREADS) {
        const int h = e >> 4, s = e & 15, b = b0 + s;
        if (h < H && b < a.B) {
            const long long o = ((long long)b * a.n_out + last_row) * H + h;
            float g = a.grad_out[o];
            if (planned && disc) g = plan_out_cotangent(a, pfwd + (a.n_steps_fwd - 1) * pw_, pout, 1, (long long)b * a.n_out, h);
            A0[e] = g;
            if (disc) {
                const float dtl = planned ? __int_as_float(pfwd[(a.n_steps_fwd - 1) * pw_]) : 1.0f;
                AS[e] = a.method == NCDE_RK4_38 ? (g * dtl) * 0.125f : dtl * g;
            } else {
                const float y = a.z_out[o];
                Y0[e] = y; U[e] = y; AS[e] = g;
            }
        }
    }
    const int dlast = a.dout[L - 1];
    const int ncq = Cp >> 2, ngrp = matmul ? (Hp >> 2) : (Hp >> 4), per_grp = matmul ? ncq : 1;
    const int njt = (dlast + 15) >> 4;
    float* sc = SC + wave * 16 * 17;
    // hidden / reset-gate weight gradients in registers (see vr_dw_acc_reg): slot 0 = layer 0's matrix, slot 1 = the other one
    f32x4 dwA[VR_DWT], dwB[VR_DWT], dwR[VR_DWT];
#pragma unroll
    for (int q = 0; q < VR_DWT; ++q) dwA[q] = dwB[q] = dwR[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int l_other = -1;
    bool dw_reg = true;
    {
        auto tiles = [](int N, int K) { return ((N + 15) >> 4) * ((K + 15) >> 4); };
        for (int l = 0; l < L; ++l) {
            if (a.gW_off[l] == a.gW_off[0]) { dw_reg = dw_reg && a.dout[l] == a.dout[0] && a.din[l] == a.din[0]; continue; }
            if (l_other < 0) l_other = l;
            dw_reg = dw_reg && a.gW_off[l] == a.gW_off[l_other] && a.dout[l] == a.dout[l_other] && a.din[l] == a.din[l_other];
        }
        dw_reg = dw_reg && tiles(a.dout[0], a.din[0]) <= VR_NW * VR_DWT && (l_other < 0 || tiles(a.dout[l_other], a.din[l_other]) <= VR_NW * VR_DWT) &&
                 (!gru || tiles(d0, d0) <= VR_NW * VR_DWT);
    }
    for (int rstep = 0; rstep < n_rsteps; ++rstep) {
        const int n = a.T - 1 - rstep, m = n_rsteps - 1 - rstep;   // default grid: reverse step n -> n-1; m = forward step transposed
        const int* pstep = planned ? (disc ? pfwd + m * pw_ : padj + rstep * pw_) : nullptr;
        const float dt = planned ? __int_as_float(pstep[0]) : 1.0f;
        for (int j = 0; j < S; ++j) {
            StageDesc sd;
            if (planned) sd = plan_stage(pstep, disc ? S - 1 - j : j);
            else sd = default_stage(a.method, disc ? (float)(n - 1) + stage_offset(a.method, S - 1 - j) : -(-(float)n + stage_offset(a.method, j)), a.n_pieces);
            const float w = disc ? 1.0f : stage_weight(a.method, j) * dt;
            vr_load_cin(a, b0, sd, a.field_input == NCDE_INPUT_EVALUATE, CIN, Cp, tid);
            if (disc) {
                const float* rec = a.stages + ((long long)(m * S + (S - 1 - j)) * a.B + b0) * H;
                for (int e = tid; e < 16 * H; e += VR_THREADS) {
                    const int s = e / H, h = e - s * H;
                    U[h * 16 + s] = b0 + s < a.B ? rec[e] : 0.0f;
                }
            }
            __syncthreads();
            if (!matmul) {
                for (int e = tid; e < C * 16; e += VR_THREADS) U[H * 16 + e] = CIN[e];
                __syncthreads();
            }
            if (gru) {
                vr_dense<1>(a.Wr, a.br, d0, d0, U, RG, wave, lane, wl_r);
                __syncthreads();
                for (int e = tid; e < ru16(d0) * 16; e += VR_THREADS) RU[e] = RG[e] * U[e];
                __syncthreads();
            }
            const float* xi = vr_net(a, U, XI, DS, true, wave, lane, lds);
            const float* xr = gru ? vr_net(a, RU, XR, DS, true, wave, lane, lds) : xi;
            // ---- heads: f, cotangents of the two pre-activations, head parameter gradients, partials of dL/dx_L ------
            f32x4 accI[VR_MAXJT], accR[VR_MAXJT];
#pragma unroll
            for (int jt = 0; jt < VR_MAXJT; ++jt) accI[jt] = accR[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int grp = wave; grp < ngrp; grp += VR_NW) {
                float ksum = 0.0f;
                for (int qi = 0; qi < per_grp; ++qi) {
                    int cq;
                    const HeadTile ht = vr_head_tile(a, grp * per_grp + qi, ncq, li, lk, cq);
                    const f32x4 pt = vr_head_gemm(a.Wo, a.bo, ht, dlast, xr, li, lk, wl_o);
                    f32x4 ps = pt;
                    if (gated) ps = vr_head_gemm(a.Wg, a.bg, ht, dlast, xi, li, lk, wl_g);
                    float dPt[4], dPs[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float th = tanh_dev(pt[r]);
                        const float sg = gated ? sigmoid_dev(ps[r]) : 1.0f;
                        const float m = gated ? sg * th : th;
                        float dm;
                        if (matmul) {
                            const float dx = CIN[(4 * cq + r) * 16 + li];
                            ksum = fmaf(m, dx, ksum);
                            dm = (ht.hD[r] < Hp ? AS[ht.hD[r] * 16 + li] : 0.0f) * dx;
                        } else {
                            if (ht.hD[r] < Hp) KOY[ht.hD[r] * 16 + li] = ht.rowD[r] >= 0 ? m : 0.0f;
                            dm = ht.hD[r] < Hp ? AS[ht.hD[r] * 16 + li] : 0.0f;
                        }
                        if (ht.rowD[r] < 0) dm = 0.0f;
                        dPt[r] = gated ? (dm * sg) * (1.0f - th * th) : dm * (1.0f - th * th);
                        dPs[r] = gated ? (dm * th) * (sg * (1.0f - sg)) : 0.0f;
                    }
                    // dL/dx_L of the pass feeding each head: rows of the tile are the K dim (k-step r, k-sub lane>>4)
#pragma unroll
                    for (int jt = 0; jt < VR_MAXJT; ++jt) {
                        if (jt < njt) {
                            const int jcol = 16 * jt + li;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const bool ok = ht.rowD[r] >= 0 && jcol < dlast;
                                const float avt = ok ? (wl_o ? wl_o[ht.rowD[r] * (ru16(dlast) + 1) + jcol] : a.Wo[(long long)ht.rowD[r] * dlast + jcol]) : 0.0f;
                                if (gru) accR[jt] = mfma16(avt, dPt[r], accR[jt]);
                                else accI[jt] = mfma16(avt, dPt[r], accI[jt]);
                                if (gated) {
                                    const float avs = ok ? (wl_g ? wl_g[ht.rowD[r] * (ru16(dlast) + 1) + jcol] : a.Wg[(long long)ht.rowD[r] * dlast + jcol]) : 0.0f;
                                    accI[jt] = mfma16(avs, dPs[r], accI[jt]);
                                }
                            }
                        }
                    }
                    if (w != 0.0f) {
#pragma unroll
                        for (int hd = 0; hd < 2; ++hd) {
                            if (hd == 1 && !gated) break;
                            const float* dP = hd == 0 ? dPt : dPs;
                            const float* xl = hd == 0 ? xr : xi;
                            float* gWh = gacc + (hd == 0 ? a.gWo_off : a.gWg_off);
                            float* gbh = gacc + (hd == 0 ? a.gbo_off : a.gbg_off);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float sum = row16_sum(dP[r]);
                                if (li == 0 && ht.rowD[r] >= 0) vr_accum(gbh + ht.rowD[r], w * sum, a.gacc_in_lds);
                            }
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                            for (int r = 0; r < 4; ++r) sc[(4 * lk + r) * 17 + li] = w * dP[r];
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            for (int jt = 0; jt < njt; ++jt) {
                                f32x4 g = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                                for (int ks = 0; ks < 4; ++ks) g = mfma16(sc[li * 17 + 4 * ks + lk], xl[(16 * jt + li) * 16 + 4 * ks + lk], g);
                                const int jcol = 16 * jt + li;
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    if (ht.rowD[r] >= 0 && jcol < dlast) vr_accum(gWh + (long long)ht.rowD[r] * dlast + jcol, g[r], a.gacc_in_lds);
                            }
                        }
                    }
                }
                if (matmul && 4 * grp + lk < Hp) KOY[(4 * grp + lk) * 16 + li] = ksum;
            }
            {
                float* pi = PW + wave * DS;
                float* pr = PW + (4 + wave) * DS;
#pragma unroll
                for (int jt = 0; jt < VR_MAXJT; ++jt)
                    if (jt < njt) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            pi[(16 * jt + 4 * lk + r) * 16 + li] = accI[jt][r];
                            if (gru) pr[(16 * jt + 4 * lk + r) * 16 + li] = accR[jt][r];
                        }
                    }
            }
            __syncthreads();
            // ---- each pass of the inner net backwards (dW/db accumulate for both) -----------------------------------------
            const int DL = ru16(dlast) * 16;
            for (int pass = 0; pass < (gru ? 2 : 1); ++pass) {
                const float* Xp = pass == 0 ? XI : XR;
                const float* pw = PW + pass * 4 * DS;
                const float* x0 = pass == 0 ? U : RU;
                float* du = pass == 0 ? DUI : DUR;
                const float* xl = Xp + (L - 1) * DS;
                for (int e = tid; e < DL; e += VR_THREADS) {
                    const float gsum = (pw[e] + pw[DS + e]) + (pw[2 * DS + e] + pw[3 * DS + e]);
                    GA[e] = xl[e] > 0.0f ? gsum : 0.0f;
                }
                __syncthreads();
                float* gpre = GA;
                float* gx = GB;
                for (int l = L - 1; l >= 0; --l) {
                    const int N = a.dout[l], K = a.din[l];
                    const float* xin = l == 0 ? x0 : Xp + (l - 1) * DS;
                    if (w != 0.0f) {
                        if (!dw_reg) vr_dw_acc(gpre, xin, N, K, w, gacc + a.gW_off[l], gacc + a.gb_off[l], tid, wave, lane, a.gacc_in_lds);
                        else if (a.gW_off[l] == a.gW_off[0]) vr_dw_acc_reg(gpre, xin, N, K, w, dwA, gacc + a.gb_off[l], tid, wave, lane, a.gacc_in_lds);
                        else vr_dw_acc_reg(gpre, xin, N, K, w, dwB, gacc + a.gb_off[l], tid, wave, lane, a.gacc_in_lds);
                    }
                    vr_bwd_data(a.W[l], N, K, gpre, l > 0 ? xin : nullptr, l == 0 ? du : gx, false, wave, lane, a.wres[l] >= 0 ? lds + a.wres[l] : nullptr);
                    __syncthreads();
                    float* tmp = gpre; gpre = gx; gx = tmp;
                }
            }
            if (gru) {
                // u enters directly and through rg * u: du += dru * rg;  cotangent of the reset pre-activation
                for (int e = tid; e < ru16(d0) * 16; e += VR_THREADS) {
                    const float rg = RG[e], u = U[e], dru = DUR[e];
                    DUI[e] += dru * rg;
                    GA[e] = (dru * u) * (rg * (1.0f - rg));
                }
                __syncthreads();
                if (w != 0.0f) {
                    if (dw_reg) vr_dw_acc_reg(GA, U, d0, d0, w, dwR, gacc + a.gbr_off, tid, wave, lane, a.gacc_in_lds);
                    else vr_dw_acc(GA, U, d0, d0, w, gacc + a.gWr_off, gacc + a.gbr_off, tid, wave, lane, a.gacc_in_lds);
                }
                vr_bwd_data(a.Wr, d0, d0, GA, nullptr, DUI, true, wave, lane, wl_r);
                __syncthreads();
            }
            // ---- Butcher bookkeeping (KOA = dL/dy of the stage = the first H rows of du) -----------------------------------
            for (int e = tid; e < HS; e += VR_THREADS) {
                const int h = e >> 4, s = e & 15, b = b0 + s;
                const bool valid = h < H && b < a.B;
                const float d = h < H ? DUI[e] : 0.0f;
                if (disc) {
                    float a0 = A0[e];
                    bool last = false;
                    float next = 0.0f;
                    if (a.method == NCDE_RK4_38) {
                        const float c4 = (a0 * dt) * 0.125f;
                        const float dt3 = dt * 0.333333343267440796f;
                        if (j == 0) { KA1[e] = d; next = 3.0f * c4 + dt * d; }
                        else if (j == 1) { KA2[e] = d; next = (3.0f * c4 - dt * KA1[e]) + dt * d; }
                        else if (j == 2) { KY1[e] = d; next = ((c4 + dt * KA1[e]) - dt3 * KA2[e]) + dt3 * d; }
                        else { a0 = (((a0 + KA1[e]) + KA2[e]) + KY1[e]) + d; last = true; }
                    } else if (a.method == NCDE_MIDPOINT) {
                        if (j == 0) { KA1[e] = d; next = (0.5f * dt) * d; }
                        else { a0 = (a0 + KA1[e]) + d; last = true; }
                    } else {
                        a0 = a0 + d; last = true;
                    }
                    if (last) {
                        float dtp = 1.0f;
                        if (planned) {
                            if (valid) {
                                const long long brow = (long long)b * a.n_out;
                                a0 = a0 + plan_out_cotangent(a, pstep, pout, 0, brow, h);
                                if (m > 0) a0 = a0 + plan_out_cotangent(a, pstep - pw_, pout, 1, brow, h);
                                else a0 = a0 + a.grad_out[brow * H + h];
                            }
                            if (m > 0) dtp = __int_as_float(pstep[-pw_]);
                        } else if (a.output == NCDE_OUT_KNOTS || n == 1) {
                            a0 = a0 + (valid ? a.grad_out[((long long)b * a.n_out + (a.output == NCDE_OUT_KNOTS ? n - 1 : 0)) * H + h] : 0.0f);
                        }
                        A0[e] = a0;
                        next = a.method == NCDE_RK4_38 ? (a0 * dtp) * 0.125f : dtp * a0;
                        if (m == 0 && valid) a.grad_z0[(long long)b * H + h] = a0;
                    }
                    AS[e] = next;
                } else {
                    float y0 = Y0[e], k1 = KY1[e], k2 = KY2[e];
                    bool last;
                    const float ys = StageCombine::apply(a.method, j, -KOY[e], dt, y0, k1, k2, last);
                    U[e] = ys; KY1[e] = k1; KY2[e] = k2;
                    float a0 = A0[e], q1 = KA1[e], q2 = KA2[e];
                    const float as = StageCombine::apply(a.method, j, d, dt, a0, q1, q2, last);
                    KA1[e] = q1; KA2[e] = q2;
                    if (!last) {
                        AS[e] = as;
                    } else {
                        if (planned) {
                            const int row = pstep[1];
                            if (row >= 0) {
                                const long long o = ((long long)b * a.n_out + row) * H + h;
                                y0 = valid ? a.z_out[o] : 0.0f;
                                a0 = a0 + (valid ? a.grad_out[o] : 0.0f);
                            }
                        } else if (a.output == NCDE_OUT_KNOTS) {
                            const long long o = ((long long)b * a.n_out + (n - 1)) * H + h;
                            y0 = valid ? a.z_out[o] : 0.0f;
                            a0 = a0 + (valid ? a.grad_out[o] : 0.0f);
                        } else if (n == 1) {
                            a0 = a0 + (valid ? a.grad_out[((long long)b * a.n_out) * H + h] : 0.0f);
                        }
                        Y0[e] = y0; U[e] = y0; A0[e] = a0; AS[e] = a0;
                        if (rstep == n_rsteps - 1 && valid) a.grad_z0[(long long)b * H + h] = a0;
                    }
                }
            }
            __syncthreads();
        }
    }
    if (dw_reg) {      // the register-held weight gradients join the partial (each element has one owner: plain adds)
        vr_dw_flush(dwA, a.dout[0], a.din[0], gacc + a.gW_off[0], wave, lane);
        if (l_other >= 0) vr_dw_flush(dwB, a.dout[l_other], a.din[l_other], gacc + a.gW_off[l_other], wave, lane);
        if (gru) vr_dw_flush(dwR, d0, d0, gacc + a.gWr_off, wave, lane);
        __syncthreads();
    }
    if (a.gacc_in_lds) {
        float* dst = a.gpart + (long long)blockIdx.x * a.theta_size;
        for (int e = tid; e < a.theta_size; e += VR_THREADS) dst[e] = GL[e];
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
namespace {

struct VrRes {
    int w[NCDE_MAX_LAYERS], o, g, r;   // float offsets of the resident copies, -1 = streamed
};
struct VrPlan {
    size_t lds_fwd, lds_adj;
    int gacc_in_lds;
    VrRes res_fwd, res_adj;
};

// Which weight matrices get an LDS-resident copy (row stride K + 1): whatever fits behind the activations -- reset gate,
// inner layers (a shared layer once), then the heads.
VrRes vr_residency(const NcdeProblem* p, const Layout& y, size_t& bytes) {
    VrRes r;
    for (int l = 0; l < NCDE_MAX_LAYERS; ++l) r.w[l] = -1;
    r.o = r.g = r.r = -1;
    size_t off = bytes / sizeof(float);
    auto ru16h = [](int v) { return (v + 15) & ~15; };
    auto take = [&](int N, int K) -> int {
        const size_t need = (size_t)ru16h(N) * (ru16h(K) + 1);
        if ((off + need) * sizeof(float) > (size_t)kLdsLimit) return -1;
        const int at = (int)off;
        off += need;
        return at;
    };
    const int d0 = p->field_input == NCDE_INPUT_MATMUL ? p->hidden : p->hidden + p->channels;
    const int rows = p->field_input == NCDE_INPUT_MATMUL ? p->hidden * p->channels : p->hidden;
    if (p->field_kind == NCDE_FIELD_GRU) r.r = take(d0, d0);
    for (int l = 0; l < p->n_layers; ++l) {
        int shared = -1;
        for (int q = 0; q < l; ++q)
            if (p->layer_W[q] == p->layer_W[l]) shared = q;
        r.w[l] = shared >= 0 ? r.w[shared] : take(p->layer_out[l], p->layer_in[l]);
    }
    r.o = take(rows, y.dlast);
    if (p->field_kind != NCDE_FIELD_ORIGINAL) r.g = take(rows, y.dlast);
    bytes = off * sizeof(float);
    return r;
}

VrPlan vr_plan(const NcdeProblem* p, const Layout& y) {
    VrPlan v{};
    const size_t HS = (size_t)y.Hp * 16, DS = (size_t)y.Dp * 16, L = (size_t)p->n_layers;
    v.lds_fwd = sizeof(float) * (7 * DS + 4 * HS + (size_t)y.Cp * 16);
    const size_t adj = sizeof(float) * ((15 + 2 * L) * DS + 9 * HS + (size_t)y.Cp * 16 + VR_NW * 16 * 17);
    v.gacc_in_lds = adj + sizeof(float) * (size_t)y.theta_size <= (size_t)kLdsLimit;
    v.lds_adj = adj + (v.gacc_in_lds ? sizeof(float) * (size_t)y.theta_size : 0);
    if (v.lds_fwd <= (size_t)kLdsLimit) v.res_fwd = vr_residency(p, y, v.lds_fwd);
    if (v.lds_adj <= (size_t)kLdsLimit) v.res_adj = vr_residency(p, y, v.lds_adj);
    return v;
}

}  // namespace

bool ncde_variant_supported(const NcdeProblem* p, int pass) {
    const Layout y = make_layout(p);
    if (p->n_layers < 1 || y.dlast > 16 * VR_MAXJT) return false;
    const VrPlan v = vr_plan(p, y);
    return pass == 0 ? v.lds_fwd <= (size_t)kLdsLimit : v.lds_adj <= (size_t)kLdsLimit;
}

int64_t ncde_variant_workspace_bytes(const NcdeProblem* p, int pass) {
    if (!ncde_variant_supported(p, pass)) return NCDE_ERR_UNSUPPORTED;
    if (pass == 0) return 256;
    const Layout y = make_layout(p);
    return (int64_t)sizeof(float) * (int64_t)y.n_wg * (int64_t)y.theta_size + 256;
}

int ncde_variant_forward(const NcdeProblem* p, float* out, float* stages, hipStream_t st) {
    if (!ncde_variant_supported(p, 0)) return NCDE_ERR_UNSUPPORTED;
    const Layout y = make_layout(p);
    const VrPlan v = vr_plan(p, y);
    KArgs a;
    fill_kargs(p, y, &a);
    a.out = out;
    a.stages = stages;
    for (int l = 0; l < NCDE_MAX_LAYERS; ++l) a.wres[l] = v.res_fwd.w[l];
    a.wres_o = v.res_fwd.o; a.wres_g = v.res_fwd.g; a.wres_r = v.res_fwd.r;
    if (ncde_lds_optin((const void*)ncde_fwd_variant, v.lds_fwd) != hipSuccess) return NCDE_ERR_HIP;
    hipLaunchKernelGGL(ncde_fwd_variant, dim3(y.n_wg), dim3(VR_THREADS), v.lds_fwd, st, a);
    return hipGetLastError() == hipSuccess ? NCDE_OK : NCDE_ERR_HIP;
}

int ncde_variant_adjoint(const NcdeProblem* p, const float* src, const float* grad_out, const NcdeGrads* g, void* ws, hipStream_t st,
                         bool main_kernel_only, bool discrete) {
    if (!ncde_variant_supported(p, 1)) return NCDE_ERR_UNSUPPORTED;
    const Layout y = make_layout(p);
    const VrPlan v = vr_plan(p, y);
    KArgs a;
    fill_kargs(p, y, &a);
    a.grad_out = grad_out; a.grad_z0 = g->grad_z0;
    if (discrete) { a.stages = const_cast<float*>(src); a.discrete = 1; }
    else a.z_out = src;
    a.gpart = (float*)ws;
    a.gacc_in_lds = v.gacc_in_lds;
    for (int l = 0; l < NCDE_MAX_LAYERS; ++l) a.wres[l] = v.res_adj.w[l];
    a.wres_o = v.res_adj.o; a.wres_g = v.res_adj.g; a.wres_r = v.res_adj.r;
    if (ncde_lds_optin((const void*)ncde_adj_variant, v.lds_adj) != hipSuccess) return NCDE_ERR_HIP;
    hipLaunchKernelGGL(ncde_adj_variant, dim3(y.n_wg), dim3(VR_THREADS), v.lds_adj, st, a);
    if (hipGetLastError() != hipSuccess) return NCDE_ERR_HIP;
    if (main_kernel_only) return NCDE_OK;
    return launch_reduce_partials(p, y, g, (const float*)ws, y.n_wg, st);
}
