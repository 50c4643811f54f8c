// Stage-level building blocks of the generic (runtime-dimension) family, shared by the persistent fixed-step kernels
// (ncde_generic.hip) and the adaptive dopri5 kernels (ncde_adaptive.hip).  One workgroup = one tile of 16 samples, 4 waves;
// every activation lives in LDS as [unit][sample] (see ncde_generic.hip for the layout rationale).
#pragma once
#include "ncde_common.h"

#define GEN_NW 4
#define GEN_THREADS (64 * GEN_NW)
#define GEN_MAXJT 8  // hidden widths up to 128 in the adjoint kernel

namespace {

__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// dX/dt(t) for the 16 samples of the tile -> DX[c*16 + s]  (rows c >= C and samples >= B are zero)
__device__ void load_dx(const KArgs& a, int b0, const StageDesc& sd, float* DX, int Cp, int tid) {
    const int idx = sd.idx;
    const float frac = sd.frac;
    for (int e = tid; e < 16 * Cp; e += GEN_THREADS) {
        const int s = e / Cp, c = e - s * Cp;
        const int b = b0 + s;
        float v = 0.0f;
        if (c < a.C && b < a.B) {
            const float* p = a.coeffs + (long long)b * a.cs_b + (long long)idx * a.cs_t;
            if (a.interp == NCDE_INTERP_LINEAR) {
                v = p[a.cs_t + c] - p[c];
                if (sd.kdt != 1.0f) v = v / sd.kdt;   // user knot grid: (c[i+1]-c[i]) / (t[i+1]-t[i]), interpolation_linear.py:198
            } else {
                const float bb = p[a.C + c], cc = p[2 * a.C + c], dd = p[3 * a.C + c];
                const float inner = cc + dd * frac;
                v = bb + inner * frac;
            }
        }
        DX[c * 16 + s] = v;
    }
}

// out[n][s] = relu(sum_k W[n][k] in[k][s] + bias[n]) for n < ru16(N) (rows >= N come out as 0).
__device__ void dense_relu(const float* __restrict__ W, const float* __restrict__ bias, int N, int K,
                           const float* in, float* out, int wave, int lane) {
    const int li = lane & 15, lk = lane >> 4;
    const int ntiles = (N + 15) >> 4, nks = (K + 3) >> 2;
    for (int t = wave; t < ntiles; t += GEN_NW) {
        const int rowA = 16 * t + li;
        const bool rv = rowA < N;
        const float* wrow = W + (long long)(rv ? rowA : 0) * K;
        f32x4 acc;
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
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(16 * t + 4 * lk + r) * 16 + li] = fmaxf(acc[r], 0.0f);
    }
}


// One evaluation of f_theta(YS) . dX/dt for the tile: hidden layers (relu) -> output tiles -> tanh -> channel
// contraction; KO[h][s] receives the stage derivative.  Ends with a workgroup barrier.
// (src/ncde/vector_fields/base.py:83-104, modules/torchcde/torchcde/solver.py:112-137)
__device__ void gen_stage_forward(const KArgs& a, const float* YS, float* ACT0, float* ACT1, const float* DX, float* KO, int Hp,
                                  int Cp, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int H = a.H, C = a.C;
    const int dlast = a.n_layers ? a.dout[a.n_layers - 1] : H;
    const int nks_o = (dlast + 3) >> 2, nhb = Hp >> 2, ncq = Cp >> 2;
    const float* in = YS;
    for (int l = 0; l < a.n_layers; ++l) {
        float* outb = (l & 1) ? ACT1 : ACT0;
        dense_relu(a.W[l], a.b[l], a.dout[l], a.din[l], in, outb, wave, lane);
        __syncthreads();
        in = outb;
    }
    // output layer + tanh + channel contraction; tile rows (g, r) <-> (h = 4hb+g, c = 4cq+r)
    for (int hb = wave; hb < nhb; hb += GEN_NW) {
        float kacc = 0.0f;
        const int hA = 4 * hb + (li >> 2), hD = 4 * hb + lk;
        for (int cq = 0; cq < ncq; ++cq) {
            const int cA = 4 * cq + (li & 3);
            const bool rv = hA < H && cA < C;
            const float* wrow = a.Wo + (long long)(rv ? hA * C + cA : 0) * dlast;
            f32x4 acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 4 * cq + r;
                acc[r] = (hD < H && c < C) ? a.bo[hD * C + c] : 0.0f;
            }
#pragma unroll 4
            for (int ks = 0; ks < nks_o; ++ks) {
                const int k = 4 * ks + lk;
                const float av = (rv && k < dlast) ? wrow[k] : 0.0f;
                acc = mfma16(av, in[k * 16 + li], acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) kacc = fmaf(tanh_dev(acc[r]), DX[(4 * cq + r) * 16 + li], kacc);
        }
        KO[hD * 16 + li] = kacc;
    }
    __syncthreads();
}

// One stage of the reverse sweep for the tile: recompute the forward at YS keeping x_1..x_L, then the VJP with cotangent AS:
//   KOY[h][s] = f(YS).dX,  KOA[h][s] = (AS^T df/dy)[h][s],  gacc[theta] += w * AS^T df/dtheta  (skipped when w == 0).
// D2X != NULL (cubic control in the adaptive adjoint): additionally gacc[a.theta_size] += w * sum_{s,h,c} AS[h][s] M[h][c][s]
// D2X[c][s], the time component vjp_t of the reference's augmented state (adjoint.py:75-98).
// (adjoint.py:73-106: func forward + autograd.grad wrt y and the parameters)
__device__ void gen_stage_vjp(const KArgs& a, const float* YS, const float* AS, const float* DX, const float* D2X, float* X, float* G0,
                              float* G1, float* PW2, float* SC, float* KOY, float* KOA, float* gacc, float w, int Hp, int Cp, int DS,
                              int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int H = a.H, C = a.C, L = a.n_layers, HS = Hp * 16;
    const int dlast = L ? a.dout[L - 1] : H;
    const int nks_o = (dlast + 3) >> 2, nhb = Hp >> 2, ncq = Cp >> 2, njt = (dlast + 15) >> 4;
    float* sc = SC + wave * 16 * 17;
    float vt = 0.0f;
    // ---- recompute the stage forward, keeping x_1..x_L -------------------------------------
    const float* in = YS;
    for (int l = 0; l < L; ++l) {
        float* outb = X + l * DS;
        dense_relu(a.W[l], a.b[l], a.dout[l], a.din[l], in, outb, wave, lane);
        __syncthreads();
        in = outb;
    }
    // ---- output layer: f, dP = a (x) dX * tanh', dbo, dWo, partial dL/dx_L -----------------
    f32x4 accJ[GEN_MAXJT];
#pragma unroll
    for (int jt = 0; jt < GEN_MAXJT; ++jt) accJ[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int hb = wave; hb < nhb; hb += GEN_NW) {
        float kacc = 0.0f;
        const int hA = 4 * hb + (li >> 2), hD = 4 * hb + lk;
        const float aval = AS[hD * 16 + li];
        for (int cq = 0; cq < ncq; ++cq) {
            const int cA = 4 * cq + (li & 3);
            const bool rv = hA < H && cA < C;
            const float* wrow = a.Wo + (long long)(rv ? hA * C + cA : 0) * dlast;
            f32x4 acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 4 * cq + r;
                acc[r] = (hD < H && c < C) ? a.bo[hD * C + c] : 0.0f;
            }
#pragma unroll 4
            for (int ks = 0; ks < nks_o; ++ks) {
                const int k = 4 * ks + lk;
                const float av = (rv && k < dlast) ? wrow[k] : 0.0f;
                acc = mfma16(av, in[k * 16 + li], acc);
            }
            float dP[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float m = tanh_dev(acc[r]);
                const float dx = DX[(4 * cq + r) * 16 + li];
                kacc = fmaf(m, dx, kacc);
                dP[r] = (aval * dx) * (1.0f - m * m);
                if (D2X) vt = fmaf(aval * m, D2X[(4 * cq + r) * 16 + li], vt);
            }
            // data gradient: dL/dx_L[j][s] += sum_u Wo[u][j] dP[u][s]; k-step <-> r, k-sub <-> lane>>4
#pragma unroll
            for (int jt = 0; jt < GEN_MAXJT; ++jt) {
                if (jt < njt) {
                    const int jcol = 16 * jt + li;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = 4 * cq + r;
                        const float av = (hD < H && c < C && jcol < dlast) ? a.Wo[(long long)(hD * C + c) * dlast + jcol] : 0.0f;
                        accJ[jt] = mfma16(av, dP[r], accJ[jt]);
                    }
                }
            }
            if (w != 0.0f) {
                // bias gradient: sum over the 16 samples of the tile
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sum = row16_sum(dP[r]);
                    const int c = 4 * cq + r;
                    if (li == 0 && hD < H && c < C) gacc[a.gbo_off + hD * C + c] += w * sum;
                }
                // weight gradient: dWo[u][j] += w sum_s dP[u][s] x_L[j][s] (samples are the K dim)
                wave_lds_fence();
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[(4 * lk + r) * 17 + li] = w * dP[r];
                wave_lds_fence();
                for (int jt = 0; jt < njt; ++jt) {
                    f32x4 g = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) g = mfma16(sc[li * 17 + 4 * ks + lk], in[(16 * jt + li) * 16 + 4 * ks + lk], g);
                    const int jcol = 16 * jt + li;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = 4 * cq + r;
                        if (hD < H && c < C && jcol < dlast) gacc[a.gWo_off + (long long)(hD * C + c) * dlast + jcol] += g[r];
                    }
                }
            }
        }
        KOY[hD * 16 + li] = kacc;
    }
    // per-wave partial of dL/dx_L -> LDS, then summed over the 4 waves
    {
        float* pw = wave == 0 ? G0 : (wave == 1 ? G1 : PW2 + (wave - 2) * DS);
#pragma unroll
        for (int jt = 0; jt < GEN_MAXJT; ++jt)
            if (jt < njt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pw[(16 * jt + 4 * lk + r) * 16 + li] = accJ[jt][r];
            }
    }
    __syncthreads();
    const int DL = ru16(dlast) * 16;
    if (L == 0) {
        for (int e = tid; e < HS; e += GEN_THREADS) KOA[e] = (G0[e] + G1[e]) + (PW2[e] + PW2[DS + e]);
        __syncthreads();
    } else {
        // dL/dpre_L = dL/dx_L * relu'(x_L)
        for (int e = tid; e < DL; e += GEN_THREADS) {
            const float gsum = (G0[e] + G1[e]) + (PW2[e] + PW2[DS + e]);
            G1[e] = in[e] > 0.0f ? gsum : 0.0f;
        }
        __syncthreads();
        float* gpre = G1;
        float* gx = G0;
        for (int l = L - 1; l >= 0; --l) {
            const int N = a.dout[l], K = a.din[l];
            const float* xin = l == 0 ? YS : X + (l - 1) * DS;
            if (w != 0.0f) {
                // bias gradient
                for (int jj = tid; jj < N; jj += GEN_THREADS) {
                    float sum = 0.0f;
#pragma unroll
                    for (int s = 0; s < 16; ++s) sum += gpre[jj * 16 + s];
                    gacc[a.gb_off[l] + jj] += w * sum;
                }
                // weight gradient tiles (jt, it)
                const int njt_l = (N + 15) >> 4, nit_l = (K + 15) >> 4;
                for (int tt = wave; tt < njt_l * nit_l; tt += GEN_NW) {
                    const int jt = tt / nit_l, it = tt - jt * nit_l;
                    f32x4 g = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        g = mfma16(gpre[(16 * jt + li) * 16 + 4 * ks + lk], xin[(16 * it + li) * 16 + 4 * ks + lk], g);
                    const int col = 16 * it + li;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * jt + 4 * lk + r;
                        if (row < N && col < K) gacc[a.gW_off[l] + (long long)row * K + col] += w * g[r];
                    }
                }
            }
            // data gradient: dL/dx_{l-1}[i][s] = sum_j W_l[j][i] dpre[j][s]
            float* outb = l == 0 ? KOA : gx;
            const int nit = (K + 15) >> 4, nks = (N + 3) >> 2;
            for (int it = wave; it < nit; it += GEN_NW) {
                const int col = 16 * it + li;
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
                for (int ks = 0; ks < nks; ++ks) {
                    const int k = 4 * ks + lk;
                    const float av = (k < N && col < K) ? a.W[l][(long long)k * K + col] : 0.0f;
                    acc = mfma16(av, gpre[k * 16 + li], acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * it + 4 * lk + r;
                    float v = acc[r];
                    if (l > 0) v = xin[row * 16 + li] > 0.0f ? v : 0.0f;  // fold relu' of x_{l-1}
                    outb[row * 16 + li] = v;
                }
            }
            __syncthreads();
            float* tmp = gpre; gpre = gx; gx = tmp;
        }
    }
    if (D2X && w != 0.0f) {      // workgroup sum of the per-lane time-gradient partials -> the extra slot behind theta
        vt = row16_sum(vt);
        vt += __shfl_xor(vt, 16, 64);
        vt += __shfl_xor(vt, 32, 64);
        if (lane == 0) sc[0] = vt;
        __syncthreads();
        if (tid == 0) gacc[a.theta_size] += w * ((SC[0] + SC[16 * 17]) + (SC[2 * 16 * 17] + SC[3 * 16 * 17]));
        __syncthreads();
    }
}

}  // namespace
