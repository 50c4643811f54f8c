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

#include <cstring>

#include "ncde_common.h"

namespace {

template <int METHOD>
struct Combine {
    // Butcher bookkeeping; returns the next stage input (or the new state after the last stage).
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

template <int METHOD> constexpr int kStages = METHOD == NCDE_RK4_38 ? 4 : (METHOD == NCDE_MIDPOINT ? 2 : 1);

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int H, int HH, int C, int NW, int INTERP, int METHOD>
__global__ __launch_bounds__(64 * NW, 1) void ncde_fwd_fast(KArgs a) {
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
            for (int ks = 0; ks < KH; ++ks) wo[nb][cq][ks] = cA < C ? a.Wo[(hA * C + cA) * HH + 4 * ks + g] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 4 * cq + r;
                biaso[nb][cq][r] = c < C ? a.bo[(4 * hb + g) * C + c] : 0.0f;
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
        eok[q] = e < 16 * DXW && c < C && (b0 + es) < a.B;
        const long long base = (long long)(eok[q] ? b0 + es : 0) * a.cs_b;
        eptr[q] = a.coeffs + base + (INTERP == NCDE_INTERP_LINEAR ? c : (part + 1) * C + c);
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
    for (int ks = 0; ks < HB; ++ks) zreg[ks] = valid ? a.z0[(long long)bs * H + 4 * ks + g] : 0.0f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        y0[nb] = valid ? a.z0[(long long)bs * H + 4 * (wave * NB + nb) + g] : 0.0f;
        k1[nb] = 0.0f;
        k2[nb] = 0.0f;
        if (valid) a.out[((long long)bs * a.n_out) * H + 4 * (wave * NB + nb) + g] = y0[nb];
    }
    __syncthreads();

    const int n_inner = a.n_layers - 1;
    int zpar = 0;
    for (int n = 0; n < a.T - 1; ++n) {
        if (n + 1 < a.n_pieces) stage_load(n + 1);  // prefetch next piece; consumed at the end of the step
#pragma unroll
        for (int j = 0; j < S; ++j) {
            const float t = (float)n + stage_offset(METHOD, j);
            const int idx = piece_index(t, a.n_pieces);
            const float frac = t - (float)idx;
            const float* dxp = dxs[idx % 3] + s * DXW;
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
                for (int r = 0; r < 4; ++r) hB[4 * tt + r] = fmaxf(acc[tt][r], 0.0f);
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
                    for (int r = 0; r < 4; ++r) hB[4 * tt + r] = fmaxf(acc[tt][r], 0.0f);
            }
            // ---- output layer tiles owned by this wave: tanh + channel contraction -------------------
            float kout[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) kout[nb] = 0.0f;
#pragma unroll
            for (int cq = 0; cq < CQ; ++cq) {
                f32x4 o[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) o[nb] = biaso[nb][cq];
#pragma unroll
                for (int ks = 0; ks < KH; ++ks)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) o[nb] = mfma16(wo[nb][cq][ks], hB[ks], o[nb]);
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
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) kout[nb] = fmaf(tanh_dev(o[nb][r]), dx[r], kout[nb]);
            }
            // ---- Butcher bookkeeping for the owned state entries, then exchange the stage input ------
            float ys[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) ys[nb] = Combine<METHOD>::apply(j, kout[nb], y0[nb], k1[nb], k2[nb]);
            if (j == S - 1) {
                if (valid && (a.output == NCDE_OUT_KNOTS || n == a.T - 2)) {
                    const int row = a.output == NCDE_OUT_KNOTS ? n + 1 : 1;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) a.out[((long long)bs * a.n_out + row) * H + 4 * (wave * NB + nb) + g] = ys[nb];
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
        }
    }
}

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

struct FastEntry {
    Shape shape;
    int nw;
    FwdFn (*fwd)(int, int);
    const char* fwd_name;
};

const FastEntry kFast[] = {
    {{32, 32, 20}, 4, pick_fwd<32, 32, 20, 4>, "ncde_fwd_fast<H32,HH32,C20,NW4>"},   // BASELINE cfg2 / cfg3
    {{64, 64, 4}, 4, pick_fwd<64, 64, 4, 4>, "ncde_fwd_fast<H64,HH64,C4,NW4>"},       // BASELINE cfg4
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

void fill_kargs_fast(const NcdeProblem* p, KArgs* a) {
    memset(a, 0, sizeof(*a));
    a->B = p->batch; a->T = p->n_knots; a->C = p->channels; a->H = p->hidden;
    a->interp = p->interp; a->method = p->method; a->output = p->output; a->n_layers = p->n_layers;
    a->n_pieces = p->n_knots - 1;
    a->n_out = p->output == NCDE_OUT_KNOTS ? p->n_knots : 2;
    for (int l = 0; l < p->n_layers; ++l) {
        a->din[l] = p->layer_in[l]; a->dout[l] = p->layer_out[l];
        a->W[l] = p->layer_W[l]; a->b[l] = p->layer_b[l];
    }
    a->Wo = p->Wo; a->bo = p->bo; a->coeffs = p->coeffs;
    a->cs_b = p->coeffs_stride_b; a->cs_t = p->coeffs_stride_t;
    a->z0 = p->z0;
}

}  // namespace

bool ncde_fast_supported(const NcdeProblem* p, int pass) {
    if (pass != 0) return false;
    return find_entry(p) != nullptr;
}

const char* ncde_fast_kernel_name(const NcdeProblem* p, int pass) {
    const FastEntry* e = find_entry(p);
    if (!e || pass != 0) return nullptr;
    return e->fwd_name;
}

int64_t ncde_fast_workspace_bytes(const NcdeProblem* p, int pass) {
    (void)p;
    return pass == 0 ? 256 : NCDE_ERR_UNSUPPORTED;
}

int ncde_fast_forward(const NcdeProblem* p, float* out, void* ws, size_t ws_bytes, hipStream_t st) {
    (void)ws; (void)ws_bytes;
    const FastEntry* e = find_entry(p);
    if (!e) return NCDE_ERR_UNSUPPORTED;
    FwdFn fn = e->fwd(p->interp, p->method);
    if (!fn) return NCDE_ERR_UNSUPPORTED;
    KArgs a;
    fill_kargs_fast(p, &a);
    a.out = out;
    const int n_wg = (p->batch + NCDE_TILE - 1) / NCDE_TILE;
    hipLaunchKernelGGL(fn, dim3(n_wg), dim3(64 * e->nw), 0, st, a);
    return hipGetLastError() == hipSuccess ? NCDE_OK : NCDE_ERR_HIP;
}

int ncde_fast_adjoint(const NcdeProblem*, const float*, const float*, const NcdeGrads*, void*, size_t, hipStream_t, bool) {
    return NCDE_ERR_UNSUPPORTED;
}
