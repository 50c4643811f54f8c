// Batch-tiled Neural-CDE kernels for gfx950: the LARGE-HIDDEN regime (BASELINE config 5: H = HH = 128, C = 80,
// |theta| = 5.4 MB >> LDS), SURVEY.md §7 step 7.
//
// The register-resident family (ncde_fast.hip) needs every weight on chip; the generic family (ncde_generic.hip)
// re-reads every weight once per 16 samples, 4 bytes per lane-load, and is bound by that stream.  Here one workgroup
// (8 waves) owns NS x 16 samples:
//   * every weight fragment is fetched ONCE per stage per workgroup as a 16-byte load per lane (64 contiguous bytes
//     per weight row and k-block, prefetched one k-block ahead) and reused for NS MFMA column tiles, so the L2 /
//     Infinity-Cache weight stream per sample drops by NS x and the load count by 4 NS x;
//   * activations live in LDS as [unit/4][sample][unit%4]: the D registers of a 16x16 MFMA tile are ONE
//     ds_write_b128 per lane and the B operand of four k-steps ONE ds_read_b128 (k = 16 kb + 4 (lane>>4) + e);
//   * the Butcher state (y0, k1, k2 ...) stays in registers, a fixed slice per thread.
// Shapes: H, layer widths multiples of 16, C a multiple of 4 (anything else runs on the generic family).
// Reference semantics restated: see the header of ncde_generic.hip (same op sequence, same citations).
#include "ncde_common.h"
#include "ncde_host.h"
#include "ncde_tiled.h"

#define TL_NW 8
#define TL_THREADS (64 * TL_NW)
#define TL_EMAX 16  // state elements per thread: H * NS * 16 <= TL_EMAX * TL_THREADS

namespace {

// dX/dt(t) of the tile's samples -> DX[(c>>2)][s][c&3]
template <int NS>
__device__ __forceinline__ void tl_load_dx(const KArgs& a, int b0, int idx, float frac, float* DX, int tid) {
    constexpr int NSP = NS * 16;
    const int C = a.C;
    for (int e = tid; e < NSP * C; e += TL_THREADS) {
        const int s = e / C, c = e - s * C;
        const int b = b0 + s;
        float v = 0.0f;
        if (b < a.B) {
            const float* p = a.coeffs + (long long)b * a.cs_b + (long long)idx * a.cs_t;
            if (a.interp == NCDE_INTERP_LINEAR) {
                v = p[a.cs_t + c] - p[c];
            } else {
                const float bb = p[C + c], cc = p[2 * C + c], dd = p[3 * C + c];
                const float inner = cc + dd * frac;
                v = bb + inner * frac;
            }
        }
        DX[((c >> 2) * NSP + s) * 4 + (c & 3)] = v;
    }
}

// One weight panel = 16 rows x PK k-blocks of 16: PK 16-byte loads per lane, all in flight together.
// PK is the largest of {8, 4, 2, 1} dividing K/16, so the panel loops carry no guards.
template <int PK>
struct Panel {
    f32x4 v[PK];
};
template <int PK>
__device__ __forceinline__ Panel<PK> tl_load_panel(const float* wrow, int kb0) {
    Panel<PK> P;
#pragma unroll
    for (int i = 0; i < PK; ++i) P.v[i] = *reinterpret_cast<const f32x4*>(wrow + 16 * (kb0 + i));
    return P;
}
// acc[st] += panel x activations (k-blocks kb0 .. kb0+PK-1 of `in`)
template <int NS, int PK>
__device__ __forceinline__ void tl_mma_panel(const Panel<PK>& P, const float* in, int kb0, int li, int lk, f32x4 (&acc)[NS]) {
    constexpr int NSP = NS * 16;
    f32x4 Bv[PK][NS];
#pragma unroll
    for (int i = 0; i < PK; ++i)
#pragma unroll
        for (int st = 0; st < NS; ++st) Bv[i][st] = *reinterpret_cast<const f32x4*>(in + ((4 * (kb0 + i) + lk) * NSP + st * 16 + li) * 4);
#pragma unroll
    for (int i = 0; i < PK; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int st = 0; st < NS; ++st) acc[st] = mfma16(P.v[i][e], Bv[i][st][e], acc[st]);
}
__host__ __device__ __forceinline__ int tl_panel_k(int nkb) { return (nkb % 8 == 0) ? 8 : ((nkb % 4 == 0) ? 4 : ((nkb % 2 == 0) ? 2 : 1)); }

// out = relu(W in + bias): W [N][K] row-major in global memory, in/out in the LDS layout above.
// The (row tile, panel) pairs of this wave form one sequence; the panel of pair q+1 is in flight while pair q computes.
template <int NS, int PK>
__device__ __forceinline__ void tl_dense_relu_pk(const float* __restrict__ W, const float* __restrict__ bias, int N, int K,
                                                 const float* in, float* out, int wave, int lane) {
    constexpr int NSP = NS * 16;
    const int li = lane & 15, lk = lane >> 4;
    const int npan = (K >> 4) / PK;
    const int ntile = ((N >> 4) - wave + TL_NW - 1) / TL_NW;       // row tiles wave, wave+8, ...
    if (ntile <= 0) return;
    const int nq = ntile * npan;
    auto wrow_of = [&](int q) { return W + (long long)(16 * (wave + TL_NW * (q / npan)) + li) * K + 4 * lk; };
    Panel<PK> Pn = tl_load_panel<PK>(wrow_of(0), 0);
    f32x4 acc[NS];
    for (int q = 0; q < nq; ++q) {
        const int ti = q / npan, pan = q - ti * npan, t = wave + TL_NW * ti;
        const Panel<PK> P = Pn;
        {
            const int qn = q + 1 < nq ? q + 1 : q;
            Pn = tl_load_panel<PK>(wrow_of(qn), (qn % npan) * PK);
        }
        if (pan == 0) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 16 * t + 4 * lk);
#pragma unroll
            for (int st = 0; st < NS; ++st) acc[st] = bv;
        }
        tl_mma_panel<NS, PK>(P, in, pan * PK, li, lk, acc);
        if (pan == npan - 1) {
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = relu_dev(acc[st][r]);
                *reinterpret_cast<f32x4*>(out + ((4 * t + lk) * NSP + st * 16 + li) * 4) = o;
            }
        }
    }
}
template <int NS>
__device__ __forceinline__ void tl_dense_relu(const float* __restrict__ W, const float* __restrict__ bias, int N, int K,
                                              const float* in, float* out, int wave, int lane) {
    switch (tl_panel_k(K >> 4)) {
        case 8: tl_dense_relu_pk<NS, 8>(W, bias, N, K, in, out, wave, lane); break;
        case 4: tl_dense_relu_pk<NS, 4>(W, bias, N, K, in, out, wave, lane); break;
        case 2: tl_dense_relu_pk<NS, 2>(W, bias, N, K, in, out, wave, lane); break;
        default: tl_dense_relu_pk<NS, 1>(W, bias, N, K, in, out, wave, lane); break;
    }
}

// output layer + tanh + channel contraction for the h-blocks of this wave -> KO.  Tile rows (g, r) <-> (h = 4hb+g,
// c = 4cq+r); sequence of (h-block, channel quad, panel) triples, next panel in flight while one computes.
template <int NS, int PK>
__device__ __forceinline__ void tl_output_pk(const KArgs& a, const float* in, const float* DX, float* KO, int dlast, int wave, int lane) {
    constexpr int NSP = NS * 16;
    const int li = lane & 15, lk = lane >> 4;
    const int C = a.C, nhb = a.H >> 2, ncq = C >> 2;
    const int npan = (dlast >> 4) / PK;
    const int nhb_w = (nhb - wave + TL_NW - 1) / TL_NW;
    if (nhb_w <= 0) return;
    const int per_hb = ncq * npan, nq = nhb_w * per_hb;
    auto wrow_of = [&](int q) {
        const int hb = wave + TL_NW * (q / per_hb), cq = (q % per_hb) / npan;
        return a.Wo + (long long)((4 * hb + (li >> 2)) * C + 4 * cq + (li & 3)) * dlast + 4 * lk;
    };
    Panel<PK> Pn = tl_load_panel<PK>(wrow_of(0), 0);
    float kacc[NS];
    f32x4 acc[NS];
    for (int q = 0; q < nq; ++q) {
        const int hi = q / per_hb, rem = q - hi * per_hb, cq = rem / npan, pan = rem - cq * npan;
        const int hb = wave + TL_NW * hi;
        const Panel<PK> P = Pn;
        {
            const int qn = q + 1 < nq ? q + 1 : q;
            Pn = tl_load_panel<PK>(wrow_of(qn), (qn % npan) * PK);
        }
        if (rem == 0) {
#pragma unroll
            for (int st = 0; st < NS; ++st) kacc[st] = 0.0f;
        }
        if (pan == 0) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bo + (4 * hb + lk) * C + 4 * cq);
#pragma unroll
            for (int st = 0; st < NS; ++st) acc[st] = bv;
        }
        tl_mma_panel<NS, PK>(P, in, pan * PK, li, lk, acc);
        if (pan == npan - 1) {
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                const f32x4 dx = *reinterpret_cast<const f32x4*>(DX + (cq * NSP + st * 16 + li) * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) kacc[st] = fmaf(tanh_dev(acc[st][r]), dx[r], kacc[st]);
            }
            if (cq == ncq - 1) {
#pragma unroll
                for (int st = 0; st < NS; ++st) KO[(hb * NSP + st * 16 + li) * 4 + lk] = kacc[st];
            }
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int NS>
__global__ __launch_bounds__(TL_THREADS) void ncde_fwd_tiled(KArgs a) {
    constexpr int NSP = NS * 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b0 = blockIdx.x * NSP;
    const int H = a.H;
    int D = H;
    for (int l = 0; l < a.n_layers; ++l) D = max(D, a.dout[l]);
    const int HS = H * NSP, DS = D * NSP;
    float* YS = lds;            // stage input
    float* ACT0 = YS + HS;
    float* ACT1 = ACT0 + DS;
    float* KO = ACT1 + DS;      // f(z).dX of the stage
    float* DX = KO + HS;        // [C/4][NSP][4]

    // state slice of this thread: element e = tid + q * TL_THREADS of the [H/4][NSP][4] arrays
    float y0[TL_EMAX], k1[TL_EMAX], k2[TL_EMAX];
#pragma unroll
    for (int q = 0; q < TL_EMAX; ++q) {
        const int e = tid + q * TL_THREADS;
        y0[q] = k1[q] = k2[q] = 0.0f;
        if (e < HS) {
            const int u = ((e >> 2) / NSP) * 4 + (e & 3), s = (e >> 2) % NSP, b = b0 + s;
            const float v = b < a.B ? a.z0[(long long)b * H + u] : 0.0f;
            y0[q] = v;
            YS[e] = v;
            if (b < a.B) a.out[((long long)b * a.n_out) * H + u] = v;
        }
    }
    const int S = n_stages(a.method);
    const int dlast = a.dout[a.n_layers - 1];
    const int nkb_o = dlast >> 4;
    int cur_idx = -1;
    for (int n = 0; n < a.T - 1; ++n) {
        for (int j = 0; j < S; ++j) {
            const float t = (float)n + stage_offset(a.method, j);
            const int idx = piece_index(t, a.n_pieces);
            if (a.interp != NCDE_INTERP_LINEAR || idx != cur_idx) {
                tl_load_dx<NS>(a, b0, idx, t - (float)idx, DX, tid);
                cur_idx = idx;
            }
            __syncthreads();
            if (a.stages) {  // record the stage input for the exact discrete backward
                float* rec = a.stages + ((long long)(n * S + j) * a.B + b0) * H;
                for (int e = tid; e < NSP * H; e += TL_THREADS) {
                    const int s = e / H, u = e - s * H;
                    if (b0 + s < a.B) rec[e] = YS[((u >> 2) * NSP + s) * 4 + (u & 3)];
                }
            }
            const float* in = YS;
            for (int l = 0; l < a.n_layers; ++l) {
                float* outb = (l & 1) ? ACT1 : ACT0;
                tl_dense_relu<NS>(a.W[l], a.b[l], a.dout[l], a.din[l], in, outb, wave, lane);
                __syncthreads();
                in = outb;
            }
            switch (tl_panel_k(nkb_o)) {
                case 8: tl_output_pk<NS, 8>(a, in, DX, KO, dlast, wave, lane); break;
                case 4: tl_output_pk<NS, 4>(a, in, DX, KO, dlast, wave, lane); break;
                case 2: tl_output_pk<NS, 2>(a, in, DX, KO, dlast, wave, lane); break;
                default: tl_output_pk<NS, 1>(a, in, DX, KO, dlast, wave, lane); break;
            }
            __syncthreads();
            // Butcher bookkeeping (same operation order as ncde_generic.hip's StageCombine)
#pragma unroll
            for (int q = 0; q < TL_EMAX; ++q) {
                const int e = tid + q * TL_THREADS;
                if (e < HS) {
                    const float k = KO[e];
                    float ys;
                    bool last = false;
                    if (a.method == NCDE_RK4_38) {
                        if (j == 0) { k1[q] = k; ys = y0[q] + k * 0.333333343267440796f; }
                        else if (j == 1) { k2[q] = k; ys = y0[q] + (k - k1[q] * 0.333333343267440796f); }
                        else if (j == 2) { ys = y0[q] + ((k1[q] - k2[q]) + k); k2[q] = k2[q] + k; }
                        else { y0[q] = y0[q] + ((k1[q] + 3.0f * k2[q]) + k) * 0.125f; ys = y0[q]; last = true; }
                    } else if (a.method == NCDE_MIDPOINT) {
                        if (j == 0) { ys = y0[q] + k * 0.5f; }
                        else { y0[q] = y0[q] + k; ys = y0[q]; last = true; }
                    } else {
                        y0[q] = y0[q] + k; ys = y0[q]; last = true;
                    }
                    YS[e] = ys;
                    if (last) {
                        const int u = ((e >> 2) / NSP) * 4 + (e & 3), s = (e >> 2) % NSP, b = b0 + s;
                        if (b < a.B) {
                            if (a.output == NCDE_OUT_KNOTS) a.out[((long long)b * a.n_out + (n + 1)) * H + u] = ys;
                            else if (n == a.T - 2) a.out[((long long)b * a.n_out + 1) * H + u] = ys;
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
namespace {

int tiled_dmax(const NcdeProblem* p) {
    int D = p->hidden;
    for (int l = 0; l < p->n_layers; ++l) D = std::max(D, p->layer_out[l]);
    return D;
}

size_t tiled_fwd_lds(const NcdeProblem* p, int ns) {
    return sizeof(float) * (size_t)(ns * 16) * (size_t)(2 * p->hidden + 2 * tiled_dmax(p) + p->channels);
}

// forward sample tiles per workgroup.  More tiles = more reuse of each weight fragment, but measured on MI355X
// (cfg5, B = 4096: NS1 515 ms, NS2 777 ms, NS4 1273 ms) a workgroup on every CU beats reuse: take the largest NS
// that still leaves >= 256 workgroups (one per CU).  Development flags 0x1000/0x2000/0x4000 force NS = 1/2/4.
int tiled_fwd_ns(const NcdeProblem* p) {
    auto fits = [&](int ns) { return tiled_fwd_lds(p, ns) <= (size_t)kLdsLimit && p->hidden * ns * 16 <= TL_EMAX * TL_THREADS; };
    if (p->flags & 0x1000u) return fits(1) ? 1 : 0;
    if (p->flags & 0x2000u) return fits(2) ? 2 : 0;
    if (p->flags & 0x4000u) return fits(4) ? 4 : 0;
    for (int ns = 4; ns >= 2; ns >>= 1)
        if (fits(ns) && (p->batch + 16 * ns - 1) / (16 * ns) >= 256) return ns;
    return fits(1) ? 1 : 0;
}

}  // namespace

bool ncde_tiled_supported(const NcdeProblem* p, int pass) {
    if (p->n_layers < 1 || p->hidden % 16 || p->channels % 4) return false;
    auto aligned = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    for (int l = 0; l < p->n_layers; ++l)
        if (p->layer_out[l] % 16 || p->layer_in[l] % 16 || !aligned(p->layer_W[l]) || !aligned(p->layer_b[l])) return false;
    if (!aligned(p->Wo) || !aligned(p->bo)) return false;
    if (pass != 0) return false;
    return tiled_fwd_ns(p) > 0;
}

const char* ncde_tiled_kernel_name(const NcdeProblem* p, int pass) {
    if (!ncde_tiled_supported(p, pass)) return nullptr;
    const int ns = tiled_fwd_ns(p);
    return ns == 4 ? "ncde_fwd_tiled<NS4>" : (ns == 2 ? "ncde_fwd_tiled<NS2>" : "ncde_fwd_tiled<NS1>");
}

int64_t ncde_tiled_workspace_bytes(const NcdeProblem* p, int pass) {
    if (!ncde_tiled_supported(p, pass)) return NCDE_ERR_UNSUPPORTED;
    return 256;
}

int ncde_tiled_forward(const NcdeProblem* p, float* out, float* stages, void* ws, size_t ws_bytes, hipStream_t st) {
    (void)ws; (void)ws_bytes;
    if (!ncde_tiled_supported(p, 0)) return NCDE_ERR_UNSUPPORTED;
    const Layout y = make_layout(p);
    KArgs a;
    fill_kargs(p, y, &a);
    a.out = out;
    a.stages = stages;
    const int ns = tiled_fwd_ns(p);
    const size_t lds = tiled_fwd_lds(p, ns);
    void (*fn)(KArgs) = ns == 4 ? ncde_fwd_tiled<4> : (ns == 2 ? ncde_fwd_tiled<2> : ncde_fwd_tiled<1>);
    if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return NCDE_ERR_HIP;
    const int nwg = (p->batch + ns * 16 - 1) / (ns * 16);
    hipLaunchKernelGGL(fn, dim3(nwg), dim3(TL_THREADS), lds, st, a);
    return hipGetLastError() == hipSuccess ? NCDE_OK : NCDE_ERR_HIP;
}
