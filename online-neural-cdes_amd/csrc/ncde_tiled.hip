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
#include <type_traits>
#include "ncde_common.h"
#include "ncde_bf3.h"
#include "ncde_host.h"
#include "ncde_tiled.h"
#include "ncde_coop.h"
#include "ncde_dwo2.h"

#define TL_NW 8   // waves per workgroup of the forward family (the backward sweep runs 4, see ncde_adj_tiled)
#define TL_THREADS (64 * TL_NW)
#define TL_EMAX 16  // forward: state elements per thread, H * NS * 16 <= TL_EMAX * TL_THREADS (instantiated: 4, 16)

namespace {

// dX/dt(t) of the tile's samples -> DX[(c>>2)][s][c&3]
template <int NS, int NT>
__device__ __forceinline__ void tl_load_dx(const KArgs& a, int b0, int idx, float frac, float kdt, float* DX, int tid) {
    constexpr int NSP = NS * 16;
    const int C = a.C, Cc = a.Cc;      // Cc: channels of the coefficient tensor (< C when the problem was zero-padded)
    for (int e = tid; e < NSP * C; e += NT) {
        const int s = e / C, c = e - s * C;
        const int b = b0 + s;
        float v = 0.0f;
        if (b < a.B && c < Cc) {
            const float* p = a.coeffs + (long long)b * a.cs_b + (long long)idx * a.cs_t;
            if (a.interp == NCDE_INTERP_LINEAR) {
                v = p[a.cs_t + c] - p[c];
                if (kdt != 1.0f) v = v / kdt;      // user knot grid (interpolation_linear.py:231-234); 1 on the default grid
            } else {
                const float bb = p[Cc + c], cc = p[2 * Cc + c], dd = p[3 * Cc + c];
                const float inner = cc + dd * frac;
                v = bb + inner * frac;
            }
        }
        DX[((c >> 2) * NSP + s) * 4 + (c & 3)] = v;
    }
}

__device__ __forceinline__ float tl_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
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
// The same fragments from the PACKED copy of the output layer (ncde_pack_panels below): tile-major, then k-block, then
// lane, so one load instruction of a wave reads 1 KB contiguous.  Read in place, a fragment load touches 64 different
// 16-byte pieces (lane -> weight row), and the texture path spends a cycle per piece: measured at cfg5, the forward
// sat at 340k cycles per stage and SIMD whether or not the MFMAs were issued and whether or not the loads hit L1.
template <int PK>
__device__ __forceinline__ Panel<PK> tl_load_panel_packed(const float* tile, int lane) {
    Panel<PK> P;
#pragma unroll
    for (int i = 0; i < PK; ++i) P.v[i] = *reinterpret_cast<const f32x4*>(tile + (i * 64 + lane) * 4);
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
// xb != NULL: additionally leave the result as three bf16 pieces in the B-operand order of v_mfma_f32_16x16x32_bf16
// (NS = 1 only): word (((c * 3 + piece) * 4 + kg) * 16 + sample) * 4 + dw holds units 32 c + 8 kg + 2 dw, + 1.
// ACT: 0 relu (hidden layers), 1 tanh, 2 sigmoid (the heads of the evaluate / derivative input modes, which are plain
// H-row dense layers)
// XH = 1: the split image of the result is the 2-way split-fp16 one (two pieces per chunk instead of three: word
// (((c * 2 + piece) * 4 + kg) * 16 + sample) * 4 + dw) and *mxp tracks the largest magnitude split (range fault, ncde_bf3.h)
template <int NS, int PK, int NWV, int ACT = 0, int XH = 0>
// ld = row stride of W in floats (0: K; a resident LDS copy is stored with K + 4 so that the 16 rows of a fragment load spread
// over the banks)
__device__ __forceinline__ void tl_dense_relu_pk(const float* __restrict__ W, const float* __restrict__ bias, int N, int K,
                                                 const float* in, float* out, int wave, int lane, unsigned* xb = nullptr, int ld = 0,
                                                 float* mxp = nullptr) {
    constexpr int NSP = NS * 16;
    const int li = lane & 15, lk = lane >> 4;
    const int npan = (K >> 4) / PK;
    const int ntile = ((N >> 4) - wave + NWV - 1) / NWV;       // row tiles wave, wave+8, ...
    if (ntile <= 0) return;
    const int nq = ntile * npan;
    const int ldw = ld ? ld : K;
    auto wrow_of = [&](int q) { return W + (long long)(16 * (wave + NWV * (q / npan)) + li) * ldw + 4 * lk; };
    Panel<PK> Pn = tl_load_panel<PK>(wrow_of(0), 0);
    f32x4 acc[NS];
    for (int q = 0; q < nq; ++q) {
        const int ti = q / npan, pan = q - ti * npan, t = wave + NWV * ti;
        const Panel<PK> P = Pn;
        if (q + 1 < nq) Pn = tl_load_panel<PK>(wrow_of(q + 1), ((q + 1) % npan) * PK);      // (nothing behind the last pair)
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
                for (int r = 0; r < 4; ++r) o[r] = ACT == 0 ? relu_dev(acc[st][r]) : (ACT == 1 ? tanh_dev(acc[st][r]) : tl_sigmoid(acc[st][r]));
                *reinterpret_cast<f32x4*>(out + ((4 * t + lk) * NSP + st * 16 + li) * 4) = o;
                if constexpr (NS == 1 && ACT == 0) {
                    if (xb) {      // units 16 t + 4 lk + r: chunk t >> 1, k-group 2 (t & 1) + (lk >> 1), dwords 2 (lk & 1), + 1
                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                        if constexpr (XH != 0) {
                            unsigned h0, l0, h1, l1;
                            split_pair_h(o[0], o[1], h0, l0, *mxp);
                            split_pair_h(o[2], o[3], h1, l1, *mxp);
                            unsigned* dst = xb + ((((t >> 1) * 2) * 4 + 2 * (t & 1) + (lk >> 1)) * 16 + li) * 4 + 2 * (lk & 1);
                            *reinterpret_cast<u32x2*>(dst) = (u32x2){h0, h1};
                            *reinterpret_cast<u32x2*>(dst + 256) = (u32x2){l0, l1};
                        } else {
                            unsigned h0, m0, l0, h1, m1, l1;
                            split_pair(o[0], o[1], h0, m0, l0);
                            split_pair(o[2], o[3], h1, m1, l1);
                            unsigned* dst = xb + ((((t >> 1) * 3) * 4 + 2 * (t & 1) + (lk >> 1)) * 16 + li) * 4 + 2 * (lk & 1);
                            *reinterpret_cast<u32x2*>(dst) = (u32x2){h0, h1};
                            *reinterpret_cast<u32x2*>(dst + 256) = (u32x2){m0, m1};
                            *reinterpret_cast<u32x2*>(dst + 512) = (u32x2){l0, l1};
                        }
                    }
                }
            }
        }
    }
}
// MAXPK: largest panel (k-blocks per 16-byte-per-lane load group) to use -- a panel is 4 PK registers, two are in flight
template <int NS, int NWV, int ACT = 0, int XH = 0, int MAXPK = 8>
__device__ __forceinline__ void tl_dense_relu(const float* __restrict__ W, const float* __restrict__ bias, int N, int K,
                                              const float* in, float* out, int wave, int lane, unsigned* xb = nullptr, int ld = 0,
                                              float* mxp = nullptr) {
    int pk_ = tl_panel_k(K >> 4);
    if (pk_ > MAXPK) pk_ = MAXPK;      // (powers of two: a smaller panel still divides K / 16)
    switch (pk_) {
        case 8: tl_dense_relu_pk<NS, 8, NWV, ACT, XH>(W, bias, N, K, in, out, wave, lane, xb, ld, mxp); break;
        case 4: tl_dense_relu_pk<NS, 4, NWV, ACT, XH>(W, bias, N, K, in, out, wave, lane, xb, ld, mxp); break;
        case 2: tl_dense_relu_pk<NS, 2, NWV, ACT, XH>(W, bias, N, K, in, out, wave, lane, xb, ld, mxp); break;
        default: tl_dense_relu_pk<NS, 1, NWV, ACT, XH>(W, bias, N, K, in, out, wave, lane, xb, ld, mxp); break;
    }
}

// Direct modes, small models: the matrices the host marked (KArgs.tres*) are copied once into LDS -- [N][K] row-major with
// row stride K + 4 and the bias behind -- so the dense phases of a stage (each otherwise starting with an exposed L2 round trip
// for its first fragment) read them from there.  Returns nothing; TlW hands the right pointers to the layer calls.
struct TlW {
    const float* W;
    const float* b;
    int ld;
};
__device__ __forceinline__ TlW tl_wref(const float* lds, int off, const float* W, const float* b, int N, int K) {
    TlW r;
    if (off) { r.W = lds + off; r.b = lds + off + N * (K + 4); r.ld = K + 4; }
    else { r.W = W; r.b = b; r.ld = 0; }
    return r;
}
template <int NT>
__device__ __forceinline__ void tl_fill_resident(const KArgs& a, float* lds, int tid) {
    auto fill = [&](const float* W, const float* b, int N, int K, int off) {
        for (int e = tid; e < N * K; e += NT) {
            const int r = e / K, c = e - r * K;
            lds[off + r * (K + 4) + c] = W[e];
        }
        for (int r = tid; r < N; r += NT) lds[off + N * (K + 4) + r] = b[r];
    };
    for (int l = 0; l < a.n_layers; ++l) {
        bool first = a.tres[l] != 0;
        for (int q = 0; q < l; ++q) first = first && a.tres[q] != a.tres[l];
        if (first) fill(a.W[l], a.b[l], a.dout[l], a.din[l], a.tres[l]);
    }
    const int dlast = a.dout[a.n_layers - 1];
    if (a.tres_o) fill(a.Wo, a.bo, a.H, dlast, a.tres_o);
    if (a.tres_g) fill(a.Wg, a.bg, a.H, dlast, a.tres_g);
}

// control input of the evaluate / derivative modes for the tile's samples -> rows row0 .. row0 + C - 1 of the activation array
// `U` ([unit/4][sample][unit%4]): value = false: dX/dt(t); value = true: X(t)  (ncde_variant.hip's vr_load_cin; linear
// interpolation_linear.py:221-234, cubic interpolation_cubic.py:324-336)
template <int NS, int NT>
__device__ __forceinline__ void tl_load_cin(const KArgs& a, int b0, const StageDesc& sd, bool value, float* U, int row0, int tid) {
    constexpr int NSP = NS * 16;
    const int C = a.C, Cc = a.Cc;
    for (int e = tid; e < NSP * C; e += NT) {
        const int s = e / C, c = e - s * C;
        const int b = b0 + s;
        float v = 0.0f;
        if (b < a.B && c < Cc) {
            const float* p = a.coeffs + (long long)b * a.cs_b + (long long)sd.idx * a.cs_t;
            if (a.interp == NCDE_INTERP_LINEAR) {
                const float d = p[a.cs_t + c] - p[c];
                v = value ? p[c] + (sd.frac * d) / sd.kdt : (sd.kdt != 1.0f ? d / sd.kdt : d);
            } else {
                const float aa = p[c], bb = p[Cc + c], cc = p[2 * Cc + c], dd = p[3 * Cc + c];
                if (value) {
                    float inner = 0.5f * cc + (dd * sd.frac) / 3.0f;
                    inner = bb + inner * sd.frac;
                    v = aa + inner * sd.frac;
                } else {
                    const float inner = cc + dd * sd.frac;
                    v = bb + inner * sd.frac;
                }
            }
        }
        const int u = row0 + c;
        U[((u >> 2) * NSP + s) * 4 + (u & 3)] = v;
    }
}

// output layer + tanh + channel contraction for the h-blocks of this wave -> KO.  Tile rows (g, r) <-> (h = 4hb+g,
// c = 4cq+r); sequence of (h-block, channel quad, panel) triples, next panel in flight while one computes.
template <int NS, int PK, int NWV>
__device__ __forceinline__ void tl_output_pk(const KArgs& a, const float* in, const float* DX, float* KO, int dlast, int wave, int lane) {
    constexpr int NSP = NS * 16;
    const int li = lane & 15, lk = lane >> 4;
    const int C = a.C, nhb = a.H >> 2, ncq = C >> 2;
    const int npan = (dlast >> 4) / PK;
    const int nhb_w = (nhb - wave + NWV - 1) / NWV;
    if (nhb_w <= 0) return;
    const int per_hb = ncq * npan, nq = nhb_w * per_hb;
    auto wrow_of = [&](int q) {
        const int hb = wave + NWV * (q / per_hb), cq = (q % per_hb) / npan;
        return a.Wo + (long long)((4 * hb + (li >> 2)) * C + 4 * cq + (li & 3)) * dlast + 4 * lk;
    };
    Panel<PK> Pn = tl_load_panel<PK>(wrow_of(0), 0);
    float kacc[NS];
    f32x4 acc[NS];
    for (int q = 0; q < nq; ++q) {
        const int hi = q / per_hb, rem = q - hi * per_hb, cq = rem / npan, pan = rem - cq * npan;
        const int hb = wave + NWV * hi;
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

// Same, for the common case K = 16 PK (one panel = the whole tile row block): a straight-line loop body (no branches:
// the waitcnt pass then keeps the next tile's loads in flight across the MFMAs), two named buffers, unrolled by two.
// one head tile as fetched: the 16-row weight panel(s) and the bias quad of this lane's D rows
template <int PK, int GATED>
struct TlTile {
    Panel<PK> P;
    f32x4 bias;
};
template <int PK>
struct TlTile<PK, 1> {
    Panel<PK> P;
    f32x4 bias;
    Panel<PK> G;       // sigmoid head
    f32x4 biasg;
};

// GATED: the minimal-gated field (gating.py:7-32) -- a second head Wg, M = sigmoid(Wg x + bg) * tanh(Wo x + bo)
template <int NS, int PK, int NWV, int GATED>
__device__ __forceinline__ void tl_output_whole(const KArgs& a, const float* in, const float* DX, float* KO, int wave, int lane) {
    constexpr int NSP = NS * 16;
    const int li = lane & 15, lk = lane >> 4;
    const int C = a.C, nhb = a.H >> 2, ncq = C >> 2;
    const int nhb_w = (nhb - wave + NWV - 1) / NWV;
    if (nhb_w <= 0) return;
    const int nq = nhb_w * ncq;
    using TileIn = TlTile<PK, GATED>;
    int fhi = 0, fcq = 0, fq = 0;
    auto fetch = [&]() {
        const int hb = wave + NWV * fhi;
        TileIn t;
        const long long woff = (long long)(hb * ncq + fcq) * (PK * 256);
        t.P = tl_load_panel_packed<PK>(a.Wo_pk + woff, lane);
        t.bias = *reinterpret_cast<const f32x4*>(a.bo + (4 * hb + lk) * C + 4 * fcq);
        if constexpr (GATED != 0) {
            t.G = tl_load_panel_packed<PK>(a.Wg_pk + woff, lane);
            t.biasg = *reinterpret_cast<const f32x4*>(a.bg + (4 * hb + lk) * C + 4 * fcq);
        }
        const bool more = fq + 1 < nq, wrap = fcq + 1 == ncq;
        fq += more ? 1 : 0;
        fhi += (more && wrap) ? 1 : 0;
        fcq = more ? (wrap ? 0 : fcq + 1) : fcq;
        return t;
    };
    int chi = 0, ccq = 0;
    float kacc[NS];
#pragma unroll
    for (int st = 0; st < NS; ++st) kacc[st] = 0.0f;
    auto step = [&](const TileIn& t) {
        const int hb = wave + NWV * chi;
        f32x4 acc[NS], accg[NS];
#pragma unroll
        for (int st = 0; st < NS; ++st) acc[st] = t.bias;
        tl_mma_panel<NS, PK>(t.P, in, 0, li, lk, acc);
        if constexpr (GATED != 0) {
#pragma unroll
            for (int st = 0; st < NS; ++st) accg[st] = t.biasg;
            tl_mma_panel<NS, PK>(t.G, in, 0, li, lk, accg);
        }
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            const f32x4 dx = *reinterpret_cast<const f32x4*>(DX + (ccq * NSP + st * 16 + li) * 4);
            float kk = ccq == 0 ? 0.0f : kacc[st];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float m = tanh_dev(acc[st][r]);
                if constexpr (GATED != 0) m = tl_sigmoid(accg[st][r]) * m;
                kk = fmaf(m, dx[r], kk);
            }
            kacc[st] = kk;
            KO[(hb * NSP + st * 16 + li) * 4 + lk] = kk;     // running sum; the last channel quad leaves the total
        }
        const bool wrap = ccq + 1 == ncq;
        chi += wrap ? 1 : 0;
        ccq = wrap ? 0 : ccq + 1;
    };
    TileIn TA = fetch(), TB;
    for (int i = 0; i < (nq >> 1); ++i) {
        // the scheduling fences keep each tile's loads a whole tile ahead of their use: left alone, the machine scheduler
        // sinks them next to the MFMAs that consume them (load, wait, four MFMAs, load, ...), which exposes the L2 latency
        TB = fetch();
        __builtin_amdgcn_sched_barrier(0);
        step(TA);
        __builtin_amdgcn_sched_barrier(0);
        TA = fetch();
        __builtin_amdgcn_sched_barrier(0);
        step(TB);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (nq & 1) step(TA);
}

// ---- split-bf16 output tiles (NS = 1, last hidden width a multiple of 32) ---------------------------------------------
// The fp32 tiles above run the MFMA pipe at the fp32 vector rate: 32 MFMAs x 32 cycles per tile.  Here both operands arrive
// as three bf16 pieces -- the weights split ONCE per call by ncde_pack_panels_bf (fragment order, 1.5 x the fp32 bytes), the
// activations split by the layer that produced them (tl_dense_relu_pk, xb) -- and a tile is 6 x (K/32) bf16 MFMAs of 16
// cycles (ncde_bf3.h: fp32-equivalent, dropped terms <= 3 * 2^-24 relative).
template <int NCH, int GATED>
struct BfTile {
    u32x4 w[NCH][3];
    f32x4 bias;
    u32x4 g[GATED ? NCH : 1][3];
    f32x4 biasg;
};
template <int NCH>
__device__ __forceinline__ void tl_load_bf(const unsigned* tile, int lane, u32x4 (&w)[NCH][3]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int p = 0; p < 3; ++p) w[c][p] = *reinterpret_cast<const u32x4*>(tile + ((c * 3 + p) * 64 + lane) * 4);
}
// acc += W(tile) x, x read from its split image; two accumulator chains (even / odd chunks)
template <int NCH>
__device__ __forceinline__ f32x4 tl_mma_bf(const u32x4 (&w)[NCH][3], const unsigned* xb, int lane, f32x4 acc) {
    f32x4 acc2 = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int kPA[6] = {2, 1, 0, 1, 0, 0}, kPB[6] = {0, 0, 0, 1, 1, 2};      // the products of mfma_split: piece 0 hi, 1 mid, 2 lo
#pragma unroll
    for (int c = 0; c < NCH; c += 2) {      // two chunks at a time: consecutive MFMAs alternate between the two accumulator chains
        u32x4 B0[3], B1[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
            B0[pc] = *reinterpret_cast<const u32x4*>(xb + ((c * 3 + pc) * 64 + lane) * 4);
            if constexpr (NCH > 1) B1[pc] = *reinterpret_cast<const u32x4*>(xb + (((c + 1) * 3 + pc) * 64 + lane) * 4);
        }
#pragma unroll
        for (int pp = 0; pp < 6; ++pp) {
            acc = mfma_bf(w[c][kPA[pp]], B0[kPB[pp]], acc);
            if constexpr (NCH > 1) acc2 = mfma_bf(w[c + 1][kPA[pp]], B1[kPB[pp]], acc2);
        }
    }
    if constexpr (NCH > 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += acc2[r];
    }
    return acc;
}

template <int NCH, int NWV, int GATED>
__device__ __forceinline__ void tl_output_bf(const KArgs& a, const unsigned* xb, const float* DX, float* KO, int wave, int lane) {
    constexpr int NSP = 16;
    const int li = lane & 15, lk = lane >> 4;
    const int C = a.C, nhb = a.H >> 2, ncq = C >> 2;
    const int nhb_w = (nhb - wave + NWV - 1) / NWV;
    if (nhb_w <= 0) return;
    const int nq = nhb_w * ncq;
    using TileIn = BfTile<NCH, GATED>;
    const unsigned* wo = reinterpret_cast<const unsigned*>(a.Wo_pk);
    const unsigned* wg = reinterpret_cast<const unsigned*>(a.Wg_pk);
    int fhi = 0, fcq = 0, fq = 0;
    auto fetch = [&]() {
        const int hb = wave + NWV * fhi;
        TileIn t;
        const long long woff = (long long)(hb * ncq + fcq) * (NCH * 3 * 256);
        tl_load_bf<NCH>(wo + woff, lane, t.w);
        t.bias = *reinterpret_cast<const f32x4*>(a.bo + (4 * hb + lk) * C + 4 * fcq);
        if constexpr (GATED != 0) {
            tl_load_bf<NCH>(wg + woff, lane, t.g);
            t.biasg = *reinterpret_cast<const f32x4*>(a.bg + (4 * hb + lk) * C + 4 * fcq);
        }
        const bool more = fq + 1 < nq, wrap = fcq + 1 == ncq;
        fq += more ? 1 : 0;
        fhi += (more && wrap) ? 1 : 0;
        fcq = more ? (wrap ? 0 : fcq + 1) : fcq;
        return t;
    };
    int chi = 0, ccq = 0;
    float kacc = 0.0f;
    auto step = [&](const TileIn& t) {
        const int hb = wave + NWV * chi;
        const f32x4 acc = tl_mma_bf<NCH>(t.w, xb, lane, t.bias);
        f32x4 accg;
        if constexpr (GATED != 0) accg = tl_mma_bf<NCH>(t.g, xb, lane, t.biasg);
        const f32x4 dx = *reinterpret_cast<const f32x4*>(DX + (ccq * NSP + li) * 4);
        float kk = ccq == 0 ? 0.0f : kacc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float m = tanh_dev(acc[r]);
            if constexpr (GATED != 0) m = tl_sigmoid(accg[r]) * m;
            kk = fmaf(m, dx[r], kk);
        }
        kacc = kk;
        KO[(hb * NSP + li) * 4 + lk] = kk;     // running sum; the last channel quad leaves the total
        const bool wrap = ccq + 1 == ncq;
        chi += wrap ? 1 : 0;
        ccq = wrap ? 0 : ccq + 1;
    };
    TileIn TA = fetch(), TB;
    for (int i = 0; i < (nq >> 1); ++i) {
        TB = fetch();          // fences: see tl_output_whole
        __builtin_amdgcn_sched_barrier(0);
        step(TA);
        __builtin_amdgcn_sched_barrier(0);
        TA = fetch();
        __builtin_amdgcn_sched_barrier(0);
        step(TB);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (nq & 1) step(TA);
}


// ---- split-fp16 output tiles (round 4; NS = 1, last hidden width a multiple of 32) ------------------------------------------------
// The forward-side operands -- weights and x_L, O(1) magnitudes -- as TWO fp16 pieces (ncde_bf3.h): 3 f16 MFMAs per K-chunk of 32
// instead of 6 bf16 ones, and a weight stream of 1.0 x the fp32 bytes instead of 1.5 x -- the stream is what bounds the output
// phase at cfg5.  The fp16 range is speculated on exactly as in the register-resident family: the
// workgroup reports a range fault for its sample tile and the split-bf16 instantiation, launched behind it, re-executes that tile.
template <int NCH, int GATED>
struct H2Tile {
    u32x4 w[NCH][2];
    f32x4 bias;
    u32x4 g[GATED ? NCH : 1][2];
    f32x4 biasg;
};
template <int NCH>
__device__ __forceinline__ void tl_load_h2(const unsigned* tile, int lane, u32x4 (&w)[NCH][2]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int p = 0; p < 2; ++p) w[c][p] = *reinterpret_cast<const u32x4*>(tile + ((c * 2 + p) * 64 + lane) * 4);
}
// bias + W(tile) x: main accumulator (hi hi) and cross accumulator (hi lo + lo hi, carries 2^11), one pair per chunk parity
template <int NCH>
__device__ __forceinline__ f32x4 tl_mma_h2(const u32x4 (&w)[NCH][2], const unsigned* xb, int lane, f32x4 bias) {
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 m0 = bias, x0 = z4, m1 = z4, x1 = z4;
#pragma unroll
    for (int c = 0; c < NCH; c += 2) {
        u32x4 B0[2], B1[2];
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            B0[pc] = *reinterpret_cast<const u32x4*>(xb + ((c * 2 + pc) * 64 + lane) * 4);
            if constexpr (NCH > 1) B1[pc] = *reinterpret_cast<const u32x4*>(xb + (((c + 1) * 2 + pc) * 64 + lane) * 4);
        }
        m0 = mfma_h(w[c][0], B0[0], m0);
        if constexpr (NCH > 1) m1 = mfma_h(w[c + 1][0], B1[0], m1);
        x0 = mfma_h(w[c][1], B0[0], x0);
        if constexpr (NCH > 1) x1 = mfma_h(w[c + 1][1], B1[0], x1);
        x0 = mfma_h(w[c][0], B0[1], x0);
        if constexpr (NCH > 1) x1 = mfma_h(w[c + 1][0], B1[1], x1);
    }
    if constexpr (NCH > 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { m0[r] += m1[r]; x0[r] += x1[r]; }
    }
    return h2_combine(m0, x0);
}

template <int NCH, int NWV, int GATED>
__device__ __forceinline__ void tl_output_h2(const KArgs& a, const unsigned* xb, const float* DX, float* KO, int wave, int lane) {
    constexpr int NSP = 16;
    const int li = lane & 15, lk = lane >> 4;
    const int C = a.C, nhb = a.H >> 2, ncq = C >> 2;
    const int nhb_w = (nhb - wave + NWV - 1) / NWV;
    if (nhb_w <= 0) return;
    const int nq = nhb_w * ncq;
    using TileIn = H2Tile<NCH, GATED>;
    const unsigned* wo = reinterpret_cast<const unsigned*>(a.Wo_pk);
    const unsigned* wg = reinterpret_cast<const unsigned*>(a.Wg_pk);
    int fhi = 0, fcq = 0, fq = 0;
    auto fetch = [&]() {
        const int hb = wave + NWV * fhi;
        TileIn t;
        const long long woff = (long long)(hb * ncq + fcq) * (NCH * 2 * 256);
        tl_load_h2<NCH>(wo + woff, lane, t.w);
        t.bias = *reinterpret_cast<const f32x4*>(a.bo + (4 * hb + lk) * C + 4 * fcq);
        if constexpr (GATED != 0) {
            tl_load_h2<NCH>(wg + woff, lane, t.g);
            t.biasg = *reinterpret_cast<const f32x4*>(a.bg + (4 * hb + lk) * C + 4 * fcq);
        }
        const bool more = fq + 1 < nq, wrap = fcq + 1 == ncq;
        fq += more ? 1 : 0;
        fhi += (more && wrap) ? 1 : 0;
        fcq = more ? (wrap ? 0 : fcq + 1) : fcq;
        return t;
    };
    int chi = 0, ccq = 0;
    float kacc = 0.0f;
    auto step = [&](const TileIn& t) {
        const int hb = wave + NWV * chi;
        const f32x4 acc = tl_mma_h2<NCH>(t.w, xb, lane, t.bias);
        f32x4 accg;
        if constexpr (GATED != 0) accg = tl_mma_h2<NCH>(t.g, xb, lane, t.biasg);
        const f32x4 dx = *reinterpret_cast<const f32x4*>(DX + (ccq * NSP + li) * 4);
        float kk = ccq == 0 ? 0.0f : kacc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float m = tanh_dev(acc[r]);
            if constexpr (GATED != 0) m = tl_sigmoid(accg[r]) * m;
            kk = fmaf(m, dx[r], kk);
        }
        kacc = kk;
        KO[(hb * NSP + li) * 4 + lk] = kk;     // running sum; the last channel quad leaves the total
        const bool wrap = ccq + 1 == ncq;
        chi += wrap ? 1 : 0;
        ccq = wrap ? 0 : ccq + 1;
    };
    TileIn TA = fetch(), TB;
    for (int i = 0; i < (nq >> 1); ++i) {
        TB = fetch();          // fences: see tl_output_whole
        __builtin_amdgcn_sched_barrier(0);
        step(TA);
        __builtin_amdgcn_sched_barrier(0);
        TA = fetch();
        __builtin_amdgcn_sched_barrier(0);
        step(TB);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (nq & 1) step(TA);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// output-layer weights in fragment order (once per call: the parameters change every training step)
// ------------------------------------------------------------------------------------------------
// dst[((tile * PK + i) * 64 + lane) * 4 + e] = W[row(tile, lane & 15)][16 i + 4 (lane >> 4) + e], tile = hb * (C/4) + cq,
// row(tile, li) = (4 hb + (li >> 2)) * C + 4 cq + (li & 3): exactly what lane `lane` feeds the i-th MFMA group of the tile.
__global__ __launch_bounds__(256) void ncde_pack_panels(const float* __restrict__ W, float* __restrict__ dst, int H, int C, int pk) {
    const long long n = (long long)H * C * pk * 4;          // float4 pieces
    const int ncq = C >> 2;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < n; v += (long long)gridDim.x * 256) {
        const int lane = (int)(v & 63);
        const long long ti = v >> 6;
        const int i = (int)(ti % pk);
        const int tile = (int)(ti / pk);
        const int hb = tile / ncq, cq = tile - hb * ncq, li = lane & 15, lk = lane >> 4;
        const long long row = (long long)(4 * hb + (li >> 2)) * C + 4 * cq + (li & 3);
        reinterpret_cast<f32x4*>(dst)[v] = *reinterpret_cast<const f32x4*>(W + row * (16 * pk) + 16 * i + 4 * lk);
    }
}

// dst[N][Kp] = W[N][K] with zero columns K .. Kp - 1 (layer 0 of the evaluate / derivative modes: K = H + C, any value)
__global__ __launch_bounds__(256) void ncde_pad_columns(const float* __restrict__ W, float* __restrict__ dst, int N, int K, int Kp) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < N * Kp; e += gridDim.x * 256) {
        const int r = e / Kp, c = e - r * Kp;
        dst[e] = c < K ? W[(long long)r * K + c] : 0.0f;
    }
}

// The split-bf16 copy: word (((tile * NCH + c) * 3 + piece) * 64 + lane) * 4 + dw = bf16 pieces of
// W[row(tile, lane & 15)][32 c + 8 (lane >> 4) + 2 dw, + 1] -- the A operand of v_mfma_f32_16x16x32_bf16.
__global__ __launch_bounds__(256) void ncde_pack_panels_bf(const float* __restrict__ W, unsigned* __restrict__ dst, int H, int C, int nch) {
    const long long n = (long long)H * C / 16 * nch * 64;          // (tile, chunk, lane) triples
    const int ncq = C >> 2;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < n; v += (long long)gridDim.x * 256) {
        const int lane = (int)(v & 63);
        const long long tc = v >> 6;
        const int c = (int)(tc % nch);
        const int tile = (int)(tc / nch);
        const int hb = tile / ncq, cq = tile - hb * ncq, li = lane & 15, kg = lane >> 4;
        const long long row = (long long)(4 * hb + (li >> 2)) * C + 4 * cq + (li & 3);
        const float* src = W + row * (32 * nch) + 32 * c + 8 * kg;
        float vals[8];
        *reinterpret_cast<f32x4*>(vals) = *reinterpret_cast<const f32x4*>(src);
        *reinterpret_cast<f32x4*>(vals + 4) = *reinterpret_cast<const f32x4*>(src + 4);
        const Split3 sp = split8(vals);
        u32x4* o = reinterpret_cast<u32x4*>(dst) + (tc * 3) * 64 + lane;
        o[0] = sp.hi; o[64] = sp.mid; o[128] = sp.lo;
    }
}

// The split-fp16 copy: word (((tile * NCH + c) * 2 + piece) * 64 + lane) * 4 + dw -- the A operand of v_mfma_f32_16x16x32_f16.
// A weight outside the fp16 range faults EVERY sample tile: *wfault is set and the forward kernel ORs it into its tile's word.
__global__ __launch_bounds__(256) void ncde_pack_panels_h2(const float* __restrict__ W, unsigned* __restrict__ dst, int H, int C, int nch, int* wfault) {
    const long long n = (long long)H * C / 16 * nch * 64;          // (tile, chunk, lane) triples
    const int ncq = C >> 2;
    float mx = 0.0f;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < n; v += (long long)gridDim.x * 256) {
        const int lane = (int)(v & 63);
        const long long tc = v >> 6;
        const int c = (int)(tc % nch);
        const int tile = (int)(tc / nch);
        const int hb = tile / ncq, cq = tile - hb * ncq, li = lane & 15, kg = lane >> 4;
        const long long row = (long long)(4 * hb + (li >> 2)) * C + 4 * cq + (li & 3);
        const float* src = W + row * (32 * nch) + 32 * c + 8 * kg;
        float vals[8];
        *reinterpret_cast<f32x4*>(vals) = *reinterpret_cast<const f32x4*>(src);
        *reinterpret_cast<f32x4*>(vals + 4) = *reinterpret_cast<const f32x4*>(src + 4);
        const Split2h sp = split8h(vals, mx);
        u32x4* o = reinterpret_cast<u32x4*>(dst) + (tc * 2) * 64 + lane;
        o[0] = sp.hi; o[64] = sp.lo;
    }
    if (h2_range_fault(mx)) *wfault = 1;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// RESH = H/16 > 0 (small square hidden stack, every layer H x H, NS = 1): the hidden matrices' fragments -- one row tile per
// wave, at most two distinct matrices -- are loaded once and stay in registers, so the hidden layers of a stage do not start
// with an L2 round trip each (the backward sweep's RES modes do the same).
// DIRECT = 1: the evaluate / derivative input modes (solver.py:112-137): the field input is u = [z, X(t)] or [z, dX/dt(t)]
// (H + C rows; layer 0 arrives zero-padded to a multiple of 16 columns, a.din[0]) and the heads are H-row dense layers with no
// channel contraction, dz/dt = tanh(Wo x_L + bo) (x sigmoid(Wg x_L + bg)).
#ifdef NCDE_TL_PROF
#define FW_TICK(k) { const unsigned long long now_ = __builtin_readcyclecounter(); fprof[k] += now_ - flast; flast = now_; }
#else
#define FW_TICK(k)
#endif
// COOP = 1 (round 5): the output phase is XCD-cooperative and weight-stationary like the reverse sweep's (ncde_coop.h): this workgroup
// owns one sample tile (hidden layers, Butcher bookkeeping) and KEEPS 20 row tiles of Wo as split-fp16 A fragments, which it applies to
// the x_L of every sample tile of its group -- two tiles per keeper iteration, waves 0..3 the even one, waves 4..7 the odd one, each
// wave five row tiles.  Per stage: owners publish x_L (scaled by an exact power of two per sample, split-fp16, B-operand order) ->
// group barrier -> keepers: P, tanh, f.dX slice of their state units -> exchange area -> group barrier -> owners read their tile's
// f.dX.  The weight stream of the per-workgroup kernel (5.2 MB per stage and CU from L2 at cfg5) becomes 13 KB per tile of
// activations; no range fault can occur (exact scaling), so the fault words stay 0.
template <int NS, int NWV, int EM, int GATED = 0, int BF = 0, int RESH = 0, int DIRECT = 0, int COOP = 0>
__global__ __launch_bounds__(64 * NWV) void ncde_fwd_tiled(KArgs a) {
    static_assert(DIRECT == 0 || (BF == 0 && RESH == 0), "direct heads: plain fp32 dense layers");
    static_assert(COOP == 0 || (NS == 1 && NWV == 8 && GATED == 0 && BF == 2 && RESH == 0 && DIRECT == 0), "cooperative output phase: original field, one sample tile, 8 waves");
    static_assert(BF == 0 || NS == 1, "the split-bf16 / split-fp16 output tiles are built for one sample tile per workgroup");
    static_assert(RESH == 0 || (NS == 1 && RESH <= NWV), "resident hidden fragments: one sample tile, one row tile per wave");
    // BF = 1: 3-way split-bf16 output tiles; BF = 2 (default since round 4): 2-way split-fp16 ones, range faults per sample tile
    // -> a.fault[blockIdx.x]; the BF = 1 instantiation launched behind it with only_faulted set re-executes exactly those tiles
    constexpr int NSP = NS * 16, NT = 64 * NWV;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int fault_s;
    if constexpr (COOP == 0) {      // re-execution behind a cooperative launch: only when that launch gave up (the word is final by now)
        if (a.run_if != nullptr && *a.run_if == 0u) return;
    }
    if constexpr (BF == 1) {
        if (a.only_faulted && a.fault[blockIdx.x] == 0) return;
    }
    float mx = 0.0f;      // largest magnitude split into fp16 pieces by this thread
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b0 = blockIdx.x * NSP;
    const int H = a.H;
    int D = H;
    for (int l = 0; l < a.n_layers; ++l) D = max(D, a.dout[l]);
    if (DIRECT) D = max(D, a.din[0]);
    const int HS = H * NSP, DS = D * NSP;
    const int US = DIRECT ? max(H, a.din[0]) * NSP : HS;      // the stage input carries the control rows in the direct modes
    float* YS = lds;            // stage input
    float* ACT0 = YS + US;
    float* ACT1 = ACT0 + DS;
    float* KO = ACT1 + DS;      // f(z).dX of the stage
    float* DX = KO + HS;        // [C/4][NSP][4]
    unsigned* XB = reinterpret_cast<unsigned*>(DX + a.C * NSP);   // BF: x_L as three bf16 pieces, B-operand order
    // COOP: staged inputs of the two group tiles of a keeper iteration, double-buffered by the iteration's parity (instead of XB):
    // x_L images [2][2][8 x 64 x 4], dX/dt [2][2][C x 16 <= 1280], scales [2][2][64], the waves' f.dX partials [2][8][64], the per-wave
    // maxima of |x_L| [8][16], this tile's scales [64], the barrier flag
    float* const CBX = reinterpret_cast<float*>(XB);
    float* const CDX = CBX + 2 * 2 * 2048;
    float* const CSC = CDX + 2 * 2 * 1280;
    float* const CKX = CSC + 2 * 2 * 64;
    float* const CMX = CKX + 2 * 8 * 64;
    float* const CSG = CMX + 8 * 16;
    float* const CBO = CSG + 64;      // [4 waves of a role][5 row tiles][4 lk][4]: the bias of this member's rows (see coop_fill_bias)
    // fragments FWD_KREG .. 39 of each wave's 40 wait in LDS, not in registers: with all 160 registers pinned the allocator kept one
    // fragment in scratch and reloaded it inside the MFMA chain of every keeper iteration -- and a scratch reload waits (vmcnt is in
    // order) for the staging loads of the next tiles issued just before it: one exposed L2 round trip per iteration
    constexpr int FWD_KREG = 36;
    unsigned* const CWL = reinterpret_cast<unsigned*>(CBO + 320);      // [4][40 - FWD_KREG][64 lanes][4]
    int* const CFL = reinterpret_cast<int*>(CWL + 4 * (40 - FWD_KREG) * 256);
    CoopWeights cw;
    CoopDims cd{};
    CoopSync csy{};
    __amdgpu_buffer_rsrc_t crs = coop_rsrc(a.coop_x);      // (a NULL base outside the cooperative mode: never dereferenced)
    int c_grp = 0, c_mem = 0;
    bool c_same = false;
    if constexpr (COOP != 0) {
        cd.H = H; cd.C = a.C; cd.dlast = 128; cd.M = a.coop_M; cd.G = a.coop_G;
        csy.words = a.coop_sync; csy.G = a.coop_G; csy.status = a.coop_status; csy.spin = a.coop_spin;
        c_grp = blockIdx.x % a.coop_G;
        c_mem = blockIdx.x / a.coop_G;
        coop_fill_bias(a.bo, a.C, c_mem, CBO, tid);      // (visible after the barriers below)
        const int same = coop_same_xcd(csy, c_grp, c_mem, a.coop_M, (int)gridDim.x, CFL, tid);
        if (same < 0) {      // (another workgroup never arrived: the launch is abandoned, this tile's rows of the solution are NaN)
            for (long long e = tid; e < (long long)NSP * a.n_out * a.Hr; e += NT)
                if (b0 + (int)(e / ((long long)a.n_out * a.Hr)) < a.B) a.out[(long long)b0 * a.n_out * a.Hr + e] = __builtin_nanf("");
            return;
        }
        c_same = same != 0;
        // this wave's 5 x 4 x 2 fragments of Wo stay in registers for the whole launch: the owner phases of the FORWARD (hidden layers in
        // panels of 4, twelve state registers) fit beside the 160 -- unlike the reverse sweep's, which re-reads them every stage
        coop_load_weights<0, FWD_KREG>(cw, a.coop_img, c_mem, wave & 3, lane);
        if (wave < 4) {
            const unsigned* wp = a.coop_img + (long long)c_mem * (coop_p_words() + coop_t_words()) + wave * (40 * 64 * 4);
#pragma unroll
            for (int k = FWD_KREG; k < 40; ++k)
                *reinterpret_cast<u32x4*>(CWL + ((wave * (40 - FWD_KREG) + (k - FWD_KREG)) * 64 + lane) * 4) = *reinterpret_cast<const u32x4*>(wp + (k * 64 + lane) * 4);
        }
    }
    if constexpr (DIRECT != 0) {
        for (int e = HS + tid; e < US; e += NT) YS[e] = 0.0f;      // rows H + C .. of the padded input stay zero
        tl_fill_resident<NT>(a, lds, tid);                           // small matrices -> LDS (visible after the first barrier below)
    }

    // state slice of this thread: element e = tid + q * NT of the [H/4][NSP][4] arrays
    float y0[EM], k1[EM], k2[EM];
#pragma unroll
    for (int q = 0; q < EM; ++q) {
        const int e = tid + q * NT;
        y0[q] = k1[q] = k2[q] = 0.0f;
        if (e < HS) {
            const int u = ((e >> 2) / NSP) * 4 + (e & 3), s = (e >> 2) % NSP, b = b0 + s;
            const bool live = b < a.B && u < a.Hr;      // (zero-padded problems: units >= Hr are not the caller's)
            const float v = live ? a.z0[(long long)b * a.Hr + u] : 0.0f;
            y0[q] = v;
            YS[e] = v;
            if (live) a.out[((long long)b * a.n_out) * a.Hr + u] = v;
        }
    }
    const int S = n_stages(a.method);
    const int dlast = a.dout[a.n_layers - 1];
    const int nkb_o = dlast >> 4;
    Panel<RESH ? RESH : 1> wf[2];      // RESH: row tile `wave` of the layer-0 matrix / of the other matrix
    f32x4 bfv[2];
    if constexpr (RESH != 0) {
        const int li = lane & 15, lk = lane >> 4;
        int l1 = 0;
        for (int l = 1; l < a.n_layers; ++l)
            if (a.W[l] != a.W[0]) { l1 = l; break; }
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int l = sl == 0 ? 0 : l1;
            const int tw = wave < RESH ? wave : 0;
            wf[sl] = tl_load_panel<RESH ? RESH : 1>(a.W[l] + (long long)(16 * tw + li) * (16 * RESH) + 4 * lk, 0);
            bfv[sl] = *reinterpret_cast<const f32x4*>(a.b[l] + 16 * tw + 4 * lk);
        }
    }
    int cur_idx = -1;
    // general time axis (a.plan != NULL; csrc/ncde_timeplan.hip): per-step dt, per-stage (piece, offset, knot spacing), and the
    // output rows each step emits -- same semantics as the generic / variant kernels, whose plan mode is pinned to the reference
    const bool planned = a.plan != nullptr;
    if (planned && !plan_header_ok(a, S)) return;      // (uniform: before the first barrier)
    const int n_steps = planned ? a.n_steps_fwd : a.T - 1;
    const int* pout = planned ? a.plan + plan_off_out(S, a.n_steps_fwd) : nullptr;
    int sc = 0;      // stage counter of the launch (COOP: the group barriers count in it)
#ifdef NCDE_TL_PROF
    // development: cycles per stage by phase -> the head of this tile's rows of the solution (garbage solution in such a build):
    // 0 dX/dt + record | 1 hidden layers | 2 scales + publish | 3 wait B1 | 4 keeper loop | 5 arrive + wait B2 | 6 f.dX read | 7 bookkeeping
    unsigned long long fprof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, flast = __builtin_readcyclecounter();
    unsigned long long kprof[3] = {0, 0, 0}, klast = 0;      // keeper loop: compute + f.dX stores | DMA wait | barrier
#endif
    for (int n = 0; n < n_steps; ++n) {
        const int* pstep = planned ? a.plan + plan_off_fwd() + n * plan_step_words(S) : nullptr;
        const float dt = planned ? __int_as_float(pstep[0]) : 1.0f;
        for (int j = 0; j < S; ++j, ++sc) {
            const StageDesc sd = planned ? plan_stage(pstep, j) : default_stage(a.method, (float)n + stage_offset(a.method, j), a.n_pieces);
            const int idx = sd.idx;
            bool dx_new = false;
            if constexpr (DIRECT != 0) {
                tl_load_cin<NS, NT>(a, b0, sd, a.field_input == NCDE_INPUT_EVALUATE, YS, H, tid);
            } else if (a.interp != NCDE_INTERP_LINEAR || idx != cur_idx) {
                tl_load_dx<NS, NT>(a, b0, idx, sd.frac, sd.kdt, DX, tid);
                cur_idx = idx;
                dx_new = true;
            }
            __syncthreads();
            if constexpr (COOP != 0) {      // this tile's dX/dt for the keepers (a linear control: once per piece; it stays in the exchange area)
                if (dx_new) {
                    const long long my_base = (long long)blockIdx.x * cd.per_tile_fwd();
                    for (int e = tid; e < a.C * NSP / 4; e += NT) coop_st16(crs, a.coop_x, c_same, my_base + cd.off_dx() + e * 4, *reinterpret_cast<const u32x4*>(DX + e * 4));
                }
            }
            if (a.stages) {  // record the stage input for the exact discrete backward
                const int Hr = a.Hr;
                float* rec = a.stages + ((long long)(n * S + j) * (a.Brec ? a.Brec : a.B) + b0) * Hr;      // (Brec: this launch is a chunk of the batch)
                for (int e = tid; e < NSP * Hr; e += NT) {
                    const int s = e / Hr, u = e - s * Hr;
                    if (b0 + s < a.B) rec[e] = YS[((u >> 2) * NSP + s) * 4 + (u & 3)];
                }
            }
            FW_TICK(0)
            const float* in = YS;
            for (int l = 0; l < a.n_layers; ++l) {
                float* outb = (l & 1) ? ACT1 : ACT0;
                if constexpr (RESH != 0) {
                    if (wave < RESH) {
                        const int li = lane & 15, lk = lane >> 4;
                        const bool s0 = a.W[l] == a.W[0];
                        f32x4 acc[1];
                        acc[0] = s0 ? bfv[0] : bfv[1];
                        if (s0) tl_mma_panel<1, RESH ? RESH : 1>(wf[0], in, 0, li, lk, acc);
                        else tl_mma_panel<1, RESH ? RESH : 1>(wf[1], in, 0, li, lk, acc);
                        f32x4 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = relu_dev(acc[0][r]);
                        *reinterpret_cast<f32x4*>(outb + ((4 * wave + lk) * NSP + li) * 4) = o;
                        if constexpr (BF != 0) {
                            if (l == a.n_layers - 1) {      // split image of x_L, as tl_dense_relu_pk writes it (row tile t = wave)
                                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                                if constexpr (BF == 2) {
                                    unsigned h0, l0, h1, l1;
                                    split_pair_h(o[0], o[1], h0, l0, mx);
                                    split_pair_h(o[2], o[3], h1, l1, mx);
                                    unsigned* dst = XB + ((((wave >> 1) * 2) * 4 + 2 * (wave & 1) + (lk >> 1)) * 16 + li) * 4 + 2 * (lk & 1);
                                    *reinterpret_cast<u32x2*>(dst) = (u32x2){h0, h1};
                                    *reinterpret_cast<u32x2*>(dst + 256) = (u32x2){l0, l1};
                                } else {
                                    unsigned h0, m0, l0, h1, m1, l1;
                                    split_pair(o[0], o[1], h0, m0, l0);
                                    split_pair(o[2], o[3], h1, m1, l1);
                                    unsigned* dst = XB + ((((wave >> 1) * 3) * 4 + 2 * (wave & 1) + (lk >> 1)) * 16 + li) * 4 + 2 * (lk & 1);
                                    *reinterpret_cast<u32x2*>(dst) = (u32x2){h0, h1};
                                    *reinterpret_cast<u32x2*>(dst + 256) = (u32x2){m0, m1};
                                    *reinterpret_cast<u32x2*>(dst + 512) = (u32x2){l0, l1};
                                }
                            }
                        }
                    }
                } else {
                    const TlW wr_ = tl_wref(lds, DIRECT != 0 ? a.tres[l] : 0, a.W[l], a.b[l], a.dout[l], a.din[l]);
                    if constexpr (COOP != 0)      // (the scaled split image is formed at publication; panels of 4: the registers wait for Wo)
                        tl_dense_relu<NS, NWV, 0, 0, 4>(wr_.W, wr_.b, a.dout[l], a.din[l], in, outb, wave, lane, nullptr, wr_.ld);
                    else if constexpr (BF == 2)
                        tl_dense_relu<NS, NWV, 0, 1>(wr_.W, wr_.b, a.dout[l], a.din[l], in, outb, wave, lane,
                                                     l == a.n_layers - 1 ? XB : nullptr, wr_.ld, &mx);
                    else
                        tl_dense_relu<NS, NWV>(wr_.W, wr_.b, a.dout[l], a.din[l], in, outb, wave, lane,
                                               (BF != 0 && l == a.n_layers - 1) ? XB : nullptr, wr_.ld);
                }
                __syncthreads();
                in = outb;
            }
            FW_TICK(1)
            float* KG = in == ACT0 ? ACT1 : ACT0;      // direct gated head: the free ping-pong buffer takes sigmoid(Wg x_L + bg)
            if constexpr (DIRECT != 0) {
                const TlW wo_ = tl_wref(lds, a.tres_o, a.Wo, a.bo, H, dlast);
                tl_dense_relu<NS, NWV, 1>(wo_.W, wo_.b, H, dlast, in, KO, wave, lane, nullptr, wo_.ld);
                if constexpr (GATED != 0) {
                    const TlW wg_ = tl_wref(lds, a.tres_g, a.Wg, a.bg, H, dlast);
                    tl_dense_relu<NS, NWV, 2>(wg_.W, wg_.b, H, dlast, in, KG, wave, lane, nullptr, wg_.ld);
                }
            } else
            if constexpr (COOP != 0) {
                const int li = lane & 15, lk = lane >> 4, C = a.C;
                const long long my_base = (long long)blockIdx.x * cd.per_tile_fwd();
                const unsigned bar1 = (2u * (unsigned)sc + 1u) * (unsigned)cd.M, bar2 = bar1 + (unsigned)cd.M;
                // -- OWNER: per-sample maximum of |x_L| -> power-of-two scale; publish the scaled split image ---------------------------
                {
                    float mxl = 0.0f;      // this thread's elements all belong to sample (tid >> 2) & 15
                    for (int e = tid; e < 128 * NSP; e += NT) mxl = fmaxf(mxl, fabsf(in[e]));
                    mxl = fmaxf(mxl, __shfl_xor(mxl, 1, 64)); mxl = fmaxf(mxl, __shfl_xor(mxl, 2, 64));
                    if ((lane & 3) == 0) CMX[wave * 16 + ((lane >> 2) & 15)] = mxl;
                }
                __syncthreads();
                if (tid < 16) {
                    float m_ = 0.0f;
#pragma unroll
                    for (int wv = 0; wv < 8; ++wv) m_ = fmaxf(m_, CMX[wv * 16 + tid]);
                    const float sx = coop_pow2_scale(m_), isx = coop_pow2_inv(sx) * a.coop_scale[1];
                    CSG[tid] = sx; CSG[16 + tid] = isx;
                    coop_st4(crs, a.coop_x, c_same, my_base + cd.off_sc() + tid, sx); coop_st4(crs, a.coop_x, c_same, my_base + cd.off_sc() + 16 + tid, isx);
                }
                __syncthreads();
                if (wave < 4) {
                    const int c = wave, ub = 32 * c + 8 * lk;      // K chunk c of this wave; units ub .. ub + 7 of sample li
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(in + (((ub >> 2)) * NSP + li) * 4);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(in + (((ub >> 2) + 1) * NSP + li) * 4);
                    const float sx = CSG[li];
                    unsigned h[4], l[4];
                    coop_split2(v0[0] * sx, v0[1] * sx, h[0], l[0]); coop_split2(v0[2] * sx, v0[3] * sx, h[1], l[1]);
                    coop_split2(v1[0] * sx, v1[1] * sx, h[2], l[2]); coop_split2(v1[2] * sx, v1[3] * sx, h[3], l[3]);
                    coop_st16(crs, a.coop_x, c_same, my_base + ((c * 2 + 0) * 64 + lane) * 4, (u32x4){h[0], h[1], h[2], h[3]});
                    coop_st16(crs, a.coop_x, c_same, my_base + ((c * 2 + 1) * 64 + lane) * 4, (u32x4){l[0], l[1], l[2], l[3]});
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0 && !(a.coop_inject != 0 && blockIdx.x == 1 && sc == 0)) coop_arrive(csy, c_grp);      // (fault injection: one arrival withheld)
                FW_TICK(2)
                // timeout / another workgroup gave up: the launch is abandoned and this tile's rows of the solution are NaN (never silently wrong)
                auto poison = [&]() {
                    const float qnan = __builtin_nanf("");
                    for (long long e = tid; e < (long long)NSP * a.n_out * a.Hr; e += NT)
                        if (b0 + (int)(e / ((long long)a.n_out * a.Hr)) < a.B) a.out[(long long)b0 * a.n_out * a.Hr + e] = qnan;
                };
                if (!coop_wait(csy, c_grp, bar1, CFL, tid)) { poison(); return; }
                FW_TICK(3)
                // -- KEEPER: iteration `it` takes the group's tiles 2 it (waves 0..3) and 2 it + 1 (waves 4..7); the f.dX slices of iteration
                // it - 1 are summed over the waves that hold their channel tiles and stored meanwhile; all threads fetch iteration it + 1's inputs
                const int ncq = C >> 2, jh = COOP_RPM / ncq, wph = 4 / jh;
                const int rw = wave & 3, half_w = wave >> 2;
                const int cq0 = (c_mem * COOP_RPM + rw * COOP_NRT) % ncq;
                const int n_dx4 = C * NSP / 4, n_it = cd.M / 2;
                typedef const __attribute__((address_space(1))) void* gptr_t;
                typedef __attribute__((address_space(3))) void* lptr_t;
                const int n_dxw = (n_dx4 + 63) >> 6;      // (wave-uniform choice of the extra piece: see the reverse sweep's stage_load)
                auto stage_load = [&](int it_, int buf, int tid) {      // both tiles of iteration it_ -> LDS (global_load_lds_dwordx4, sc1)
                    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), w64 = wv * 64;
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const float* src = a.coop_x + (long long)(c_grp + cd.G * (2 * it_ + hf)) * cd.per_tile_fwd();
                        __builtin_amdgcn_global_load_lds((gptr_t)(src + tid * 4), (lptr_t)(CBX + (buf * 2 + hf) * 2048 + w64 * 4), 16, 0, 16);
                        if (wv < n_dxw) __builtin_amdgcn_global_load_lds((gptr_t)(src + cd.off_dx() + tid * 4), (lptr_t)(CDX + (buf * 2 + hf) * 1280 + w64 * 4), 16, 0, 16);
                        else if (wv == 6 && tid < 400)
                            __builtin_amdgcn_global_load_lds((gptr_t)(src + cd.off_sc() + (tid - 384) * 4), (lptr_t)(CSC + (buf * 2 + hf) * 64), 16, 0, 16);
                    }
                };
                stage_load(0, 0, tid);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                const int tid_outer = tid;
#ifdef NCDE_TL_PROF
                klast = __builtin_readcyclecounter();
#endif
                for (int it = 0; it <= n_it; ++it) {
                    int tid = tid_outer;      // (offsets re-derived from an opaque copy every iteration: see the reverse sweep's keeper loops)
                    asm volatile("" : "+v"(tid));
                    const int lane = tid & 63, li = tid & 15, lk = (tid >> 4) & 3;
                    const int buf = it & 1;
                    if (it + 1 < n_it) stage_load(it + 1, buf ^ 1, tid);
                    if (it < n_it) {
                        const float* bxs = CBX + (buf * 2 + half_w) * 2048;
                        const float isx = CSC[(buf * 2 + half_w) * 64 + 16 + li];
                        const float sxw = coop_pow2_inv(isx);      // sx sw: the bias joins the scaled accumulator exactly
                        const float isx2 = isx * NCDE_TANH_PRESCALE;      // (isx is a power of two: folded into the tanh pre-scale, bit for bit)
                        float kk = 0.0f;
                        // (k is a compile-time constant after unrolling: fragments below FWD_KREG are registers, the rest LDS reads)
                        auto wfr = [&](int k) -> u32x4 {
                            if (k < FWD_KREG) return cw.f[k];
                            return *reinterpret_cast<const u32x4*>(CWL + ((rw * (40 - FWD_KREG) + (k - FWD_KREG)) * 64 + lane) * 4);
                        };
                        auto p_batch = [&](auto q0c, auto nqc) {
                            constexpr int Q0 = decltype(q0c)::value, NQ = decltype(nqc)::value;
                            f32x4 pm[NQ], px[NQ];
#pragma unroll
                            for (int i = 0; i < NQ; ++i) {
                                const f32x4 bsv = *reinterpret_cast<const f32x4*>(CBO + ((rw * COOP_NRT + Q0 + i) * 4 + lk) * 4);
#pragma unroll
                                for (int r = 0; r < 4; ++r) pm[i][r] = bsv[r] * sxw;
                                px[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                            }
                            u32x4 b0_ = *reinterpret_cast<const u32x4*>(bxs + lane * 4), b1_ = *reinterpret_cast<const u32x4*>(bxs + (64 + lane) * 4);
#pragma unroll
                            for (int c = 0; c < COOP_NCH; ++c) {
                                const u32x4 c0_ = b0_, c1_ = b1_;
                                if (c + 1 < COOP_NCH) {
                                    b0_ = *reinterpret_cast<const u32x4*>(bxs + (((c + 1) * 2 + 0) * 64 + lane) * 4);
                                    b1_ = *reinterpret_cast<const u32x4*>(bxs + (((c + 1) * 2 + 1) * 64 + lane) * 4);
                                }
#pragma unroll
                                for (int i = 0; i < NQ; ++i) pm[i] = mfma_h(wfr(((Q0 + i) * 4 + c) * 2 + 0), c0_, pm[i]);
#pragma unroll
                                for (int i = 0; i < NQ; ++i) px[i] = mfma_h(wfr(((Q0 + i) * 4 + c) * 2 + 1), c0_, px[i]);
#pragma unroll
                                for (int i = 0; i < NQ; ++i) px[i] = mfma_h(wfr(((Q0 + i) * 4 + c) * 2 + 0), c1_, px[i]);
                            }
#pragma unroll
                            for (int i = 0; i < NQ; ++i) {
                                const f32x4 pc4 = h2_combine(pm[i], px[i]);
                                const f32x4 dxv = *reinterpret_cast<const f32x4*>(CDX + (buf * 2 + half_w) * 1280 + ((cq0 + Q0 + i) * NSP + li) * 4);
#pragma unroll
                                for (int r = 0; r < 4; ++r) kk = fmaf(tanh_prescaled(pc4[r] * isx2), dxv[r], kk);
                            }
                        };
                        // neighbouring SIMDs take their two batches in opposite order (round 6: forward -5 %; keyed on the SIMD pair instead --
                        // (w, w + 4) share one -- it is -1.4 %, keyed on both it is slower than either)
                        if ((wave & 1) == 0) {
                            p_batch(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});
                            p_batch(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{});
                        } else {
                            p_batch(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{});
                            p_batch(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});
                        }
                        CKX[(buf * 8 + wave) * 64 + lane] = kk;
                    }
                    if (it >= 1 && tid < 2 * 64 * jh) {      // f.dX of this member's state units for the two tiles of iteration it - 1
                        const int hf = tid / (64 * jh), hl = (tid >> 6) % jh, ln = tid & 63, pb = buf ^ 1;
                        float ksum = 0.0f;
                        for (int wv = hl * wph; wv < (hl + 1) * wph; ++wv) ksum += CKX[(pb * 8 + 4 * hf + wv) * 64 + ln];
                        const long long tb = (long long)(c_grp + cd.G * (2 * (it - 1) + hf)) * cd.per_tile_fwd();
                        coop_st4(crs, a.coop_x, c_same, tb + cd.off_ko() + ((c_mem * jh + hl) * NSP + (ln & 15)) * 4 + (ln >> 4), ksum);
                    }
#ifdef NCDE_TL_PROF
                    { const unsigned long long n_ = __builtin_readcyclecounter(); kprof[0] += n_ - klast; klast = n_; }
#endif
                    if (it + 1 < n_it) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the next iteration's inputs: requested at the top)
#ifdef NCDE_TL_PROF
                    { const unsigned long long n_ = __builtin_readcyclecounter(); kprof[1] += n_ - klast; klast = n_; }
#endif
                    __syncthreads();
#ifdef NCDE_TL_PROF
                    { const unsigned long long n_ = __builtin_readcyclecounter(); kprof[2] += n_ - klast; klast = n_; }
#endif
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                FW_TICK(4)
                if (tid == 0) coop_arrive(csy, c_grp);
                if (!coop_wait(csy, c_grp, bar2, CFL, tid)) { poison(); return; }
                FW_TICK(5)
                for (int e4 = tid * 4; e4 < HS; e4 += NT * 4) *reinterpret_cast<f32x4*>(KO + e4) = coop_ld16f(crs, my_base + cd.off_ko() + e4);
            } else
            if constexpr (BF == 2) {
                switch (nkb_o) {
                    case 8: tl_output_h2<4, NWV, GATED>(a, XB, DX, KO, wave, lane); break;
                    case 4: tl_output_h2<2, NWV, GATED>(a, XB, DX, KO, wave, lane); break;
                    default: tl_output_h2<1, NWV, GATED>(a, XB, DX, KO, wave, lane); break;
                }
            } else if constexpr (BF != 0) {
                switch (nkb_o) {
                    case 8: tl_output_bf<4, NWV, GATED>(a, XB, DX, KO, wave, lane); break;
                    case 4: tl_output_bf<2, NWV, GATED>(a, XB, DX, KO, wave, lane); break;
                    default: tl_output_bf<1, NWV, GATED>(a, XB, DX, KO, wave, lane); break;
                }
            } else
            switch (nkb_o) {
                case 8: tl_output_whole<NS, 8, NWV, GATED>(a, in, DX, KO, wave, lane); break;
                case 4: tl_output_whole<NS, 4, NWV, GATED>(a, in, DX, KO, wave, lane); break;
                case 2: tl_output_whole<NS, 2, NWV, GATED>(a, in, DX, KO, wave, lane); break;
                case 1: tl_output_whole<NS, 1, NWV, GATED>(a, in, DX, KO, wave, lane); break;
                default:
                    if (tl_panel_k(nkb_o) == 2) tl_output_pk<NS, 2, NWV>(a, in, DX, KO, dlast, wave, lane);
                    else tl_output_pk<NS, 1, NWV>(a, in, DX, KO, dlast, wave, lane);
                    break;
            }
            __syncthreads();
            FW_TICK(6)
            // Butcher bookkeeping (same operation order as ncde_generic.hip's StageCombine)
#pragma unroll
            for (int q = 0; q < EM; ++q) {
                const int e = tid + q * NT;
                if (e < HS) {
                    const float yprev = y0[q];
                    bool last;
                    const float kst = (DIRECT != 0 && GATED != 0) ? KG[e] * KO[e] : KO[e];
                    const float ys = StageCombine::apply(a.method, j, kst, dt, y0[q], k1[q], k2[q], last);   // dt = 1: exact products
                    YS[e] = ys;
                    if (last) {
                        const int u = ((e >> 2) / NSP) * 4 + (e & 3), s = (e >> 2) % NSP, b = b0 + s;
                        if (b < a.B && u < a.Hr) {
                            const int H = a.Hr;      // row width of the caller's solution tensor
                            if (planned) {      // output pick / interpolation between the step's end points (solvers.py:103-117)
                                const int q0 = pstep[1], q1 = q0 + pstep[2];
                                for (int r = q0; r < q1; ++r) {
                                    const int kind = pout[2 * r];
                                    const float slope = __int_as_float(pout[2 * r + 1]);
                                    a.out[((long long)b * a.n_out + r) * H + u] = kind == 1 ? ys : (kind == 0 ? yprev : yprev + slope * (ys - yprev));
                                }
                            } else if (a.output == NCDE_OUT_KNOTS) a.out[((long long)b * a.n_out + (n + 1)) * H + u] = ys;
                            else if (n == a.T - 2) a.out[((long long)b * a.n_out + 1) * H + u] = ys;
                        }
                    }
                }
            }
            FW_TICK(7)
        }
    }
#ifdef NCDE_TL_PROF
    __syncthreads();
    if (lane == 0 && sc > 0)
        for (int k = 0; k < 8; ++k) a.out[(long long)b0 * a.n_out * a.Hr + wave * 8 + k] = (float)fprof[k] / (float)sc;
    if (lane == 0 && sc > 0)
        for (int k = 0; k < 3; ++k) a.out[(long long)b0 * a.n_out * a.Hr + 64 + wave * 3 + k] = (float)kprof[k] / (float)sc;
#endif
    if constexpr (COOP != 0) {      // exact scaling: no range fault can occur
        if (a.fault != nullptr && tid == 0) a.fault[blockIdx.x] = 0;
    } else
    if constexpr (BF == 2) {      // range fault of this sample tile (or of the weights: the pack kernel's word behind the tiles')
        if (a.fault != nullptr) {
            if (tid == 0) fault_s = a.fault[gridDim.x];
            __syncthreads();
            if (__builtin_amdgcn_ballot_w64(h2_range_fault(mx)) != 0 && lane == 0) fault_s = 1;
            __syncthreads();
            if (tid == 0) a.fault[blockIdx.x] = fault_s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// adjoint / exact backward, pass A: the reverse sweep (one workgroup = 16 samples)
// ------------------------------------------------------------------------------------------------
// With |Wo| = 5.2 MB per workgroup partial, accumulating dL/dWo inside the sweep costs a read-modify-write of the
// whole matrix per stage and workgroup (2.7 GB per stage at cfg5).  The sweep therefore leaves the output-layer
// parameter gradient to pass B (ncde_dwo_tiled below) and only RECORDS what that pass needs per stage: x_L (in the
// two operand layouts), the weighted cotangent and dX/dt.  Everything on the dependency chain stays here:
//   forward recompute -> P = Wo x_L, m = tanh P, f (y re-integration), dP -> dL/dx_L = Wo^T dP -> hidden layers
//   backwards (their dW/db accumulate in MFMA accumulator registers for the whole solve) -> Butcher bookkeeping.
// Wo^T fragments come from the SAME 16-byte panel loads as the forward fragments, transposed through a per-wave LDS
// scratch (row stride K + 4), so the weight stream is read once per stage.
// development: s_memtime phase counters of the sweep (-DNCDE_TL_PROF): cycles per stage by phase -> the head of the workgroup's
// hidden-layer partial (garbage gradients in such a build).  Slots: 0 forward recompute, 1 output tiles + VJP, 2 records,
// 3 partial sum, 4 hidden backward, 5 bookkeeping + publish, 7 waiting at barriers
#ifdef NCDE_TL_PROF
#define TL_TICK(k) { const unsigned long long now_ = __builtin_readcyclecounter(); tprof[k] += now_ - tlast; tlast = now_; }
#define TL_SYNC(k) { TL_TICK(k) __syncthreads(); TL_TICK(7) }
#else
#define TL_TICK(k)
#define TL_SYNC(k) __syncthreads();
#endif
#define TL_ADJ_NW 8  // waves per workgroup of the sweep.  Measured at cfg5: 8 waves (two per SIMD, 256 registers each, the
                     // cold hidden-dW accumulators spilled) 736 ms; 4 waves (one per SIMD, 494 registers, no spill) 929 ms

namespace {

template <int PK>
using WoTile = TlTile<PK, 0>;

// RES = 1: the (at most two) output tiles of this wave are resident in registers (`res`), nothing is fetched.
// BFP = 1 (original field, streamed weights): P from the split-bf16 copy of the tile (prefetched a tile ahead) and the split
// image xb of x_L; the fp32 copy of the SAME tile, needed only for the transposed products, is requested at the top of the
// step and lands under the P MFMAs and the tanh.
// PK = 16 (round 5: last hidden width 256): the transposed product goes through a HALF-size scratch (8 k-blocks at a time), and the
// wave's partial of dL/dx_L does not go to a scratch of its own (4 x 16 KB that the LDS plan does not have) but is added into `gsum` by
// the waves in turn -- wave 0 stores, 1 .. NWV-2 add, the last adds and applies relu'(x_L): same order every run.
template <int PK, int NWV, int RES, int GATED, int BFP = 0>
__device__ __forceinline__ void tl_output_vjp(const KArgs& a, const float* xL, const float* AS, const float* DX, float* KOY,
                                              float* scr, int wave, int lane, const WoTile<PK>* res, const unsigned* xb = nullptr,
                                              float* gsum = nullptr) {
    static_assert(!(RES == 1 && GATED != 0), "resident output tiles are built for the original field only");
    static_assert(BFP == 0 || (RES != 1 && GATED == 0 && PK >= 2), "split-bf16 P: original field, streamed output tiles");
    static_assert(PK <= 8 || (PK == 16 && RES == 0 && GATED == 0 && BFP == 0), "256-wide last layer: original field, streamed fp32 tiles");
    constexpr int NCH = PK >= 2 ? PK / 2 : 1;
    constexpr int PKH = PK > 8 ? 8 : PK;      // k-blocks per pass through the scratch
    constexpr int NSP = 16, SCS = 16 * PKH + 4;
    const int li = lane & 15, lk = lane >> 4;
    const int C = a.C, nhb = a.H >> 2, ncq = C >> 2;
    f32x4 accJ[PK];
#pragma unroll
    for (int jt = 0; jt < PK; ++jt) accJ[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nhb_w = (nhb - wave + NWV - 1) / NWV;
    const int nq = nhb_w > 0 ? nhb_w * ncq : 0;
    struct BfIn {
        u32x4 w[NCH][3];
        f32x4 bias;
    };
    using TileIn = std::conditional_t<BFP != 0, BfIn, TlTile<PK, GATED>>;
    int fhi = 0, fcq = 0, fq = 0;
    auto fetch = [&]() {
        const int hb = wave + NWV * fhi;
        TileIn t;
        const long long woff = (long long)(hb * ncq + fcq) * (PK * 256);
        if constexpr (BFP != 0) tl_load_bf<NCH>(a.Wo_bf + (long long)(hb * ncq + fcq) * (NCH * 3 * 256), lane, t.w);
        else t.P = tl_load_panel_packed<PK>(a.Wo_pk + woff, lane);
        t.bias = *reinterpret_cast<const f32x4*>(a.bo + (4 * hb + lk) * C + 4 * fcq);
        if constexpr (GATED != 0) {
            t.G = tl_load_panel_packed<PK>(a.Wg_pk + woff, lane);
            t.biasg = *reinterpret_cast<const f32x4*>(a.bg + (4 * hb + lk) * C + 4 * fcq);
        }
        const bool more = fq + 1 < nq, wrap = fcq + 1 == ncq;
        fq += more ? 1 : 0;
        fhi += (more && wrap) ? 1 : 0;
        fcq = more ? (wrap ? 0 : fcq + 1) : fcq;
        return t;
    };
    int chi = 0, ccq = 0;
    float kacc = 0.0f;
    // straight-line body (no branches), two named buffers: the next tile's loads stay in flight across the MFMAs
    auto step = [&](const TileIn& t) {
        const int hb = wave + NWV * chi, cq = ccq;
        f32x4 acc[1];
        if constexpr (BFP != 0) {
            const Panel<PK> Pf = tl_load_panel_packed<PK>(a.Wo_pk + (long long)(hb * ncq + cq) * (PK * 256), lane);
            acc[0] = tl_mma_bf<NCH>(t.w, xb, lane, t.bias);
#pragma unroll
            for (int i = 0; i < PK; ++i) *reinterpret_cast<f32x4*>(scr + li * SCS + 16 * i + 4 * lk) = Pf.v[i];
        } else {
#pragma unroll
            for (int i = 0; i < PKH; ++i) *reinterpret_cast<f32x4*>(scr + li * SCS + 16 * i + 4 * lk) = t.P.v[i];
            acc[0] = t.bias;
            tl_mma_panel<1, PK>(t.P, xL, 0, li, lk, acc);
        }
        const float aval = AS[(hb * NSP + li) * 4 + lk];
        const f32x4 dx = *reinterpret_cast<const f32x4*>(DX + (cq * NSP + li) * 4);
        float dP[4], dPg[4];
        float kk = cq == 0 ? 0.0f : kacc;
        f32x4 accg[1];
        if constexpr (GATED != 0) {
            accg[0] = t.biasg;
            tl_mma_panel<1, PK>(t.G, xL, 0, li, lk, accg);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float th = tanh_dev(acc[0][r]);
            const float dm = aval * dx[r];
            if constexpr (GATED != 0) {
                const float sg = tl_sigmoid(accg[0][r]);
                kk = fmaf(sg * th, dx[r], kk);
                dP[r] = (dm * sg) * (1.0f - th * th);
                dPg[r] = (dm * th) * (sg * (1.0f - sg));
            } else {
                kk = fmaf(th, dx[r], kk);
                dP[r] = dm * (1.0f - th * th);
            }
        }
        kacc = kk;
        KOY[(hb * NSP + li) * 4 + lk] = kk;      // running sum; the last channel quad leaves the total
        // dL/dx_L[j][s] += sum_u Wo[u][j] dP[u][s]: the tile's 16 rows are the K dim, k = 4 (lane>>4) + r
#pragma unroll
        for (int jt = 0; jt < PKH; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) accJ[jt] = mfma16(scr[(4 * lk + r) * SCS + 16 * jt + li], dP[r], accJ[jt]);
        if constexpr (PK > 8) {         // k-blocks 8 .. 15 through the same scratch (fp32 tiles only: t.P is the tile)
            if constexpr (BFP == 0) {
#pragma unroll
                for (int i = 0; i < PKH; ++i) *reinterpret_cast<f32x4*>(scr + li * SCS + 16 * i + 4 * lk) = t.P.v[PKH + i];
            }
#pragma unroll
            for (int jt = 0; jt < PKH; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) accJ[PKH + jt] = mfma16(scr[(4 * lk + r) * SCS + 16 * jt + li], dP[r], accJ[PKH + jt]);
        }
        if constexpr (GATED != 0) {     // ... + sum_u Wg[u][j] dPg[u][s], through the same scratch
#pragma unroll
            for (int i = 0; i < PK; ++i) *reinterpret_cast<f32x4*>(scr + li * SCS + 16 * i + 4 * lk) = t.G.v[i];
#pragma unroll
            for (int jt = 0; jt < PK; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) accJ[jt] = mfma16(scr[(4 * lk + r) * SCS + 16 * jt + li], dPg[r], accJ[jt]);
        }
        const bool wrap = ccq + 1 == ncq;
        chi += wrap ? 1 : 0;
        ccq = wrap ? 0 : ccq + 1;
    };
    if constexpr (RES == 1) {
        if (nq > 0) step(res[0]);
        if (nq > 1) step(res[1]);
    } else {
        if (nq > 0) {
            TileIn TA = fetch(), TB;
            for (int i = 0; i < (nq >> 1); ++i) {
                TB = fetch();          // fences: see tl_output_whole
                __builtin_amdgcn_sched_barrier(0);
                step(TA);
                __builtin_amdgcn_sched_barrier(0);
                TA = fetch();
                __builtin_amdgcn_sched_barrier(0);
                step(TB);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (nq & 1) step(TA);
        }
    }
    if constexpr (PK > 8) {
        // the waves' partials, added in wave order into gsum (which may alias the scratches: they are dead behind the first barrier)
        for (int wv = 0; wv < NWV; ++wv) {
            __syncthreads();
            if (wave == wv) {
#pragma unroll
                for (int jt = 0; jt < PK; ++jt) {
                    const int o = ((4 * jt + lk) * NSP + li) * 4;
                    f32x4 g = accJ[jt];
                    if (wv > 0) {
                        const f32x4 q = *reinterpret_cast<const f32x4*>(gsum + o);
#pragma unroll
                        for (int r = 0; r < 4; ++r) g[r] = q[r] + g[r];
                    }
                    if (wv == NWV - 1) {
                        const f32x4 xv = *reinterpret_cast<const f32x4*>(xL + o);
#pragma unroll
                        for (int r = 0; r < 4; ++r) g[r] = xv[r] > 0.0f ? g[r] : 0.0f;
                    }
                    *reinterpret_cast<f32x4*>(gsum + o) = g;
                }
            }
        }
        return;
    }
    // this wave's partial of dL/dx_L -> its scratch, in the activation layout
#pragma unroll
    for (int jt = 0; jt < PK; ++jt) *reinterpret_cast<f32x4*>(scr + ((4 * jt + lk) * NSP + li) * 4) = accJ[jt];
}

// out[i][s] = sum_j W[j][i] gpre[j][s]  (x relu'(xin[i][s]) when mask); W [N][K] row-major.
// W2 / gpre2 (same shape): + sum_j W2[j][i] gpre2[j][s] in the same accumulator (the two heads of the gated direct modes).
template <int NWV>
__device__ __forceinline__ void tl_hidden_bwd(const float* __restrict__ W, int N, int K, const float* gpre, const float* xin,
                                              bool mask, float* out, int wave, int lane, const float* __restrict__ W2 = nullptr,
                                              const float* gpre2 = nullptr, int ld = 0) {
    constexpr int NSP = 16;
    const int li = lane & 15, lk = lane >> 4;
    const int nkb = N >> 4;
    const int ldw = ld ? ld : K;      // row stride of W (and W2): K, or K + 4 for a resident LDS copy
    for (int it = wave; it < (K >> 4); it += NWV) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* wcol = W + (long long)(4 * lk) * ldw + 16 * it + li;
#pragma unroll 4
        for (int kb = 0; kb < nkb; ++kb) {
            float av[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) av[e] = wcol[(long long)(16 * kb + e) * ldw];
            const f32x4 Bv = *reinterpret_cast<const f32x4*>(gpre + ((4 * kb + lk) * NSP + li) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = mfma16(av[e], Bv[e], acc);
        }
        if (W2) {
            const float* wcol2 = W2 + (long long)(4 * lk) * ldw + 16 * it + li;
#pragma unroll 4
            for (int kb = 0; kb < nkb; ++kb) {
                float av[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) av[e] = wcol2[(long long)(16 * kb + e) * ldw];
                const f32x4 Bv = *reinterpret_cast<const f32x4*>(gpre2 + ((4 * kb + lk) * NSP + li) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = mfma16(av[e], Bv[e], acc);
            }
        }
        const int o = ((4 * it + lk) * NSP + li) * 4;
        if (mask) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xin + o);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = xv[r] > 0.0f ? acc[r] : 0.0f;
        }
        *reinterpret_cast<f32x4*>(out + o) = acc;
    }
}

// dW[j][i] += w sum_s gpre[j][s] xin[i][s] into this wave's accumulator tiles (tile tt = wave + 8 q <-> (jt, it))
template <int NWV, int DWT>
__device__ __forceinline__ void tl_dw_acc(const float* gpre, const float* xin, int N, int K, float w, f32x4 (&dw)[DWT], int wave,
                                          int lane) {
    constexpr int NSP = 16;
    const int li = lane & 15, lk = lane >> 4;
    const int nit = K >> 4, ntile = (N >> 4) * nit;
#pragma unroll
    for (int q = 0; q < DWT; ++q) {
        const int tt = wave + NWV * q;
        if (tt < ntile) {
            const int jt = tt / nit, it = tt - jt * nit;
            const float* ap = gpre + ((4 * jt + (li >> 2)) * NSP + lk) * 4 + (li & 3);
            const float* bp = xin + ((4 * it + (li >> 2)) * NSP + lk) * 4 + (li & 3);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) dw[q] = mfma16(w * ap[16 * ks], bp[16 * ks], dw[q]);
        }
    }
}

// The same into this workgroup's partial in GLOBAL memory (cooperative sweep: its register file holds the output-layer weights): every
// tile is owned by one wave, which reads, accumulates and writes it in program order (deterministic); the next tile's accumulators
// are requested before the current tile's MFMAs.
template <int NWV>
__device__ __forceinline__ void tl_dw_rmw(const float* gpre, const float* xin, int N, int K, float w, float* gW, int wave, int lane) {
    constexpr int NSP = 16;
    const int li = lane & 15, lk = lane >> 4;
    const int nit = K >> 4, ntile = (N >> 4) * nit;
    auto addr = [&](int tt) {
        const int jt = tt / nit, it = tt - jt * nit;
        return gW + (long long)(16 * jt + 4 * lk) * K + 16 * it + li;
    };
    auto fetch = [&](int tt) {
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (tt < ntile) {
            const float* p = addr(tt);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = p[(long long)r * K];
        }
        return v;
    };
    f32x4 cur = fetch(wave);
    for (int tt = wave; tt < ntile; tt += NWV) {
        const f32x4 nxt = fetch(tt + NWV);
        const int jt = tt / nit, it = tt - jt * nit;
        const float* ap = gpre + ((4 * jt + (li >> 2)) * NSP + lk) * 4 + (li & 3);
        const float* bp = xin + ((4 * it + (li >> 2)) * NSP + lk) * 4 + (li & 3);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) cur = mfma16(w * ap[16 * ks], bp[16 * ks], cur);
        float* p = addr(tt);
#pragma unroll
        for (int r = 0; r < 4; ++r) p[(long long)r * K] = cur[r];
        cur = nxt;
    }
}

}  // namespace

// RES = 1 ("everything resident", small square models: H = every layer width = 16 PK <= 64, H*C/16 <= 2 NWV output
// tiles): every weight fragment a wave needs -- its forward row tile and its transposed tile of the (at most two)
// hidden matrices, its (at most two) output tiles -- is loaded ONCE and stays in registers, so no phase of a stage
// waits on an L2 round trip (cfg4: 11.3 -> see DESIGN.md).
// RES = 2: only the hidden matrices are resident (their fragments cost 16 PK registers), the output tiles are streamed: any
// number of output tiles, gated heads, split records -- the hidden phases, six per stage, no longer start with an L2 round trip.
// BF = 1: x_L is additionally kept as three bf16 pieces (XBA, written by the last hidden layer) and THAT image is what pass B
// gets as its record A, so the recompute of P there runs on the bf16 matrix cores.
// DIRECT = 1: the evaluate / derivative input modes (see ncde_fwd_tiled): the stage input carries the control rows, layer 0
// arrives column-padded (a.din[0]; its gradient is written back with the real row stride a.d0), the heads are H-row dense
// layers whose VJP and parameter gradients (register accumulators, like the hidden layers') stay in this kernel -- no records,
// no pass B, one launch for the whole sweep.
// BIGH = 1 (round 5): 128 < H <= 256 with every hidden width <= 128 (the reference's hyper-parameter range draws hidden_dim up to
// 256, configurations.json5:34): the state slice per thread and the hidden-dW accumulator tiles per wave double twice over, which
// only the 512-register file of ONE wave per SIMD holds -- launched with NWV = 4.
// COOP = 1 (round 5): the output phase is XCD-cooperative and weight-stationary (ncde_coop.h) -- this workgroup keeps 20 row tiles
// of Wo in registers for the whole launch and applies them to the x_L of every sample tile of its group; its own tile's dL/dx_L is the
// sum of the group's partials.  One wave per SIMD; the hidden-layer weight gradients live in the workgroup's global partial.
template <int PK, int NWV, int RES = 0, int GATED = 0, int BF = 0, int DIRECT = 0, int BIGH = 0, int COOP = 0>
__global__ __launch_bounds__(64 * NWV) void ncde_adj_tiled(KArgs a) {
    static_assert(BF == 0 || (RES != 1 && PK >= 2), "split record: streamed output tiles, last hidden width a multiple of 32");
    static_assert(DIRECT == 0 || (RES == 0 && BF == 0), "direct heads: streamed fp32 layers");
    static_assert(BIGH == 0 || (RES == 0 && DIRECT == 0 && NWV == 4), "wide state: streamed weights, matmul input, one wave per SIMD");
    static_assert(PK <= 8 || (PK == 16 && BIGH == 1 && GATED == 0 && BF == 0 && COOP == 0), "256-wide last layer: the wide instantiation, fp32 records");
    static_assert(COOP == 0 || (PK == 2 * COOP_NCH && NWV == 8 && RES == 0 && GATED == 0 && BF == 1 && DIRECT == 0 && BIGH == 0),
                  "cooperative output phase: original field, last hidden width 128, split records, two waves per SIMD (P role / T role)");
    constexpr int NT = 64 * NWV, TL_EADJ = RES ? (16 * 16 * PK + NT - 1) / NT : (BIGH ? 4096 : 2048) / NT;
    // W16 (round 5): last hidden width 256.  Up to 256 hidden-dW tiles per matrix do not fit the register file either: like the cooperative
    // sweep it accumulates them in the workgroup's global partial (DWG), and the output phase's per-wave scratches alias the cotangent
    // buffers (see the LDS plan below and tl_output_vjp)
    constexpr bool W16 = PK == 16, DWG = COOP != 0 || W16;
    constexpr int TL_DWT = DWG ? 1 : (RES ? (PK * PK + NWV - 1) / NWV : (BIGH ? 128 : 64) / NWV);   // hidden dW tiles per wave and weight slot
    constexpr int COOP_LDS = 2 * 10 * 2 * 64 * 4 + 2 * 8 * 64 * 4 + 2 * 80 * 16 + 2 * 256 + 2 * 64 + 2 * 4 * 64 + 2 * 16 + 8;      // floats, see CDP .. CFL below
    constexpr int NSP = 16, SCW = W16 ? 16 * (16 * 8 + 4) : ((16 * (16 * PK + 4) > 16 * PK * NSP) ? 16 * (16 * PK + 4) : 16 * PK * NSP);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if constexpr (COOP == 0) {      // re-execution behind a cooperative launch sequence: only when it gave up (see KArgs.run_if)
        if (a.run_if != nullptr && *a.run_if == 0u) return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b0 = blockIdx.x * NSP;
    const int H = a.H, C = a.C, L = a.n_layers;
    // x_1 .. x_L and the two cotangent buffers hold layer OUTPUTS (and, in the direct modes, the padded field input): their row count
    // is the widest layer, not max(H, widest layer) -- with H = 256 over 128-wide layers the difference is 48 KB of LDS
    int D = DIRECT ? max(H, a.din[0]) : 16;
    for (int l = 0; l < L; ++l) D = max(D, a.dout[l]);
    const int HS = H * NSP, DS = D * NSP;
    const int US = DIRECT ? max(H, a.din[0]) * NSP : HS;      // direct modes: H + C (padded) rows of field input / of its cotangent
    float* YS = lds;               // stage input y (= x_0)
    float* AS = YS + US;           // stage cotangent
    float* KOY = AS + HS;          // f(y).dX of the stage
    // W16: J^T cotangent behind the cotangent ping-pong buffers, so that [G0 | G1 | KOA] -- all dead during the output phase -- is one
    // area of 2 DS + HS >= NWV SCW floats for the waves' scratches (the host checks that)
    float* X = KOY + HS + (W16 ? 0 : US);           // x_1 .. x_L
    float* G0 = X + L * DS;
    float* G1 = G0 + DS;
    float* KOA = W16 ? G1 + DS : KOY + HS;         // J^T cotangent of the stage
    float* DX = G1 + DS + (W16 ? HS : 0);           // [C/4][16][4]
    float* SC = W16 ? G0 : DX + C * NSP;      // per-wave scratch (COOP: the exchange area instead)
    float* scr = SC + (COOP ? 0 : wave * SCW);
    unsigned* XBA = reinterpret_cast<unsigned*>(SC + (COOP ? COOP_LDS : NWV * SCW));   // BF: split image of x_L (24 words per unit)
    const bool disc = a.discrete != 0;
    const int S = n_stages(a.method);
    const int dlast = DIRECT ? a.dout[L - 1] : 16 * PK;      // direct heads: any multiple of 16 (PK only sizes the unused tile scratch)
    const int last_row = a.n_out - 1;
    const int n_st = gridDim.x;
    // general time axis (a.plan): reverse step n = n_rsteps .. 1 is plan step rstep = n_rsteps - n of the adjoint table
    // (continuous adjoint) or the transpose of forward step m = n - 1 (discrete backward) -- as in ncde_variant.hip, whose
    // plan mode is pinned to the reference
    // (the all-resident variant, RES = 1, is the latency-critical one -- cfg4: +6 % with the plan's extra live state -- and runs
    // the default axis only: the host sends planned problems to RES = 2 / streamed kernels)
    const bool planned = RES != 1 && a.plan != nullptr;      // (round 6: the cooperative sweep walks the plan too)
    if (planned && !plan_header_ok(a, S)) return;      // (uniform: before the first barrier)
    const int pw_ = plan_step_words(S);
    const int* pfwd = planned ? a.plan + plan_off_fwd() : nullptr;
    const int* pout = planned ? a.plan + plan_off_out(S, a.n_steps_fwd) : nullptr;
    const int* padj = planned ? a.plan + plan_off_adj(S, a.n_steps_fwd, a.n_out) : nullptr;
    const int n_rsteps = planned ? (disc ? a.n_steps_fwd : a.n_steps_adj) : a.T - 1;
    auto step_of = [&](int n) { return planned ? (disc ? pfwd + (n - 1) * pw_ : padj + (n_rsteps - n) * pw_) : nullptr; };

    float y0[TL_EADJ], ky1[TL_EADJ], ky2[TL_EADJ], a0[TL_EADJ], ka1[TL_EADJ], ka2[TL_EADJ];
#pragma unroll
    for (int q = 0; q < TL_EADJ; ++q) {
        const int e = tid + q * NT;
        y0[q] = ky1[q] = ky2[q] = a0[q] = ka1[q] = ka2[q] = 0.0f;
        if (e < HS) {
            const int u = ((e >> 2) / NSP) * 4 + (e & 3), s = (e >> 2) % NSP, b = b0 + s;
            const bool live = b < a.B && u < a.Hr;      // (zero-padded problems: units >= Hr are not the caller's; they stay 0)
            const long long o = ((long long)b * a.n_out + last_row) * a.Hr + u;
            float g, y;
            if (a.resume) {      // a later time window: (y, a) at knot win_hi as the previous launch left them
                y = a.carry[(long long)blockIdx.x * HS + e];
                g = a.carry[((long long)gridDim.x + blockIdx.x) * HS + e];
            } else {
                g = live ? a.grad_out[o] : 0.0f;
                if (planned && disc && live) g = plan_out_cotangent(a, pfwd + (a.n_steps_fwd - 1) * pw_, pout, 1, (long long)b * a.n_out, u);
                y = (!disc && live) ? a.z_out[o] : 0.0f;
            }
            a0[q] = g;
            if (disc) {
                // cotangent of the LAST stage input of the step about to be transposed (step a.win_hi)
                const float dtl = planned ? __int_as_float(step_of(a.win_hi)[0]) : 1.0f;
                AS[e] = a.method == NCDE_RK4_38 ? (g * dtl) * 0.125f : dtl * g;
                YS[e] = 0.0f;
            } else {
                y0[q] = y;
                YS[e] = y;
                AS[e] = g;
            }
        }
    }
    if constexpr (DIRECT != 0) {
        for (int e = HS + tid; e < US; e += NT) YS[e] = 0.0f;      // rows H + C .. of the padded field input stay zero
        tl_fill_resident<NT>(a, lds, tid);                           // small matrices -> LDS (a barrier follows before the first stage)
    }
    // hidden-layer parameter gradients: at most two distinct (W, b) pairs (layer 0, and ONE matrix shared by the rest)
    f32x4 dw0[TL_DWT], dw1[TL_DWT];
    float db0 = 0.0f, db1 = 0.0f;
#pragma unroll
    for (int q = 0; q < TL_DWT; ++q) dw0[q] = dw1[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // DIRECT: the heads' parameter gradients, same scheme (tiles tt = wave + NWV q of [H][dlast])
    f32x4 dwo[DIRECT ? TL_DWT : 1], dwg[(DIRECT && GATED) ? TL_DWT : 1];
    float dbo = 0.0f, dbg = 0.0f;
#pragma unroll
    for (int q = 0; q < (DIRECT ? TL_DWT : 1); ++q) dwo[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < ((DIRECT && GATED) ? TL_DWT : 1); ++q) dwg[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.resume) {      // continue this workgroup's hidden-layer partial where the previous window stopped
        const float* gp = a.gpart + (long long)blockIdx.x * a.gstride;
        const int li = lane & 15, lk = lane >> 4;
        int l1 = -1;
        for (int l = 1; l < L; ++l)
            if (a.gW_off[l] != a.gW_off[0]) { l1 = l; break; }
#pragma unroll
        for (int slot = 0; slot < 2; ++slot) {
            const int l = slot == 0 ? 0 : l1;
            if (l < 0) continue;
            const int N = a.dout[l], K = a.din[l], nit = K >> 4, ntile = (N >> 4) * nit;
            if constexpr (!DWG) {
#pragma unroll
                for (int q = 0; q < TL_DWT; ++q) {
                    const int tt = wave + NWV * q;
                    if (tt < ntile) {
                        const int jt = tt / nit, it = tt - jt * nit;
                        f32x4 v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = gp[a.gW_off[l] + (16 * jt + 4 * lk + r) * K + 16 * it + li];
                        if (slot == 0) dw0[q] = v;
                        else dw1[q] = v;
                    }
                }
            }
            if (tid < N) {
                if (slot == 0) db0 = gp[a.gb_off[l] + tid];
                else db1 = gp[a.gb_off[l] + tid];
            }
        }
    }

    // ---- resident weight fragments (RES) -------------------------------------------------------------------------------
    Panel<PK> wf[2];          // forward row tile `wave` of the layer-0 matrix / of the other matrix
    f32x4 bfv[2];
    float wb[2][4 * PK];      // transposed tile `wave` (input units 16 wave ..) of the same two matrices
    WoTile<PK> wo[2];
    if constexpr (RES != 0) {
        const int li = lane & 15, lk = lane >> 4;
        int l1 = 0;
        for (int l = 1; l < L; ++l)
            if (a.gW_off[l] != a.gW_off[0]) { l1 = l; break; }
        const int K = dlast;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int l = sl == 0 ? 0 : l1;
            const int tw = wave < PK ? wave : 0;
            wf[sl] = tl_load_panel<PK>(a.W[l] + (long long)(16 * tw + li) * K + 4 * lk, 0);
            bfv[sl] = *reinterpret_cast<const f32x4*>(a.b[l] + 16 * tw + 4 * lk);
#pragma unroll
            for (int kb = 0; kb < PK; ++kb)
#pragma unroll
                for (int e = 0; e < 4; ++e) wb[sl][4 * kb + e] = a.W[l][(long long)(16 * kb + 4 * lk + e) * K + 16 * tw + li];
        }
        if constexpr (RES == 1) {
            const int nhb = H >> 2, ncq = C >> 2;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                int hb = wave + NWV * (q / ncq), cq = q % ncq;
                if (hb >= nhb) { hb = 0; cq = 0; }
                wo[q].P = tl_load_panel<PK>(a.Wo + (long long)((4 * hb + (li >> 2)) * C + 4 * cq + (li & 3)) * dlast + 4 * lk, 0);
                wo[q].bias = *reinterpret_cast<const f32x4*>(a.bo + (4 * hb + lk) * C + 4 * cq);
            }
        }
    }

    // ---- cooperative output phase: ids, exchange area, resident weights (ncde_coop.h) ----------------------------------------
    CoopWeights cw;
    CoopDims cd{};
    CoopSync csy{};
    __amdgpu_buffer_rsrc_t crs = coop_rsrc(a.coop_x);      // (a NULL base outside the cooperative mode: never dereferenced)
    int c_grp = 0, c_mem = 0, c_hbw = 0;
    bool c_same = false;      // every member of the group on one XCD: plain payload stores (see coop_st16)
    // exchange area (floats): dP operands of the group tile in flight [2][10 pairs][2 pieces][64 lanes][4], then the STAGED inputs of
    // the next tile -- x_L image [2][8][64][4], dX/dt [2][C x 16], the a-rows of this member's state units [2][256], scales [2][64] --,
    // the waves' f.dX partials [2][4][64] and the barrier flag.  The maxima / scales of the owner phase alias the head of the dP area.
    float* const CDP = SC;
    float* const CBX = CDP + 2 * 10 * 2 * 64 * 4;
    float* const CDX = CBX + 2 * 8 * 64 * 4;
    float* const CAS = CDX + 2 * 80 * 16;
    float* const CSC = CAS + 2 * 256;
    float* const CKX = CSC + 2 * 64;
    float* const CIS = CKX + 2 * 4 * 64;                     // [2][16]     1/(sd sw) of the tile the T role is working on
    int* const CFL = reinterpret_cast<int*>(CIS + 2 * 16);
    float* const CBO = reinterpret_cast<float*>(XBA) + 2048;      // [4][5][4][4] bias of this member's rows (keeper phase only; behind the parked a0)
    float* const CMX = CDP;                                  // [3][8][16]  per-wave maxima of |x_L|, |a|, |dX/dt| per sample
    float* const CSG = CDP + 3 * 8 * 16;                     // [4][16]     sx, 1/(sx sw), sd, 1/(sd sw) of this tile
    if constexpr (COOP != 0) {
        cd.H = H; cd.C = C; cd.dlast = dlast; cd.M = a.coop_M; cd.G = a.coop_G;
        csy.words = a.coop_sync; csy.G = a.coop_G; csy.status = a.coop_status; csy.spin = a.coop_spin;
        c_grp = blockIdx.x % a.coop_G;
        c_mem = blockIdx.x / a.coop_G;
        c_hbw = (c_mem * COOP_RPM + (wave & 3) * COOP_NRT) / (C >> 2);      // P role: the state-unit block every row tile of this wave belongs to
        coop_load_weights<0, COOP_PIN>(cw, a.coop_img, c_mem, wave, lane);      // the pinned part of this wave's 40 fragments
        const int same = coop_same_xcd(csy, c_grp, c_mem, a.coop_M, (int)gridDim.x, CFL, tid);
        if (same < 0) return;      // (another workgroup never arrived: the launch is abandoned, grad_z0 keeps the caller's fill)
        c_same = same != 0;
    }

    // dX/dt (and, for the discrete backward, the recorded stage input) of stage (n, j) are fetched into registers one
    // stage ahead and stored to LDS in the bookkeeping phase, so no stage starts with an exposed global round trip
    constexpr int DXMAXC = NWV == 8 ? 80 : 160;      // widest control the register prefetch holds (3 / 10 values per thread)
    constexpr int DXE = (NSP * DXMAXC + NT - 1) / NT, YSE = TL_EADJ;
    float dxn[DXE], ysn[YSE];
    // round 5: more channels than that (the reference has no limit; VERDICT round 4, missing item 2) are read in the bookkeeping phase
    // itself -- one exposed global round trip per stage -- instead of one stage ahead through registers
    const bool wide_c = C > DXMAXC;
    StageDesc sdn;      // descriptor of the stage whose inputs are being fetched
    const float* recn = nullptr;      // discrete backward: its recorded stage input
    auto stage_desc = [&](int n, int j) {
        if (planned) return plan_stage(step_of(n), disc ? S - 1 - j : j);
        return default_stage(a.method, disc ? (float)(n - 1) + stage_offset(a.method, S - 1 - j) : -(-(float)n + stage_offset(a.method, j)), a.n_pieces);
    };
    auto dx_value = [&](int e, const StageDesc& sd) {
        const int idx = sd.idx;
        const float frac = sd.frac, kdt = sd.kdt;
        float v = 0.0f;
        const int s = e / C, c = e - s * C, b = b0 + s;
        const int Cc = a.Cc;      // channels of the coefficient tensor (zero-padded problems: < C)
        if (b < a.B && c < Cc) {
            const float* cp = a.coeffs + (long long)b * a.cs_b + (long long)idx * a.cs_t;
            const bool value = DIRECT != 0 && a.field_input == NCDE_INPUT_EVALUATE;      // X(t) instead of dX/dt(t)
            if (a.interp == NCDE_INTERP_LINEAR) {
                v = cp[a.cs_t + c] - cp[c];
                if (value) v = cp[c] + (frac * v) / kdt;
                else if (kdt != 1.0f) v = v / kdt;
            } else {
                const float bb = cp[Cc + c], cc = cp[2 * Cc + c], dd = cp[3 * Cc + c];
                if (value) {
                    float inner = 0.5f * cc + (dd * frac) / 3.0f;
                    inner = bb + inner * frac;
                    v = cp[c] + inner * frac;
                } else {
                    const float inner = cc + dd * frac;
                    v = bb + inner * frac;
                }
            }
        }
        return v;
    };
    auto prefetch = [&](int n, int j) {
        sdn = stage_desc(n, j);
        if (!wide_c) {
#pragma unroll
            for (int q = 0; q < DXE; ++q) {
                const int e = tid + q * NT;
                dxn[q] = e < NSP * C ? dx_value(e, sdn) : 0.0f;
            }
        }
        recn = disc ? a.stages + ((long long)((n - 1) * S + (S - 1 - j)) * (a.Brec ? a.Brec : a.B) + b0) * a.Hr : nullptr;
        if (disc) {
            const int Hr = a.Hr;
            const float* rec = recn;
#pragma unroll
            for (int q = 0; q < YSE; ++q) {
                const int e = tid + q * NT;
                float v = 0.0f;
                if (e < NSP * Hr) {
                    const int s = e / Hr;
                    if (b0 + s < a.B) v = rec[e];
                }
                ysn[q] = v;
            }
        }
    };
    auto publish = [&]() {
        auto put = [&](int e, float v) {
            const int s = e / C, c = e - s * C;
            if constexpr (DIRECT != 0) YS[(((H + c) >> 2) * NSP + s) * 4 + ((H + c) & 3)] = v;      // control rows of the field input
            else DX[((c >> 2) * NSP + s) * 4 + (c & 3)] = v;
        };
        if (wide_c) {
            for (int e = tid; e < NSP * C; e += NT) put(e, dx_value(e, sdn));
        } else {
#pragma unroll
            for (int q = 0; q < DXE; ++q) {
                const int e = tid + q * NT;
                if (e < NSP * C) put(e, dxn[q]);
            }
        }
        if (disc) {
#pragma unroll
            for (int q = 0; q < YSE; ++q) {
                const int e = tid + q * NT;
                if (e < NSP * a.Hr) {
                    const int s = e / a.Hr, u = e - s * a.Hr;
                    YS[((u >> 2) * NSP + s) * 4 + (u & 3)] = ysn[q];
                }
            }
        }
    };
    prefetch(a.win_hi, 0);
    publish();
    __syncthreads();

    int sc = 0;      // stage counter within this time window = record index
#ifdef NCDE_TL_PROF
    unsigned long long tprof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();      // 8 .. 11: cooperative phases
    unsigned long long kprof[3] = {0, 0, 0}, klast = 0;      // keeper loop: compute | DMA wait | barrier
#endif
    for (int n = a.win_hi; n > a.win_lo; --n) {
        const int* pstep = step_of(n);
        const float dt = planned ? __int_as_float(pstep[0]) : 1.0f;
        for (int j = 0; j < S; ++j, ++sc) {
            const float w = disc ? 1.0f : stage_weight(a.method, j) * dt;
            if constexpr (COOP == 0) {   // next stage's inputs; consumed (publish) in this stage's bookkeeping phase
                const int jn = j + 1 < S ? j + 1 : 0, nn = j + 1 < S ? n : n - 1;
                if (nn > a.win_lo) prefetch(nn, jn);
            }
            // ---- forward recompute, keeping x_1 .. x_L -------------------------------------------------------------
            const float* in = YS;
            for (int l = 0; l < L; ++l) {
                float* outb = X + l * DS;
                if constexpr (RES != 0) {
                    if (wave < PK) {
                        const int li = lane & 15, lk = lane >> 4;
                        const bool s0 = a.gW_off[l] == a.gW_off[0];
                        f32x4 acc[1];
                        acc[0] = s0 ? bfv[0] : bfv[1];
                        if (s0) tl_mma_panel<1, PK>(wf[0], in, 0, li, lk, acc);
                        else tl_mma_panel<1, PK>(wf[1], in, 0, li, lk, acc);
                        f32x4 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = relu_dev(acc[0][r]);
                        *reinterpret_cast<f32x4*>(outb + ((4 * wave + lk) * NSP + li) * 4) = o;
                        if constexpr (BF != 0) {
                            if (l == L - 1) {      // split image of x_L (same layout as tl_dense_relu_pk's xb), row tile t = wave
                                unsigned h0, m0, l0, h1, m1, l1;
                                split_pair(o[0], o[1], h0, m0, l0);
                                split_pair(o[2], o[3], h1, m1, l1);
                                unsigned* dst = XBA + ((((wave >> 1) * 3) * 4 + 2 * (wave & 1) + (lk >> 1)) * 16 + li) * 4 + 2 * (lk & 1);
                                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                                *reinterpret_cast<u32x2*>(dst) = (u32x2){h0, h1};
                                *reinterpret_cast<u32x2*>(dst + 256) = (u32x2){m0, m1};
                                *reinterpret_cast<u32x2*>(dst + 512) = (u32x2){l0, l1};
                            }
                        }
                    }
                } else {
                    const TlW wr_ = tl_wref(lds, DIRECT != 0 ? a.tres[l] : 0, a.W[l], a.b[l], a.dout[l], a.din[l]);
                    tl_dense_relu<1, NWV, 0, 0, COOP ? 4 : 8>(wr_.W, wr_.b, a.dout[l], a.din[l], in, outb, wave, lane, (BF != 0 && l == L - 1) ? XBA : nullptr, wr_.ld);
                }
                TL_SYNC(0)
                in = outb;
            }
            auto write_records = [&]() {
                    const long long tile = (long long)sc * n_st + blockIdx.x;
                    float* ra = a.recA + tile * (BF != 0 ? dlast * 24 : dlast * NSP);
                    float* rc = a.recC + tile * (H * NSP);
                    float* rd = a.recD + tile * (C * NSP);
                    if constexpr (BF != 0) {
                        for (int e = tid; e < dlast * 6; e += NT)
                            reinterpret_cast<u32x4*>(ra)[e] = reinterpret_cast<const u32x4*>(XBA)[e];
                        // record B, split and PAIRED: sample tiles 2i and 2i+1 share one block, the K = 32 samples of the bf16 MFMA
                        // that accumulates dWo in pass B.  Word ((jt * 3 + piece) * 64 + lane) * 4 + 2 half + d of the pair's block
                        // holds x_L[k = 16 jt + (lane & 15)][samples 4 (lane >> 4) + 2 d, + 1] of tile `half`.
                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                        unsigned* rbp = reinterpret_cast<unsigned*>(a.recB) + ((long long)sc * ((n_st + 1) >> 1) + (blockIdx.x >> 1)) * (dlast * 48);
                        const int half = blockIdx.x & 1;
                        for (int e = tid; e < dlast * 4; e += NT) {
                            const int k = e >> 2, kg = e & 3, ln = (k & 15) + 16 * kg;
                            float v[4];
    #pragma unroll
                            for (int q = 0; q < 4; ++q) v[q] = in[((k >> 2) * NSP + 4 * kg + q) * 4 + (k & 3)];
                            unsigned h0, m0, l0, h1, m1, l1;
                            split_pair(v[0], v[1], h0, m0, l0);
                            split_pair(v[2], v[3], h1, m1, l1);
                            unsigned* dst = rbp + (((k >> 4) * 3) * 64 + ln) * 4 + 2 * half;
                            *reinterpret_cast<u32x2*>(dst) = (u32x2){h0, h1};
                            *reinterpret_cast<u32x2*>(dst + 256) = (u32x2){m0, m1};
                            *reinterpret_cast<u32x2*>(dst + 512) = (u32x2){l0, l1};
                            if (half == 0 && blockIdx.x + 1 == n_st) {      // odd tile count: the missing partner contributes zeros
                                *reinterpret_cast<u32x2*>(dst + 2) = (u32x2){0u, 0u};
                                *reinterpret_cast<u32x2*>(dst + 258) = (u32x2){0u, 0u};
                                *reinterpret_cast<u32x2*>(dst + 514) = (u32x2){0u, 0u};
                            }
                        }
                    } else {
                        float* rb = a.recB + tile * (dlast * NSP);
                        for (int e = tid; e < dlast * NSP; e += NT) {
                            ra[e] = in[e];
                            const int jj = e >> 4, s = e & 15;
                            rb[e] = in[((jj >> 2) * NSP + s) * 4 + (jj & 3)];
                        }
                    }
                    for (int e = tid; e < H * NSP; e += NT) {
                        const int hh = e >> 4, s = e & 15;
                        rc[e] = w * AS[((hh >> 2) * NSP + s) * 4 + (hh & 3)];
                    }
                    for (int e = tid; e < C * NSP; e += NT) rd[e] = DX[e];
            };
            if constexpr (DIRECT != 0) {
                // ---- direct heads: m = tanh(Wo x_L + bo) (x sigmoid(Wg x_L + bg)) IS dz/dt; cotangents of the two pre-activations ----
                float* DPT = G0;      // dL/dPt  [H][16]
                float* SG = SC;       // sigmoid(Pg), then dL/dPg (the per-wave scratch area is free in this mode)
                const TlW wo_ = tl_wref(lds, a.tres_o, a.Wo, a.bo, H, dlast);
                const TlW wg_ = tl_wref(lds, GATED != 0 ? a.tres_g : 0, a.Wg, a.bg, H, dlast);      // resident together with Wo or not at all
                tl_dense_relu<1, NWV, 1>(wo_.W, wo_.b, H, dlast, in, KOY, wave, lane, nullptr, wo_.ld);
                if constexpr (GATED != 0) tl_dense_relu<1, NWV, 2>(wg_.W, wg_.b, H, dlast, in, SG, wave, lane, nullptr, wg_.ld);
                __syncthreads();
                for (int e = tid; e < HS; e += NT) {
                    const float th = KOY[e], dm = AS[e];
                    if constexpr (GATED != 0) {
                        const float sg = SG[e];
                        KOY[e] = sg * th;
                        DPT[e] = (dm * sg) * (1.0f - th * th);
                        SG[e] = (dm * th) * (sg * (1.0f - sg));
                    } else {
                        DPT[e] = dm * (1.0f - th * th);
                    }
                }
                __syncthreads();
                if (w != 0.0f) {      // head parameter gradients (x_L = `in`)
                    tl_dw_acc<NWV, TL_DWT>(DPT, in, H, dlast, w, dwo, wave, lane);
                    if constexpr (GATED != 0) tl_dw_acc<NWV, TL_DWT>(SG, in, H, dlast, w, dwg, wave, lane);
                    if (tid < H) {
                        float sum = 0.0f, sumg = 0.0f;
#pragma unroll
                        for (int s = 0; s < NSP; ++s) {
                            sum += DPT[((tid >> 2) * NSP + s) * 4 + (tid & 3)];
                            if constexpr (GATED != 0) sumg += SG[((tid >> 2) * NSP + s) * 4 + (tid & 3)];
                        }
                        dbo += w * sum;
                        dbg += w * sumg;
                    }
                }
                // dL/dpre_L = (Wo^T dPt + Wg^T dPg) relu'(x_L)
                tl_hidden_bwd<NWV>(wo_.W, H, dlast, DPT, in, true, G1, wave, lane, GATED != 0 ? wg_.W : nullptr, GATED != 0 ? SG : nullptr, wo_.ld);
                __syncthreads();
            } else {
                // ---- output layer: f, dP, per-wave partial of dL/dx_L -----------------------------------------------------
                if constexpr (COOP != 0) {
                    const int li = lane & 15, lk = lane >> 4;
                    const long long my_base = (long long)blockIdx.x * cd.per_tile();
                    const unsigned bar1 = (2u * (unsigned)sc + 1u) * (unsigned)cd.M, bar2 = bar1 + (unsigned)cd.M;
                    // -- OWNER: per-sample maxima -> power-of-two scales ------------------------------------------------------
                    {
                        float mx = 0.0f, ma = 0.0f, md = 0.0f;      // this thread's elements all belong to sample (tid >> 2) & 15
                        for (int e = tid; e < dlast * NSP; e += NT) mx = fmaxf(mx, fabsf(in[e]));
                        for (int e = tid; e < HS; e += NT) ma = fmaxf(ma, fabsf(AS[e]));
                        for (int e = tid; e < C * NSP; e += NT) md = fmaxf(md, fabsf(DX[e]));
                        mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
                        ma = fmaxf(ma, __shfl_xor(ma, 1, 64)); ma = fmaxf(ma, __shfl_xor(ma, 2, 64));
                        md = fmaxf(md, __shfl_xor(md, 1, 64)); md = fmaxf(md, __shfl_xor(md, 2, 64));
                        if ((lane & 3) == 0) {
                            const int s_ = (lane >> 2) & 15;
                            CMX[(0 * 8 + wave) * 16 + s_] = mx; CMX[(1 * 8 + wave) * 16 + s_] = ma; CMX[(2 * 8 + wave) * 16 + s_] = md;
                        }
                    }
                    __syncthreads();
                    if (tid < 16) {
                        float m3[3];
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            float m_ = 0.0f;
#pragma unroll
                            for (int wv = 0; wv < 8; ++wv) m_ = fmaxf(m_, CMX[(k * 8 + wv) * 16 + tid]);
                            m3[k] = m_;
                        }
                        const float sw_inv = a.coop_scale[1];
                        const float sx = coop_pow2_scale(m3[0]), sd = coop_pow2_scale(m3[1] * m3[2]);
                        const float isx = coop_pow2_inv(sx) * sw_inv, isd = coop_pow2_inv(sd) * sw_inv;
                        CSG[tid] = sx; CSG[16 + tid] = isx; CSG[32 + tid] = sd; CSG[48 + tid] = isd;
                        coop_st4(crs, a.coop_x, c_same, my_base + cd.off_sc() + tid, sx); coop_st4(crs, a.coop_x, c_same, my_base + cd.off_sc() + 16 + tid, isx);
                        coop_st4(crs, a.coop_x, c_same, my_base + cd.off_sc() + 32 + tid, sd); coop_st4(crs, a.coop_x, c_same, my_base + cd.off_sc() + 48 + tid, isd);
                        // records for ncde_dwo_h2: the samples of a tile PAIR are the K dimension of its dWo products, so x_L^T carries ONE
                        // power-of-two scale per tile (u_T, from the tile's largest |x_L|) and dP one per time window -- sigma / u_T, sigma
                        // from the window's largest (weighted cotangent bound / u_T), gathered here with an atomic max of float bits
                        float mt = m3[0], bt = fabsf(w) * (m3[1] * m3[2]);
#pragma unroll
                        for (int o = 8; o > 0; o >>= 1) { mt = fmaxf(mt, __shfl_xor(mt, o, 64)); bt = fmaxf(bt, __shfl_xor(bt, o, 64)); }
                        const float uT = coop_pow2_scale(mt), iuT = coop_pow2_inv(uT);
                        float* rs = a.recS + ((long long)sc * n_st + blockIdx.x) * 32;
                        rs[tid] = isx;
                        if (tid == 0) {
                            CSG[64] = uT;
                            rs[16] = uT; rs[17] = iuT;
                            const float bq = bt * iuT;
                            if (bq > 0.0f && bq < 3.0e38f) atomicMax(a.win_max, __float_as_uint(bq));
                        }
                    }
                    __syncthreads();
                    // -- OWNER: publish x_L (scaled, split-fp16, B-operand order), a, dX/dt ------------------------------------
                    if (wave < 4) {
                        const int c = wave, ub = 32 * c + 8 * lk;      // K chunk c of this wave; units ub .. ub + 7 of sample li
                        const f32x4 v0 = *reinterpret_cast<const f32x4*>(in + (((ub >> 2)) * NSP + li) * 4);
                        const f32x4 v1 = *reinterpret_cast<const f32x4*>(in + (((ub >> 2) + 1) * NSP + li) * 4);
                        const float sx = CSG[li];
                        unsigned h[4], l[4];
                        coop_split2(v0[0] * sx, v0[1] * sx, h[0], l[0]); coop_split2(v0[2] * sx, v0[3] * sx, h[1], l[1]);
                        coop_split2(v1[0] * sx, v1[1] * sx, h[2], l[2]); coop_split2(v1[2] * sx, v1[3] * sx, h[3], l[3]);
                        coop_st16(crs, a.coop_x, c_same, my_base + ((c * 2 + 0) * 64 + lane) * 4, (u32x4){h[0], h[1], h[2], h[3]});
                        coop_st16(crs, a.coop_x, c_same, my_base + ((c * 2 + 1) * 64 + lane) * 4, (u32x4){l[0], l[1], l[2], l[3]});
                    }
                    {
                        for (int e = tid; e < HS / 4; e += NT) coop_st16(crs, a.coop_x, c_same, my_base + cd.off_as() + e * 4, *reinterpret_cast<const u32x4*>(AS + e * 4));
                        for (int e = tid; e < C * NSP / 4; e += NT) coop_st16(crs, a.coop_x, c_same, my_base + cd.off_dx() + e * 4, *reinterpret_cast<const u32x4*>(DX + e * 4));
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (tid == 0 && !(a.coop_inject != 0 && blockIdx.x == 1 && sc == 0 && a.resume == 0)) coop_arrive(csy, c_grp);      // (fault injection)
                    TL_TICK(8)
                    coop_load_weights<COOP_PIN, 40>(cw, a.coop_img, c_mem, wave, lane);      // the re-read part: in flight under the records and while the group assembles
                    {   // records for ncde_dwo_h2 (needs a, dX/dt, x_L, the scales: all final -- and dead for this stage afterwards)
                        const long long tile = (long long)sc * n_st + blockIdx.x;
                        if (wave < 4) {      // record A: the scaled split-fp16 image of x_L exactly as published to the keepers (8 KB)
                            const int c = wave, ub = 32 * c + 8 * lk;
                            const f32x4 v0 = *reinterpret_cast<const f32x4*>(in + (((ub >> 2)) * NSP + li) * 4);
                            const f32x4 v1 = *reinterpret_cast<const f32x4*>(in + (((ub >> 2) + 1) * NSP + li) * 4);
                            const float sx = CSG[li];
                            unsigned h[4], l[4];
                            coop_split2(v0[0] * sx, v0[1] * sx, h[0], l[0]); coop_split2(v0[2] * sx, v0[3] * sx, h[1], l[1]);
                            coop_split2(v1[0] * sx, v1[1] * sx, h[2], l[2]); coop_split2(v1[2] * sx, v1[3] * sx, h[3], l[3]);
                            unsigned* ra = reinterpret_cast<unsigned*>(a.recA) + tile * 2048;
                            *reinterpret_cast<u32x4*>(ra + ((c * 2 + 0) * 64 + lane) * 4) = (u32x4){h[0], h[1], h[2], h[3]};
                            *reinterpret_cast<u32x4*>(ra + ((c * 2 + 1) * 64 + lane) * 4) = (u32x4){l[0], l[1], l[2], l[3]};
                        }
                        // record B: x_L^T of the tile PAIR (2i, 2i + 1), scaled by this tile's u_T, two fp16 pieces: word
                        // ((jt * 2 + piece) * 64 + lane) * 4 + 2 half + d = x_L[16 jt + (lane & 15)][samples 4 (lane >> 4) + 2 d, + 1] of tile `half`
                        {
                            const float uT = CSG[64];
                            unsigned* rbp = reinterpret_cast<unsigned*>(a.recB) + ((long long)sc * (n_st >> 1) + (blockIdx.x >> 1)) * 4096;
                            const int half = blockIdx.x & 1;
                            for (int e = tid; e < 128 * 4; e += NT) {
                                const int k = e >> 2, kg = e & 3, ln = (k & 15) + 16 * kg;
                                float v[4];
#pragma unroll
                                for (int q = 0; q < 4; ++q) v[q] = in[((k >> 2) * NSP + 4 * kg + q) * 4 + (k & 3)] * uT;
                                unsigned h0, l0, h1, l1;
                                coop_split2(v[0], v[1], h0, l0);
                                coop_split2(v[2], v[3], h1, l1);
                                unsigned* dst = rbp + (((k >> 4) * 2) * 64 + ln) * 4 + 2 * half;
                                *reinterpret_cast<u32x2c*>(dst) = (u32x2c){h0, h1};
                                *reinterpret_cast<u32x2c*>(dst + 256) = (u32x2c){l0, l1};
                            }
                        }
                        float* rc = a.recC + tile * (H * NSP);      // w a: [h][16 samples]
                        for (int e = tid; e < H * NSP; e += NT) {
                            const int hh = e >> 4, s_ = e & 15;
                            rc[e] = w * AS[((hh >> 2) * NSP + s_) * 4 + (hh & 3)];
                        }
                        float* rd = a.recD + tile * (C * NSP);      // dX/dt: [c][16 samples] (the per-workgroup sweep's record is [c/4][16][4])
                        for (int e = tid; e < C * NSP; e += NT) {
                            const int cc = e >> 4, s_ = e & 15;
                            rd[e] = DX[((cc >> 2) * NSP + s_) * 4 + (cc & 3)];
                        }
                    }
                    __syncthreads();
                    // the Butcher k-registers wait in LDS arrays that are dead until the reduction (KOY, KOA, G0, G1: 2048 floats each)
#pragma unroll
                    for (int q = 0; q < TL_EADJ; ++q) {
                        if (q * NT < HS) {      // (H < 128: the upper slots of the state slice are unused, the LDS arrays hold HS floats)
                            KOY[q * NT + tid] = ky1[q]; KOA[q * NT + tid] = ky2[q];
                            G0[q * NT + tid] = ka1[q]; G1[q * NT + tid] = ka2[q];
                            AS[q * NT + tid] = y0[q]; reinterpret_cast<float*>(XBA)[q * NT + tid] = a0[q];      // (a and the split image: recorded above)
                        }
                    }
                    // the bias of this member's rows -> the free tail of the split-image area (rewritten by the last hidden layer every stage):
                    // a global load inside the keeper loop would wait -- vmcnt is in order -- for the staging loads of the next tile
                    coop_fill_bias(a.bo, C, c_mem, CBO, tid);
                    if (!coop_wait(csy, c_grp, bar1, CFL, tid)) {      // timeout / another workgroup gave up: poison this tile's gradient
                        for (int e = tid; e < NSP * a.Hr; e += NT)
                            if (b0 + e / a.Hr < a.B) a.grad_z0[(long long)b0 * a.Hr + e] = __builtin_nanf("");
                        return;
                    }
                    TL_TICK(9)
#ifdef NCDE_TL_PROF
                    tprof[6] += c_same ? 1000 : 0;      // (development: column 6 = 1000 when the group shares an XCD)
#endif
                    // -- KEEPER: the rows of Wo this workgroup holds x every sample tile of the group -----------------------------
                    // Iteration `it` of the loop: the P role (waves 0..3) takes sample tile it -- P over its 5 row tiles, tanh, f.dX
                    // partial, dP -> LDS as the split-fp16 B operands of the 10 row-tile pairs -- while the T role (waves 4..7)
                    // takes tile it - 1: Wo^T dP over its 2 column tiles x all 10 pairs -> the member's partial of dL/dx_L straight to
                    // the exchange area.  All threads fetch tile it + 1's inputs into LDS meanwhile.  One workgroup barrier per
                    // iteration; dP, the staged inputs and the small exchange arrays are double-buffered by the parity of `it`.
                    const int ncq = C >> 2, jh = COOP_RPM / ncq, wph = 4 / jh;      // state-unit blocks per member, P waves per block
                    const bool prole = wave < 4;
                    const int rw = wave & 3;      // wave index within its role
                    const int cq0 = (c_mem * COOP_RPM + rw * COOP_NRT) % ncq;      // first channel quad of this wave's row tiles (5 | ncq: no wrap)
                    // staging of one tile's inputs: thread t moves piece t of the x_L image (512) and one piece of {dX/dt (<= 320) |
                    // the a-rows of this member's state units (<= 64, threads 320 ..) | the scales (16, threads 384 ..)}
                    const int n_dx4 = C * NSP / 4, n_as4 = jh * 16;
                    // LDS-DMA (global_load_lds_dwordx4, sc1): a wave's 64 lanes land in 1 KB of LDS starting at a wave-uniform base, no
                    // register in between; the issuing wave's vmcnt covers it, the workgroup barrier publishes it
                    typedef const __attribute__((address_space(1))) void* gptr_t;
                    typedef __attribute__((address_space(3))) void* lptr_t;
                    // Which extra piece a wave fetches is WAVE-uniform (scalar branches, no exec-mask sequences in the loop): waves below
                    // n_dxw whole 1 KB blocks of dX/dt (lanes past C x 16 floats read on into the tile's scale / f.dX area: never used, and
                    // the 1280-float slot holds five blocks), wave 5 the 64 a-rows from this member's first state unit on (jh x 16 of them
                    // are its own), wave 6 -- its first 16 lanes -- the scales
                    const int n_dxw = (n_dx4 + 63) >> 6;
                    (void)n_as4;
                    auto stage_load = [&](int tile, int buf, int tid) {
                        const float* src = a.coop_x + (long long)tile * cd.per_tile();
                        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), w64 = wv * 64;      // (this wave; its first thread)
                        __builtin_amdgcn_global_load_lds((gptr_t)(src + tid * 4), (lptr_t)(CBX + buf * 2048 + w64 * 4), 16, 0, 16);
                        if (wv < n_dxw) __builtin_amdgcn_global_load_lds((gptr_t)(src + cd.off_dx() + tid * 4), (lptr_t)(CDX + buf * 1280 + w64 * 4), 16, 0, 16);
                        else if (wv == 5)
                            __builtin_amdgcn_global_load_lds((gptr_t)(src + cd.off_as() + (c_mem * jh * NSP) * 4 + (tid - 320) * 4), (lptr_t)(CAS + buf * 256), 16, 0, 16);
                        else if (wv == 6 && tid < 400)
                            __builtin_amdgcn_global_load_lds((gptr_t)(src + cd.off_sc() + (tid - 384) * 4), (lptr_t)(CSC + buf * 64), 16, 0, 16);
                    };
                    auto stage_store = [&](int, int) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
                    stage_load(c_grp, 0, tid);
                    stage_store(0, tid);
                    __syncthreads();
                    const int tid_outer = tid;
#ifdef NCDE_TL_PROF
                    klast = __builtin_readcyclecounter();
#endif
                    const int hl_w_outer = rw / wph, cq0_outer = cq0;      // P role: which of this member's state-unit blocks this wave's row tiles belong to
                    // (one loop per role -- same trip count, same barriers -- so that neither carries the other's live registers)
                    if (prole) {
                        for (int it = 0; it <= cd.M; ++it) {
                            // (thread-id derived offsets are re-derived from an opaque copy every iteration: hoisted out of the loop they were
                            // spilled, and every scratch reload waits -- vmcnt is in order -- for the staging loads of the NEXT tile too)
                            int tid = tid_outer;
                            asm volatile("" : "+v"(tid));
                            const int lane = tid & 63, li = tid & 15, lk = (tid >> 4) & 3;
                            const int buf = it & 1;
                            int hl_w = hl_w_outer, cq0 = cq0_outer;      // (opaque copies too: an address base derived from them was hoisted and spilled)
                            asm volatile("" : "+v"(hl_w), "+v"(cq0));
                            if (it + 1 < cd.M) stage_load(c_grp + cd.G * (it + 1), buf ^ 1, tid);
                            if (it < cd.M) {
                                const float* bxs = CBX + buf * 2048;
                                const float isx = CSC[buf * 64 + 16 + li], sd = CSC[buf * 64 + 32 + li];
                                if (wave == 0 && lane < 16) CIS[buf * 16 + lane] = CSC[buf * 64 + 48 + lane];
                                const float sxw = coop_pow2_inv(isx);      // sx sw: the bias joins the scaled accumulator exactly
                                // (isx and sd are powers of two: folding them into the tanh pre-scale and into the cotangent changes no bit)
                                const float isx2 = isx * NCDE_TANH_PRESCALE;
                                const float asd = CAS[buf * 256 + (hl_w * NSP + li) * 4 + lk] * sd;
                                float kk = 0.0f;
                                unsigned* dpx = reinterpret_cast<unsigned*>(CDP) + buf * (10 * 2 * 64 * 4);
                                // P = Wo x_L on the f16 matrix cores, main (h1 h1, starts at the scaled bias) and cross (h1 h2 + h2 h1)
                                // accumulators; the wave's five row tiles in two batches (3 + 2) to keep the accumulators at 24 registers
                                auto p_batch = [&](auto q0c, auto nqc) {
                                    constexpr int Q0 = decltype(q0c)::value, NQ = decltype(nqc)::value;
                                    f32x4 pm[NQ], px[NQ];
#pragma unroll
                                    for (int i = 0; i < NQ; ++i) {
                                        const f32x4 bsv = *reinterpret_cast<const f32x4*>(CBO + ((rw * COOP_NRT + Q0 + i) * 4 + lk) * 4);
#pragma unroll
                                        for (int r = 0; r < 4; ++r) pm[i][r] = bsv[r] * sxw;
                                        px[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                                    }
                                    u32x4 b0_ = *reinterpret_cast<const u32x4*>(bxs + lane * 4), b1_ = *reinterpret_cast<const u32x4*>(bxs + (64 + lane) * 4);
#pragma unroll
                                    for (int c = 0; c < COOP_NCH; ++c) {
                                        const u32x4 c0_ = b0_, c1_ = b1_;
                                        if (c + 1 < COOP_NCH) {      // the next chunk's operand is read while this chunk multiplies
                                            b0_ = *reinterpret_cast<const u32x4*>(bxs + (((c + 1) * 2 + 0) * 64 + lane) * 4);
                                            b1_ = *reinterpret_cast<const u32x4*>(bxs + (((c + 1) * 2 + 1) * 64 + lane) * 4);
                                        }
#pragma unroll
                                        for (int i = 0; i < NQ; ++i) pm[i] = mfma_h(cw.f[((Q0 + i) * 4 + c) * 2 + 0], c0_, pm[i]);
#pragma unroll
                                        for (int i = 0; i < NQ; ++i) px[i] = mfma_h(cw.f[((Q0 + i) * 4 + c) * 2 + 1], c0_, px[i]);
#pragma unroll
                                        for (int i = 0; i < NQ; ++i) px[i] = mfma_h(cw.f[((Q0 + i) * 4 + c) * 2 + 0], c1_, px[i]);
                                    }
#pragma unroll
                                    for (int i = 0; i < NQ; ++i) {
                                        const f32x4 pc4 = h2_combine(pm[i], px[i]);
                                        const f32x4 dxv = *reinterpret_cast<const f32x4*>(CDX + buf * 1280 + ((cq0 + Q0 + i) * NSP + li) * 4);
                                        float dps[4];
#pragma unroll
                                        for (int r = 0; r < 4; ++r) {
                                            const float th = tanh_prescaled(pc4[r] * isx2);
                                            kk = fmaf(th, dxv[r], kk);
                                            dps[r] = (asd * dxv[r]) * (1.0f - th * th);
                                        }
                                        // this tile's half of its pair's B operand: the lane's own four values (k = 8 g + 4 (tile & 1) + r)
                                        const int rtl = rw * COOP_NRT + Q0 + i, pr = rtl >> 1, half = rtl & 1;
                                        unsigned h0, l0, h1, l1;
                                        coop_split2(dps[0], dps[1], h0, l0);
                                        coop_split2(dps[2], dps[3], h1, l1);
                                        *reinterpret_cast<u32x2c*>(dpx + ((pr * 2 + 0) * 64 + lane) * 4 + 2 * half) = (u32x2c){h0, h1};
                                        *reinterpret_cast<u32x2c*>(dpx + ((pr * 2 + 1) * 64 + lane) * 4 + 2 * half) = (u32x2c){l0, l1};
                                    }
                                };
                                p_batch(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});
                                p_batch(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{});
                                CKX[(buf * 4 + rw) * 64 + lane] = kk;
                            }
#ifdef NCDE_TL_PROF
                            { const unsigned long long n_ = __builtin_readcyclecounter(); kprof[0] += n_ - klast; klast = n_; }
#endif
                            if (it + 1 < cd.M) stage_store(buf ^ 1, tid);      // (requested at the top of the iteration)
#ifdef NCDE_TL_PROF
                            { const unsigned long long n_ = __builtin_readcyclecounter(); kprof[1] += n_ - klast; klast = n_; }
#endif
                            __syncthreads();
#ifdef NCDE_TL_PROF
                            { const unsigned long long n_ = __builtin_readcyclecounter(); kprof[2] += n_ - klast; klast = n_; }
#endif
                        }
                    } else {
                        for (int it = 0; it <= cd.M; ++it) {
                            // (thread-id derived offsets are re-derived from an opaque copy every iteration: hoisted out of the loop they were
                            // spilled, and every scratch reload waits -- vmcnt is in order -- for the staging loads of the NEXT tile too)
                            int tid = tid_outer;
                            asm volatile("" : "+v"(tid));
                            const int lane = tid & 63, li = tid & 15, lk = (tid >> 4) & 3;
                            const int buf = it & 1;
                            if (it + 1 < cd.M) stage_load(c_grp + cd.G * (it + 1), buf ^ 1, tid);
                            if (it >= 1) {
                                // T role, tile it - 1: this wave's two column tiles of the 128 hidden units over the 10 row-tile pairs
                                const int pb = buf ^ 1, tt = c_grp + cd.G * (it - 1);
                                const long long tb = (long long)tt * cd.per_tile();
                                const unsigned* dpx = reinterpret_cast<const unsigned*>(CDP) + pb * (10 * 2 * 64 * 4);
                                const float isd = CIS[pb * 16 + li];
                                f32x4 tm[2], tx[2];
#pragma unroll
                                for (int ci = 0; ci < 2; ++ci) tm[ci] = tx[ci] = (f32x4){0.f, 0.f, 0.f, 0.f};
                                u32x4 nh = *reinterpret_cast<const u32x4*>(dpx + lane * 4), nl = *reinterpret_cast<const u32x4*>(dpx + (64 + lane) * 4);
#pragma unroll
                                for (int pr = 0; pr < COOP_RPM / 2; ++pr) {
                                    const u32x4 bh = nh, bl = nl;
                                    if (pr + 1 < COOP_RPM / 2) {      // the next pair's operand is read while this pair multiplies
                                        nh = *reinterpret_cast<const u32x4*>(dpx + (((pr + 1) * 2 + 0) * 64 + lane) * 4);
                                        nl = *reinterpret_cast<const u32x4*>(dpx + (((pr + 1) * 2 + 1) * 64 + lane) * 4);
                                    }
#pragma unroll
                                    for (int ci = 0; ci < 2; ++ci) tm[ci] = mfma_h(cw.f[(pr * 2 + ci) * 2 + 0], bh, tm[ci]);
#pragma unroll
                                    for (int ci = 0; ci < 2; ++ci) tx[ci] = mfma_h(cw.f[(pr * 2 + ci) * 2 + 1], bh, tx[ci]);
#pragma unroll
                                    for (int ci = 0; ci < 2; ++ci) tx[ci] = mfma_h(cw.f[(pr * 2 + ci) * 2 + 0], bl, tx[ci]);
                                }
#pragma unroll
                                for (int ci = 0; ci < 2; ++ci) {
                                    f32x4 o = h2_combine(tm[ci], tx[ci]);
#pragma unroll
                                    for (int r = 0; r < 4; ++r) o[r] *= isd;
                                    // units 16 ct + 4 lk + r of sample li, ct = 2 rw + ci, in the activation layout [unit / 4][16][4]
                                    coop_st16(crs, a.coop_x, c_same, tb + cd.off_part() + (long long)c_mem * (dlast * NSP) + ((4 * (2 * rw + ci) + lk) * NSP + li) * 4,
                                              __builtin_bit_cast(u32x4, o));
                                }
                                if (tid - 256 < 64 * jh) {      // f.dX of this member's state units: sum of the P waves that hold their channel tiles
                                    const int hl = (tid - 256) >> 6, ln = tid & 63;
                                    float ksum = 0.0f;
                                    for (int wv = hl * wph; wv < (hl + 1) * wph; ++wv) ksum += CKX[(pb * 4 + wv) * 64 + ln];
                                    coop_st4(crs, a.coop_x, c_same, tb + cd.off_ko() + ((c_mem * jh + hl) * NSP + (ln & 15)) * 4 + (ln >> 4), ksum);
                                }
                            }
#ifdef NCDE_TL_PROF
                            { const unsigned long long n_ = __builtin_readcyclecounter(); kprof[0] += n_ - klast; klast = n_; }
#endif
                            if (it + 1 < cd.M) stage_store(buf ^ 1, tid);      // (requested at the top of the iteration)
#ifdef NCDE_TL_PROF
                            { const unsigned long long n_ = __builtin_readcyclecounter(); kprof[1] += n_ - klast; klast = n_; }
#endif
                            __syncthreads();
#ifdef NCDE_TL_PROF
                            { const unsigned long long n_ = __builtin_readcyclecounter(); kprof[2] += n_ - klast; klast = n_; }
#endif
                        }
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (tid == 0) coop_arrive(csy, c_grp);
                    TL_TICK(10)
#pragma unroll
                    for (int q = 0; q < TL_EADJ; ++q) {
                        if (q * NT < HS) {
                            ky1[q] = KOY[q * NT + tid]; ky2[q] = KOA[q * NT + tid];
                            ka1[q] = G0[q * NT + tid]; ka2[q] = G1[q * NT + tid];
                            y0[q] = AS[q * NT + tid]; a0[q] = reinterpret_cast<const float*>(XBA)[q * NT + tid];
                        }
                    }
                    {   // next stage's inputs (requested here, not at the top of the stage: the keeper phase needs the registers)
                        const int jn = j + 1 < S ? j + 1 : 0, nn = j + 1 < S ? n : n - 1;
                        if (nn > a.win_lo) prefetch(nn, jn);
                    }
                } else
                tl_output_vjp<PK, NWV, RES, GATED, (BF != 0 && GATED == 0) ? 1 : 0>(a, in, AS, DX, KOY, scr, wave, lane, wo, XBA, G1);
                TL_TICK(1)
                // ---- records for pass B (x_L twice, weighted cotangent, dX/dt) --------------------------------------------
                if constexpr (COOP == 0) write_records();      // (COOP: written while the group assembles, see above)
                TL_SYNC(2)
                // ---- dL/dpre_L = (sum of the 8 partials) * relu'(x_L) ---------------------------------------------------------
                if constexpr (COOP != 0) {      // ... of the group's M partials of this tile, in member order; f.dX from its keepers' slices
                    const unsigned bar2 = (2u * (unsigned)sc + 2u) * (unsigned)cd.M;
                    if (!coop_wait(csy, c_grp, bar2, CFL, tid)) {
                        for (int e = tid; e < NSP * a.Hr; e += NT)
                            if (b0 + e / a.Hr < a.B) a.grad_z0[(long long)b0 * a.Hr + e] = __builtin_nanf("");
                        return;
                    }
                    TL_TICK(11)
                    const long long my_base = (long long)blockIdx.x * cd.per_tile();
                    for (int e4 = tid * 4; e4 < dlast * NSP; e4 += NT * 4) {
                        f32x4 g = (f32x4){0.f, 0.f, 0.f, 0.f};
                        for (int mm = 0; mm < cd.M; mm += 16) {      // sixteen partials in flight (four were eight dependent L2 round trips at M = 32)
                            f32x4 p[16];
#pragma unroll
                            for (int k = 0; k < 16; ++k)
                                p[k] = mm + k < cd.M ? coop_ld16f(crs, my_base + cd.off_part() + (long long)(mm + k) * (dlast * NSP) + e4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int k = 0; k < 16; ++k)      // (member order, as before: the zeros past M change nothing)
#pragma unroll
                                for (int r = 0; r < 4; ++r) g[r] += p[k][r];
                        }
                        const f32x4 xv = *reinterpret_cast<const f32x4*>(in + e4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) g[r] = xv[r] > 0.0f ? g[r] : 0.0f;
                        *reinterpret_cast<f32x4*>(G1 + e4) = g;
                    }
                    for (int e4 = tid * 4; e4 < HS; e4 += NT * 4) *reinterpret_cast<f32x4*>(KOY + e4) = coop_ld16f(crs, my_base + cd.off_ko() + e4);
                } else if constexpr (!W16)      // (W16: tl_output_vjp left the masked sum in G1)
                for (int e = tid; e < dlast * NSP; e += NT) {
                    float g = 0.0f;
    #pragma unroll
                    for (int wv = 0; wv < NWV; ++wv) g += SC[wv * SCW + e];
                    G1[e] = in[e] > 0.0f ? g : 0.0f;
                }
                TL_SYNC(3)
            }
            // ---- hidden layers backwards ----------------------------------------------------------------------------------
            float* gpre = G1;
            float* gx = G0;
            for (int l = L - 1; l >= 0; --l) {
                const int N = a.dout[l], K = a.din[l];
                const float* xin = l == 0 ? YS : X + (l - 1) * DS;
                if (w != 0.0f) {
                    const bool slot0 = a.gW_off[l] == a.gW_off[0];
                    if constexpr (DWG) tl_dw_rmw<NWV>(gpre, xin, N, K, w, a.gpart + (long long)blockIdx.x * a.gstride + a.gW_off[l], wave, lane);
                    else if (slot0) tl_dw_acc<NWV, TL_DWT>(gpre, xin, N, K, w, dw0, wave, lane);
                    else tl_dw_acc<NWV, TL_DWT>(gpre, xin, N, K, w, dw1, wave, lane);
                    if (tid < N) {
                        float sum = 0.0f;
#pragma unroll
                        for (int s = 0; s < NSP; ++s) sum += gpre[((tid >> 2) * NSP + s) * 4 + (tid & 3)];
                        if (slot0) db0 += w * sum;
                        else db1 += w * sum;
                    }
                }
                if constexpr (RES != 0) {
                    if (wave < PK) {
                        const int li = lane & 15, lk = lane >> 4;
                        const bool s0 = a.gW_off[l] == a.gW_off[0];
                        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kb = 0; kb < PK; ++kb) {
                            const f32x4 Bv = *reinterpret_cast<const f32x4*>(gpre + ((4 * kb + lk) * NSP + li) * 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc = mfma16(s0 ? wb[0][4 * kb + e] : wb[1][4 * kb + e], Bv[e], acc);
                        }
                        const int o = ((4 * wave + lk) * NSP + li) * 4;
                        if (l > 0) {
                            const f32x4 xv = *reinterpret_cast<const f32x4*>(xin + o);
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[r] = xv[r] > 0.0f ? acc[r] : 0.0f;
                        }
                        *reinterpret_cast<f32x4*>((l == 0 ? KOA : gx) + o) = acc;
                    }
                } else {
                    const TlW wr_ = tl_wref(lds, DIRECT != 0 ? a.tres[l] : 0, a.W[l], a.b[l], N, K);
                    tl_hidden_bwd<NWV>(wr_.W, N, K, gpre, xin, l > 0, l == 0 ? KOA : gx, wave, lane, nullptr, nullptr, wr_.ld);
                }
                TL_SYNC(4)
                float* tmp = gpre; gpre = gx; gx = tmp;
            }
            // ---- Butcher bookkeeping (registers) ---------------------------------------------------------------------------
#pragma unroll
            for (int q = 0; q < TL_EADJ; ++q) {
                const int e = tid + q * NT;
                if (e < HS) {
                    const int u = ((e >> 2) / NSP) * 4 + (e & 3), s = (e >> 2) % NSP, b = b0 + s;
                    const bool valid = b < a.B && u < a.Hr;      // (zero-padded problems: units >= Hr are not the caller's)
                    const int H = a.Hr;                          // row width of the caller's z_out / grad_out / grad_z0
                    const float d = KOA[e];
                    if (disc) {
                        // transpose of the Butcher step: d = dL/dY of this stage (see ncde_generic.hip / ncde_variant.hip)
                        bool last = false;
                        float next = 0.0f;
                        if (a.method == NCDE_RK4_38) {
                            const float c4 = (a0[q] * dt) * 0.125f;
                            const float dt3 = dt * 0.333333343267440796f;
                            if (j == 0) { ka1[q] = d; next = 3.0f * c4 + dt * d; }
                            else if (j == 1) { ka2[q] = d; next = (3.0f * c4 - dt * ka1[q]) + dt * d; }
                            else if (j == 2) { ky1[q] = d; next = ((c4 + dt * ka1[q]) - dt3 * ka2[q]) + dt3 * d; }
                            else { a0[q] = (((a0[q] + ka1[q]) + ka2[q]) + ky1[q]) + d; last = true; }
                        } else if (a.method == NCDE_MIDPOINT) {
                            if (j == 0) { ka1[q] = d; next = (0.5f * dt) * d; }
                            else { a0[q] = (a0[q] + ka1[q]) + d; last = true; }
                        } else {
                            a0[q] = a0[q] + d; last = true;
                        }
                        if (last) {
                            float dtp = 1.0f;      // dt of the forward step transposed next (n - 1 -> n - 2)
                            if (planned) {
                                if (valid) {
                                    const long long brow = (long long)b * a.n_out;
                                    a0[q] = a0[q] + plan_out_cotangent(a, pstep, pout, 0, brow, u);
                                    if (n > 1) a0[q] = a0[q] + plan_out_cotangent(a, pstep - pw_, pout, 1, brow, u);
                                    else a0[q] = a0[q] + a.grad_out[brow * H + u];
                                }
                                if (n > 1) dtp = __int_as_float(pstep[-pw_]);
                            } else if (a.output == NCDE_OUT_KNOTS || n == 1) {
                                a0[q] += valid ? a.grad_out[((long long)b * a.n_out + (a.output == NCDE_OUT_KNOTS ? n - 1 : 0)) * H + u] : 0.0f;
                            }
                            next = a.method == NCDE_RK4_38 ? (a0[q] * dtp) * 0.125f : dtp * a0[q];
                            if (n == 1 && valid) a.grad_z0[(long long)b * H + u] = a0[q];
                        }
                        AS[e] = next;
                    } else {
                        // negated time: dy/ds = -f, da/ds = +a^T df/dy; same operation order as StageCombine
                        bool last;
                        const float ys0 = StageCombine::apply(a.method, j, -KOY[e], dt, y0[q], ky1[q], ky2[q], last);
                        const float as0 = StageCombine::apply(a.method, j, d, dt, a0[q], ka1[q], ka2[q], last);
                        float ys = ys0, as = as0;
                        if (last) {
                            if (planned) {      // one reverse solve per output interval: reset y to the stored value, add dL/dz there
                                const int row = pstep[1];
                                if (row >= 0) {
                                    const long long o = ((long long)b * a.n_out + row) * H + u;
                                    y0[q] = valid ? a.z_out[o] : 0.0f;
                                    a0[q] = a0[q] + (valid ? a.grad_out[o] : 0.0f);
                                }
                            } else if (a.output == NCDE_OUT_KNOTS) {  // reset y to the stored value, add dL/dz at this knot
                                const long long o = ((long long)b * a.n_out + (n - 1)) * H + u;
                                y0[q] = valid ? a.z_out[o] : 0.0f;
                                a0[q] = a0[q] + (valid ? a.grad_out[o] : 0.0f);
                            } else if (n == 1) {
                                a0[q] = a0[q] + (valid ? a.grad_out[((long long)b * a.n_out) * H + u] : 0.0f);
                            }
                            ys = y0[q];
                            as = a0[q];
                            if (n == 1 && valid) a.grad_z0[(long long)b * H + u] = a0[q];
                        }
                        YS[e] = ys;
                        AS[e] = as;
                    }
                }
            }
            publish();
            TL_SYNC(5)
        }
    }
    if (a.win_lo > 0) {      // hand (y, a) at knot win_lo to the next time window
#pragma unroll
        for (int q = 0; q < TL_EADJ; ++q) {
            const int e = tid + q * NT;
            if (e < HS) {
                a.carry[(long long)blockIdx.x * HS + e] = y0[q];
                a.carry[((long long)gridDim.x + blockIdx.x) * HS + e] = a0[q];
            }
        }
    }
    // ---- this workgroup's partial of the hidden-layer parameter gradients ------------------------------------------
    float* gp = a.gpart + (long long)blockIdx.x * a.gstride;
#ifdef NCDE_TL_PROF
    if (lane == 0 && sc > 0)
        for (int k = 0; k < 12; ++k) a.grad_z0[(long long)b0 * a.Hr + wave * 12 + k] = (float)tprof[k] / (float)sc;      // (over the tile's dz0 rows)
    if (lane == 0 && sc > 0)
        for (int k = 0; k < 3; ++k) a.grad_z0[(long long)b0 * a.Hr + 96 + wave * 3 + k] = (float)kprof[k] / (float)sc;
#endif
    {
        const int li = lane & 15, lk = lane >> 4;
        int l1 = -1;
        for (int l = 1; l < L; ++l)
            if (a.gW_off[l] != a.gW_off[0]) { l1 = l; break; }
#pragma unroll
        for (int slot = 0; slot < 2; ++slot) {
            const int l = slot == 0 ? 0 : l1;
            if (l < 0) continue;
            const int N = a.dout[l], K = a.din[l], nit = K >> 4, ntile = (N >> 4) * nit;
#pragma unroll
            for (int q = 0; q < (DWG ? 0 : TL_DWT); ++q) {
                const int tt = wave + NWV * q;
                if (tt < ntile) {
                    const int jt = tt / nit, it = tt - jt * nit;
                    const f32x4 v = slot == 0 ? dw0[q] : dw1[q];
                    // DIRECT: layer 0 was processed with its columns padded to K; the gradient goes back with the real row stride
                    const int Kreal = (DIRECT != 0 && l == 0) ? a.d0 : K;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (16 * it + li < Kreal) gp[a.gW_off[l] + (16 * jt + 4 * lk + r) * Kreal + 16 * it + li] = v[r];
                }
            }
            if (tid < N) gp[a.gb_off[l] + tid] = slot == 0 ? db0 : db1;
        }
        if constexpr (DIRECT != 0) {      // the heads' gradients: [H][dlast] each
            const int nit = dlast >> 4, ntile = (H >> 4) * nit;
#pragma unroll
            for (int q = 0; q < TL_DWT; ++q) {
                const int tt = wave + NWV * q;
                if (tt < ntile) {
                    const int jt = tt / nit, it = tt - jt * nit;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        gp[a.gWo_off + (16 * jt + 4 * lk + r) * dlast + 16 * it + li] = dwo[q][r];
                        if constexpr (GATED != 0) gp[a.gWg_off + (16 * jt + 4 * lk + r) * dlast + 16 * it + li] = dwg[q][r];
                    }
                }
            }
            if (tid < H) {
                gp[a.gbo_off + tid] = dbo;
                if constexpr (GATED != 0) gp[a.gbg_off + tid] = dbg;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// pass B: dL/dWo, dL/dbo from the records -- weights and gradient accumulators never leave the registers
// ------------------------------------------------------------------------------------------------
// grid = (H*C/16 row tiles, parts); one WAVE = one 16-row tile of Wo x one 1/(4*parts) slice of the sample
// tiles, for ALL stages: P = Wo_tile x_L + bo is recomputed (PK*4 MFMAs), dP = cot (x) dX (1 - tanh^2 P), and
// dWo_tile += dP x_L^T (PK*4 MFMAs, samples are the K dim; dP transposed through a 16x17 LDS patch).
// HEAD 0: the original field (tanh head).  Minimal-gated field: HEAD 1 = gradient of the tanh head (Wo, bo), HEAD 2 = of the
// sigmoid head (Wg, bg); either run recomputes both pre-activations (M = sigmoid(Pg) * tanh(Pt)), accumulates one.
// NRT = row tiles per wave: each fragment of records (16 KB at cfg5, re-read by every row tile) then feeds NRT tiles --
// measured at cfg5 with one tile per wave the pass moved 13.7 TB/s through the L2 (3.9 M fragments x 16 KB per 4.7 ms
// window) and sat at that, not at the MFMA rate.  One wave per SIMD, so the 512-register file holds NRT = 4 tiles.
// (fp32 records; the sweep's split-record mode is consumed by ncde_dwo_pair below)
template <int PK, int HEAD = 0, int NRT = 1>
__global__ __launch_bounds__(256) void ncde_dwo_tiled(KArgs a, int n_sc, int n_st, float* gpartB) {
    // n_sc = stages recorded in this time window; a.resume != 0: add to the partial the earlier windows left in gpartB
    // (the waves' accumulators are summed PKR column tiles at a time: 16 x 256 floats x 4 waves would not fit the 64 KB of static LDS)
    constexpr int PKR = PK > 8 ? 8 : PK;
    if (a.run_if != nullptr && *a.run_if == 0u) return;      // (re-execution behind a cooperative sequence: see KArgs.run_if)
    __shared__ float patch[4][NRT][16 * 17];
    __shared__ __attribute__((aligned(16))) float red[4][PKR * 256 + 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int C = a.C, H = a.H, ncq = C >> 2, dlast = 16 * PK;
    const int part = blockIdx.y * 4 + wave, nparts = gridDim.y * 4;
    int hb[NRT], cq[NRT];
    Panel<PK> Wp[NRT];
    f32x4 bv[NRT];
    Panel<PK> Wq[HEAD != 0 ? NRT : 1];        // sigmoid head (gated field)
    f32x4 bq[HEAD != 0 ? NRT : 1];
    f32x4 gW[NRT][PK];
    float gb[NRT][4];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
        const int tile = blockIdx.x * NRT + rt;
        hb[rt] = tile / ncq;
        cq[rt] = tile - hb[rt] * ncq;
        const long long woff = (long long)((4 * hb[rt] + (li >> 2)) * C + 4 * cq[rt] + (li & 3)) * dlast + 4 * lk;
        bv[rt] = *reinterpret_cast<const f32x4*>(a.bo + (4 * hb[rt] + lk) * C + 4 * cq[rt]);
        Wp[rt] = tl_load_panel<PK>(a.Wo + woff, 0);
        if constexpr (HEAD != 0) {
            bq[rt] = *reinterpret_cast<const f32x4*>(a.bg + (4 * hb[rt] + lk) * C + 4 * cq[rt]);
            Wq[rt] = tl_load_panel<PK>(a.Wg + woff, 0);
        }
#pragma unroll
        for (int jt = 0; jt < PK; ++jt) gW[rt][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) gb[rt][r] = 0.0f;
    }
    const int my_n = part < n_st ? (n_st - part + nparts - 1) / nparts : 0;   // sample tiles part, part + nparts, ...
    struct Frag {
        f32x4 xa[PK], xb[PK], dx[NRT];
        float cot[NRT];
    };
    // fragment of (stage sc, k-th sample tile of this wave); past the end: the last one again with a zero cotangent
    auto load_frag = [&](int sc, int k, bool live) {
        const long long t = (long long)sc * n_st + (part + nparts * k);
        Frag f;
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
            f.cot[rt] = a.recC[t * (H * 16) + (4 * hb[rt] + lk) * 16 + li];
            if (!live) f.cot[rt] = 0.0f;
            f.dx[rt] = *reinterpret_cast<const f32x4*>(a.recD + t * (C * 16) + (cq[rt] * 16 + li) * 4);
        }
        const float* ra = a.recA + t * (dlast * 16) + (lk * 16 + li) * 4;
        const float* rb = a.recB + t * (dlast * 16) + li * 16 + 4 * lk;
#pragma unroll
        for (int i = 0; i < PK; ++i) {
            f.xa[i] = *reinterpret_cast<const f32x4*>(ra + i * 256);
            f.xb[i] = *reinterpret_cast<const f32x4*>(rb + i * 256);
        }
        return f;
    };
    auto step = [&](const Frag& f) {
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
            float* pt = patch[wave][rt];
            f32x4 acc = bv[rt];
#pragma unroll
            for (int i = 0; i < PK; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = mfma16(Wp[rt].v[i][e], f.xa[i][e], acc);
            f32x4 accq;
            if constexpr (HEAD != 0) {
                accq = bq[rt];
#pragma unroll
                for (int i = 0; i < PK; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) accq = mfma16(Wq[rt].v[i][e], f.xa[i][e], accq);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float m = tanh_dev(acc[r]);
                const float dm = f.cot[rt] * f.dx[rt][r];
                float dp;
                if constexpr (HEAD == 0) {
                    dp = dm * (1.0f - m * m);
                } else {
                    const float sg = tl_sigmoid(accq[r]);
                    dp = HEAD == 1 ? (dm * sg) * (1.0f - m * m) : (dm * m) * (sg * (1.0f - sg));
                }
                gb[rt][r] += dp;
                pt[(4 * lk + r) * 17 + li] = dp;
            }
            float av[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) av[e] = pt[li * 17 + 4 * lk + e];
#pragma unroll
            for (int jt = 0; jt < PK; ++jt)
#pragma unroll
                for (int e = 0; e < 4; ++e) gW[rt][jt] = mfma16(av[e], f.xb[jt][e], gW[rt][jt]);
        }
    };
    // two named fragment buffers, loop unrolled by two: the loads of one buffer are in flight while the other computes
    if (my_n > 0 && n_sc > 0) {
        int sc = 0, k = 0;          // position of the NEXT fragment to fetch
        bool live = true;
        auto fetch = [&]() {
            const Frag f = load_frag(sc, k, live);
            if (live) {
                if (++k == my_n) { k = 0; ++sc; }
                if (sc == n_sc) { live = false; sc = n_sc - 1; k = my_n - 1; }
            }
            return f;
        };
        const long long nq = (long long)n_sc * my_n;
        Frag fA = fetch(), fB;
        for (long long q = 0; q < nq; q += 2) {
            fB = fetch();
            step(fA);
            fA = fetch();
            step(fB);
        }
    }
    // ---- sum the four waves of the workgroup, write this part-group's partial ----------------------------------------
    const long long wo_sz = (long long)H * C * dlast, theta_o = wo_sz + (long long)H * C;
    float* gp = gpartB + (long long)blockIdx.y * theta_o;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
#pragma unroll
        for (int j0 = 0; j0 < PK; j0 += PKR) {
            if (rt > 0 || j0 > 0) __syncthreads();
#pragma unroll
            for (int jt = 0; jt < PKR; ++jt) *reinterpret_cast<f32x4*>(&red[wave][(jt * 64 + lane) * 4]) = gW[rt][j0 + jt];
            if (j0 == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = gb[rt][r];
                    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
                    if (li == 0) red[wave][PKR * 256 + 4 * lk + r] = v;
                }
            }
            __syncthreads();
            for (int e = tid; e < PKR * 256; e += 256) {
                const float v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
                const int r = e & 3, ln = (e >> 2) & 63, jt = j0 + (e >> 8);
                const int row = (4 * hb[rt] + (ln >> 4)) * C + 4 * cq[rt] + r;
                float* dst = gp + (long long)row * dlast + 16 * jt + (ln & 15);
                *dst = a.resume ? *dst + v : v;
            }
            if (j0 == 0 && tid < 16) {
                const float v = (red[0][PKR * 256 + tid] + red[1][PKR * 256 + tid]) + (red[2][PKR * 256 + tid] + red[3][PKR * 256 + tid]);
                float* dst = gp + wo_sz + (4 * hb[rt] + (tid >> 2)) * C + 4 * cq[rt] + (tid & 3);
                *dst = a.resume ? *dst + v : v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// pass B on the bf16 matrix cores: sample tiles in PAIRS
// ------------------------------------------------------------------------------------------------
// Same job and same grid as ncde_dwo_tiled, for the split records the sweep writes in its BF mode.  One step = one PAIR of
// sample tiles (2i, 2i+1) of one stage:
//   P_a, P_b = Wo_tile x_L + bo        6 PK/2 bf16 MFMAs each (weights split once at the start, x_L split in record A)
//   dP       = cot (x) dX (1 - tanh^2) per tile, transposed through a 16x17 LDS patch, split on the fly
//   dWo_tile += [dP_a dP_b] [x_L,a x_L,b]^T   the 32 samples of the pair are the K dim: 6 bf16 MFMAs per 16 columns
// against 4 PK + 4 PK fp32-input MFMAs of twice the cycles per tile.  Every record piece has ONE register buffer that is
// re-requested for the next pair right after its last use (one wave per SIMD: the file is the limit, not occupancy).
template <int PK, int HEAD = 0, int NRT = 1>
__global__ __launch_bounds__(256) void ncde_dwo_pair(KArgs a, int n_sc, int n_st, float* gpartB) {
    static_assert(PK >= 2, "split records need a last hidden width that is a multiple of 32");
    constexpr int NCH = PK / 2;
    constexpr int NH = HEAD == 3 ? 2 : 1;      // HEAD 3 (minimal-gated field): both heads' gradients from ONE recompute of P
    if (a.run_if != nullptr && *a.run_if == 0u) return;      // (re-execution behind a cooperative sequence: see KArgs.run_if)
    __shared__ float patch[4][NRT][2][NH][16 * 17];
    __shared__ __attribute__((aligned(16))) float red[4][PK * 256 + 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int C = a.C, H = a.H, ncq = C >> 2, dlast = 16 * PK;
    const int part = blockIdx.y * 4 + wave, nparts = gridDim.y * 4;
    const int n_pair = (n_st + 1) >> 1;
    int hb[NRT], cq[NRT];
    f32x4 bv[NRT], bq[HEAD != 0 ? NRT : 1];
    u32x4 Ws[NRT][NCH][3], Wqs[HEAD != 0 ? NRT : 1][NCH][3];
    f32x4 gW[NH][NRT][PK];
    float gb[NH][NRT][4];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
        const int tile = blockIdx.x * NRT + rt;
        hb[rt] = tile / ncq;
        cq[rt] = tile - hb[rt] * ncq;
        const long long wrow = (long long)((4 * hb[rt] + (li >> 2)) * C + 4 * cq[rt] + (li & 3)) * dlast;
        bv[rt] = *reinterpret_cast<const f32x4*>(a.bo + (4 * hb[rt] + lk) * C + 4 * cq[rt]);
        if constexpr (HEAD != 0) bq[rt] = *reinterpret_cast<const f32x4*>(a.bg + (4 * hb[rt] + lk) * C + 4 * cq[rt]);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {      // lane (row li, k-group lk) of chunk c: k = 32 c + 8 lk + 0..7
            float v[8];
            *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(a.Wo + wrow + 32 * c + 8 * lk);
            *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(a.Wo + wrow + 32 * c + 8 * lk + 4);
            const Split3 sp = split8(v);
            Ws[rt][c][0] = sp.hi; Ws[rt][c][1] = sp.mid; Ws[rt][c][2] = sp.lo;
            if constexpr (HEAD != 0) {
                *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(a.Wg + wrow + 32 * c + 8 * lk);
                *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(a.Wg + wrow + 32 * c + 8 * lk + 4);
                const Split3 sq = split8(v);
                Wqs[rt][c][0] = sq.hi; Wqs[rt][c][1] = sq.mid; Wqs[rt][c][2] = sq.lo;
            }
        }
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
#pragma unroll
            for (int jt = 0; jt < PK; ++jt) gW[hh][rt][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) gb[hh][rt][r] = 0.0f;
        }
    }
    const int my_n = part < n_pair ? (n_pair - part + nparts - 1) / nparts : 0;   // pairs part, part + nparts, ...
    // record pieces of the pair being processed / requested
    u32x4 xs[2][NCH][3];     // x_L of tile a / b, B operand of P
    u32x4 xp[PK][3];         // x_L^T of the pair, B operand of dWo
    f32x4 dxv[2][NRT];
    float cot[2][NRT];
    int sc = 0, k = 0;       // position of the pair to request next
    auto tiles_of = [&](int kk, int& ta, int& tb, bool& has_b) {
        const int pr = part + nparts * kk;
        ta = 2 * pr;
        has_b = ta + 1 < n_st;
        tb = has_b ? ta + 1 : ta;       // odd tile count: tile a again, with a zero cotangent (its record-B half is zeros)
    };
    auto load_xs = [&](int which) {
        int ta, tb; bool hb_;
        tiles_of(k, ta, tb, hb_);
        const long long t = (long long)sc * n_st + (which == 0 ? ta : tb);
        const unsigned* ra = reinterpret_cast<const unsigned*>(a.recA + t * (dlast * 24)) + lane * 4;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) xs[which][c][pc] = *reinterpret_cast<const u32x4*>(ra + (c * 3 + pc) * 256);
    };
    auto load_meta = [&]() {
        int ta, tb; bool has_b;
        tiles_of(k, ta, tb, has_b);
#pragma unroll
        for (int w2 = 0; w2 < 2; ++w2) {
            const long long t = (long long)sc * n_st + (w2 == 0 ? ta : tb);
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                const float cv = a.recC[t * (H * 16) + (4 * hb[rt] + lk) * 16 + li];
                cot[w2][rt] = (w2 == 1 && !has_b) ? 0.0f : cv;
                dxv[w2][rt] = *reinterpret_cast<const f32x4*>(a.recD + t * (C * 16) + (cq[rt] * 16 + li) * 4);
            }
        }
    };
    auto load_xp = [&]() {
        const int pr = part + nparts * k;
        const unsigned* rb = reinterpret_cast<const unsigned*>(a.recB) + ((long long)sc * n_pair + pr) * (dlast * 48) + lane * 4;
#pragma unroll
        for (int jt = 0; jt < PK; ++jt)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) xp[jt][pc] = *reinterpret_cast<const u32x4*>(rb + (jt * 3 + pc) * 256);
    };
    if (my_n > 0 && n_sc > 0) {
        const long long nq = (long long)n_sc * my_n;
        load_xs(0); load_xs(1); load_meta(); load_xp();
        for (long long q = 0; q < nq; ++q) {
            // position of the NEXT pair (the last iteration re-requests its own: harmless)
            if (q + 1 < nq) { if (++k == my_n) { k = 0; ++sc; } }
            // One wave per SIMD: nothing hides the latency of an MFMA that waits for the previous one's accumulator, so the six
            // partial products of a split product (kPA x kPB, the order of mfma_split) form the OUTER loop and consecutive MFMAs
            // go to different accumulators -- here 2 tiles x NRT row tiles x 2 chunk parities (x 2 heads) chains.  Every
            // accumulator still receives its products in the same order as before.
            constexpr int kPA[6] = {2, 1, 0, 1, 0, 0}, kPB[6] = {0, 0, 0, 1, 1, 2};      // piece index: 0 hi, 1 mid, 2 lo
            f32x4 acc[2][NRT], accq[2][HEAD != 0 ? NRT : 1];
            {
                f32x4 c0[2][NRT], c1[2][NRT], q0[2][HEAD != 0 ? NRT : 1], q1[2][HEAD != 0 ? NRT : 1];
#pragma unroll
                for (int w2 = 0; w2 < 2; ++w2)
#pragma unroll
                    for (int rt = 0; rt < NRT; ++rt) {
                        c0[w2][rt] = bv[rt];
                        c1[w2][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if constexpr (HEAD != 0) { q0[w2][rt] = bq[rt]; q1[w2][rt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
                    }
#pragma unroll
                for (int c2 = 0; c2 < NCH; c2 += 2)
#pragma unroll
                    for (int pp = 0; pp < 6; ++pp)
#pragma unroll
                        for (int w2 = 0; w2 < 2; ++w2)
#pragma unroll
                            for (int rt = 0; rt < NRT; ++rt) {
                                c0[w2][rt] = mfma_bf(Ws[rt][c2][kPA[pp]], xs[w2][c2][kPB[pp]], c0[w2][rt]);
                                if constexpr (HEAD != 0) q0[w2][rt] = mfma_bf(Wqs[rt][c2][kPA[pp]], xs[w2][c2][kPB[pp]], q0[w2][rt]);
                                if constexpr (NCH > 1) {
                                    c1[w2][rt] = mfma_bf(Ws[rt][c2 + 1][kPA[pp]], xs[w2][c2 + 1][kPB[pp]], c1[w2][rt]);
                                    if constexpr (HEAD != 0) q1[w2][rt] = mfma_bf(Wqs[rt][c2 + 1][kPA[pp]], xs[w2][c2 + 1][kPB[pp]], q1[w2][rt]);
                                }
                            }
#pragma unroll
                for (int w2 = 0; w2 < 2; ++w2)
#pragma unroll
                    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            acc[w2][rt][r] = NCH > 1 ? c0[w2][rt][r] + c1[w2][rt][r] : c0[w2][rt][r];
                            if constexpr (HEAD != 0) accq[w2][rt][r] = NCH > 1 ? q0[w2][rt][r] + q1[w2][rt][r] : q0[w2][rt][r];
                        }
            }
            __builtin_amdgcn_sched_barrier(0);
            load_xs(0);            // both buffers are free: request the next pair's
            load_xs(1);
            __builtin_amdgcn_sched_barrier(0);
            // dP of both tiles -> patches (row u = 4 lk + r, column = sample li)
#pragma unroll
            for (int w2 = 0; w2 < 2; ++w2)
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float m = tanh_dev(acc[w2][rt][r]);
                        const float dm = cot[w2][rt] * dxv[w2][rt][r];
                        float dp[NH];
                        if constexpr (HEAD == 0) {
                            dp[0] = dm * (1.0f - m * m);
                        } else {
                            const float sg = tl_sigmoid(accq[w2][rt][r]);
                            const float dt = (dm * sg) * (1.0f - m * m), dg = (dm * m) * (sg * (1.0f - sg));
                            if constexpr (HEAD == 3) { dp[0] = dt; dp[1] = dg; }
                            else dp[0] = HEAD == 1 ? dt : dg;
                        }
#pragma unroll
                        for (int hh = 0; hh < NH; ++hh) {
                            gb[hh][rt][r] += dp[hh];
                            patch[wave][rt][w2][hh][(4 * lk + r) * 17 + li] = dp[hh];
                        }
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
            load_meta();
            __builtin_amdgcn_sched_barrier(0);
            // A operand of dWo: lane (row u = li, k-group lk) holds samples 4 lk .. 4 lk + 3 of tile a, then of tile b
            u32x4 Ap[NH][NRT][3];
#pragma unroll
            for (int hh = 0; hh < NH; ++hh)
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = patch[wave][rt][0][hh][li * 17 + 4 * lk + e];
                        v[4 + e] = patch[wave][rt][1][hh][li * 17 + 4 * lk + e];
                    }
                    const Split3 A = split8(v);
                    Ap[hh][rt][0] = A.hi; Ap[hh][rt][1] = A.mid; Ap[hh][rt][2] = A.lo;
                }
#pragma unroll
            for (int pp = 0; pp < 6; ++pp)           // products outermost: NH x NRT x PK independent accumulators in a row
#pragma unroll
                for (int jt = 0; jt < PK; ++jt)
#pragma unroll
                    for (int hh = 0; hh < NH; ++hh)
#pragma unroll
                        for (int rt = 0; rt < NRT; ++rt) gW[hh][rt][jt] = mfma_bf(Ap[hh][rt][kPA[pp]], xp[jt][kPB[pp]], gW[hh][rt][jt]);
            __builtin_amdgcn_sched_barrier(0);
            load_xp();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- sum the four waves of the workgroup, write this part-group's partial ----------------------------------------
    const long long wo_sz = (long long)H * C * dlast, theta_o = wo_sz + (long long)H * C;
#pragma unroll
    for (int hr = 0; hr < NH * NRT; ++hr) {
        const int hh = hr / NRT, rt = hr - hh * NRT;
        // HEAD 3: the gate head's partials follow the tanh head's gridDim.y partials (the host's gB2 = gB + parts * theta_o)
        float* gp = gpartB + ((long long)hh * gridDim.y + blockIdx.y) * theta_o;
        if (hr > 0) __syncthreads();
#pragma unroll
        for (int jt = 0; jt < PK; ++jt) *reinterpret_cast<f32x4*>(&red[wave][(jt * 64 + lane) * 4]) = gW[hh][rt][jt];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = gb[hh][rt][r];
            v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
            if (li == 0) red[wave][PK * 256 + 4 * lk + r] = v;
        }
        __syncthreads();
        for (int e = tid; e < PK * 256; e += 256) {
            const float v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
            const int r = e & 3, ln = (e >> 2) & 63, jt = e >> 8;
            const int row = (4 * hb[rt] + (ln >> 4)) * C + 4 * cq[rt] + r;
            float* dst = gp + (long long)row * dlast + 16 * jt + (ln & 15);
            *dst = a.resume ? *dst + v : v;
        }
        if (tid < 16) {
            const float v = (red[0][PK * 256 + tid] + red[1][PK * 256 + tid]) + (red[2][PK * 256 + tid] + red[3][PK * 256 + tid]);
            float* dst = gp + wo_sz + (4 * hb[rt] + (tid >> 2)) * C + 4 * cq[rt] + (tid & 3);
            *dst = a.resume ? *dst + v : v;
        }
    }
}

// ncde_reduce_partials (ncde_generic.hip) for the re-execution sequence: runs only when the cooperative sequence gave up
__global__ __launch_bounds__(256) void ncde_reduce_partials_if(const unsigned* run_if, const float* __restrict__ gpart, int n_part, int theta_size, ReduceSegs segs) {
    if (*run_if == 0u) return;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= theta_size) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int p = 0;
    for (; p + 3 < n_part; p += 4) {      // (the summation order of ncde_reduce_partials)
        s0 += gpart[(long long)p * theta_size + k];
        s1 += gpart[(long long)(p + 1) * theta_size + k];
        s2 += gpart[(long long)(p + 2) * theta_size + k];
        s3 += gpart[(long long)(p + 3) * theta_size + k];
    }
    for (; p < n_part; ++p) s0 += gpart[(long long)p * theta_size + k];
    const float total = (s0 + s1) + (s2 + s3);
    for (int i = 0; i < segs.n; ++i)
        if (k >= segs.off[i] && k < segs.off[i] + segs.len[i]) segs.dst[i][k - segs.off[i]] = total;
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

int tiled_fwd_ns(const NcdeProblem* p);
// Split-bf16 output tiles in the forward: one sample tile per workgroup, last hidden width 32 / 64 / 128, unless the caller
// asks for plain fp32-input MFMA (NCDE_FLAG_FP32_MFMA).
bool tiled_fwd_bf(const NcdeProblem* p) {
    const int dl = p->layer_out[p->n_layers - 1];
    return !(p->flags & NCDE_FLAG_FP32_MFMA) && (dl == 32 || dl == 64 || dl == 128) && tiled_fwd_ns(p) == 1;
}
// which split the forward's output tiles multiply in: 0 none (fp32-input MFMA), 2 = 2-way split-fp16 (default), 1 = 3-way
// split-bf16 (NCDE_FLAG_SPLIT_BF16; also what re-executes range-faulted sample tiles of mode 2)
int tiled_fwd_split(const NcdeProblem* p) { return !tiled_fwd_bf(p) ? 0 : ((p->flags & NCDE_FLAG_SPLIT_BF16) ? 1 : 2); }
int64_t tiled_fault_floats(const NcdeProblem* p) { return (((int64_t)(p->batch + 15) / 16 + 1) + 63) & ~(int64_t)63; }   // one word per tile + the weights' word
// Floats of the fragment-ordered copy of the output layer (and of the gate head): 0 when the last hidden width is not one
// of the whole-panel cases the kernels read packed.  The split-bf16 copy (forward, pass 0) takes 1.5 x.
int64_t tiled_pack_floats(const NcdeProblem* p, bool bf, bool adj = false) {
    const int dl = p->layer_out[p->n_layers - 1];
    if (dl != 16 && dl != 32 && dl != 64 && dl != 128 && !(dl == 256 && adj)) return 0;      // (256: the backward's W16 mode only)
    const int64_t n = (int64_t)p->hidden * p->channels * dl * (p->field_kind == NCDE_FIELD_MINIMAL ? 2 : 1);
    return bf ? n + n / 2 : n;
}
// Direct modes: which matrices get an LDS-resident copy behind the kernel's own `base` bytes of LDS (tl_fill_resident): the inner
// layers (a shared one once) as far as they fit, then the heads (both or neither).  Fills a->tres*, returns the LDS bytes to launch with.
size_t tiled_direct_residency(const NcdeProblem* p, KArgs* a, size_t base) {
    for (int l = 0; l < NCDE_MAX_LAYERS; ++l) a->tres[l] = 0;
    a->tres_o = a->tres_g = 0;
    if (ncde_dev_env("NCDE_TILED_NO_RESIDENT")) return base;
    size_t off = (base + 15) / 16 * 4;      // floats, 16-byte aligned
    auto need = [](int N, int K) { return ((size_t)N * (K + 4) + N + 3) & ~(size_t)3; };
    auto fits = [&](size_t n) { return (off + n) * sizeof(float) <= (size_t)kLdsLimit; };
    for (int l = 0; l < p->n_layers; ++l) {
        int shared = -1;
        for (int q = 0; q < l; ++q)
            if (a->W[q] == a->W[l]) shared = q;
        if (shared >= 0) { a->tres[l] = a->tres[shared]; continue; }
        const size_t n = need(a->dout[l], a->din[l]);
        if (!fits(n)) continue;
        a->tres[l] = (int)off;
        off += n;
    }
    const int dl = p->layer_out[p->n_layers - 1];
    const bool gated = p->field_kind != NCDE_FIELD_ORIGINAL;
    const size_t nh = need(p->hidden, dl);
    if (fits(nh * (gated ? 2 : 1))) {
        a->tres_o = (int)off; off += nh;
        if (gated) { a->tres_g = (int)off; off += nh; }
    }
    return off * sizeof(float);
}

// Packs Wo (and Wg) into `dst` and hands the copies to the kernels.
// mode: 0 = fp32 fragments, 1 = split-bf16 (bf = true), 2 = split-fp16 (same size as the fp32 copy; wfault = the weights' range-fault word)
void tiled_pack_launch(const NcdeProblem* p, KArgs* a, float* dst, bool bf, hipStream_t st, int mode = -1, int* wfault = nullptr) {
    const int dl = p->layer_out[p->n_layers - 1];
    const long long n4 = (long long)p->hidden * p->channels * dl / 4;
    const long long per = bf ? n4 * 6 : n4 * 4;       // floats per packed matrix
    const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    const bool gated = p->field_kind == NCDE_FIELD_MINIMAL;
    if (mode == 2) {
        hipLaunchKernelGGL(ncde_pack_panels_h2, dim3(grid), dim3(256), 0, st, a->Wo, (unsigned*)dst, p->hidden, p->channels, dl / 32, wfault);
        if (gated) hipLaunchKernelGGL(ncde_pack_panels_h2, dim3(grid), dim3(256), 0, st, a->Wg, (unsigned*)(dst + per), p->hidden, p->channels, dl / 32, wfault);
    } else if (bf) {
        hipLaunchKernelGGL(ncde_pack_panels_bf, dim3(grid), dim3(256), 0, st, a->Wo, (unsigned*)dst, p->hidden, p->channels, dl / 32);
        if (gated) hipLaunchKernelGGL(ncde_pack_panels_bf, dim3(grid), dim3(256), 0, st, a->Wg, (unsigned*)(dst + per), p->hidden, p->channels, dl / 32);
    } else {
        hipLaunchKernelGGL(ncde_pack_panels, dim3(grid), dim3(256), 0, st, a->Wo, dst, p->hidden, p->channels, dl / 16);
        if (gated) hipLaunchKernelGGL(ncde_pack_panels, dim3(grid), dim3(256), 0, st, a->Wg, dst + per, p->hidden, p->channels, dl / 16);
    }
    a->Wo_pk = dst;
    if (gated) a->Wg_pk = dst + per;
}

size_t tiled_fwd_lds(const NcdeProblem* p, int ns) {
    // (+ the split image of x_L for the bf16 output tiles, NS = 1: 6 bytes per element)
    const size_t xb = ns == 1 ? (size_t)p->layer_out[p->n_layers - 1] * 16 * 6 : 0;
    if (p->field_input != NCDE_INPUT_MATMUL) {      // direct modes: the stage input carries H + C rows (padded), D covers it
        const int d0p = (p->layer_in[0] + 15) & ~15, D = std::max(tiled_dmax(p), d0p);
        return sizeof(float) * (size_t)(ns * 16) * (size_t)(std::max(p->hidden, d0p) + p->hidden + 2 * D + p->channels) + xb;
    }
    return sizeof(float) * (size_t)(ns * 16) * (size_t)(2 * p->hidden + 2 * tiled_dmax(p) + p->channels) + xb;
}

// forward sample tiles per workgroup.  More tiles = more reuse of each weight fragment, but measured on MI355X
// (cfg5, B = 4096: NS1 515 ms, NS2 777 ms, NS4 1273 ms) a workgroup on every CU beats reuse: take the largest NS
// that still leaves >= 256 workgroups (one per CU).  Development flags 0x1000/0x2000/0x4000 force NS = 1/2/4.
int tiled_fwd_ns(const NcdeProblem* p) {
    auto fits = [&](int ns) { return tiled_fwd_lds(p, ns) <= (size_t)kLdsLimit && p->hidden * ns * 16 <= TL_EMAX * TL_THREADS; };
    if (p->flags & NCDE_FLAG_TILED_NS1) return fits(1) ? 1 : 0;
    if (p->flags & NCDE_FLAG_TILED_NS2) return fits(2) ? 2 : 0;
    if (p->flags & NCDE_FLAG_TILED_NS4) return fits(4) ? 4 : 0;
    for (int ns = 4; ns >= 2; ns >>= 1)
        if (fits(ns) && (p->batch + 16 * ns - 1) / (16 * ns) >= 256) return ns;
    return fits(1) ? 1 : 0;
}

// ---- adjoint / exact backward -------------------------------------------------------------------------------------
int tiled_adj_pk(const NcdeProblem* p) {
    const int dlast = p->layer_out[p->n_layers - 1];
    // (round 5: 256 -- the W16 mode of the wide instantiation, original field / matmul input only: see tiled_adj_ok)
    return (dlast == 256) ? 16 : ((dlast == 128) ? 8 : (dlast == 64 ? 4 : (dlast == 32 ? 2 : (dlast == 16 ? 1 : 0))));
}

// small square models: every weight fragment register-resident (see ncde_adj_tiled, RES)
bool tiled_adj_res(const NcdeProblem* p) {
    const int pk = tiled_adj_pk(p);
    bool res = p->field_kind != NCDE_FIELD_MINIMAL && pk >= 1 && pk <= 4 && p->hidden == 16 * pk && p->hidden * p->channels / 16 <= 2 * TL_ADJ_NW &&
               p->output != NCDE_OUT_TIMES;      // default time axis only (see ncde_adj_tiled)
    for (int l = 0; l < p->n_layers; ++l) res = res && p->layer_out[l] == 16 * pk && p->layer_in[l] == 16 * pk;
    return res;
}
// round 5: 128 < H <= 256 over hidden widths <= 128 runs the BIGH instantiation of the sweep: 4 waves (one per SIMD, 512 registers)
// ... and so does a last hidden width of 256 (PK = 16: hidden widths 129 .. 256 are zero-padded to it, ncde_abi.hip) at any H <= 256
bool tiled_adj_bigh(const NcdeProblem* p) { return p->hidden > 128 || p->layer_out[p->n_layers - 1] == 256; }
int tiled_adj_nwv(const NcdeProblem* p) { return tiled_adj_bigh(p) ? 4 : TL_ADJ_NW; }
size_t tiled_adj_lds_base(const NcdeProblem* p) {
    const int pk = p->field_input != NCDE_INPUT_MATMUL ? 1 : tiled_adj_pk(p);      // direct modes run the PK = 1 instantiation
    const int scw = std::max(16 * (16 * pk + 4), 16 * pk * 16);
    if (p->field_input != NCDE_INPUT_MATMUL) {      // direct modes: the field input and its cotangent carry H + C (padded) rows
        const int d0p = (p->layer_in[0] + 15) & ~15, U = std::max(p->hidden, d0p);
        const int D = std::max(tiled_dmax(p), d0p);
        return sizeof(float) * (size_t)(2 * U * 16 + 2 * p->hidden * 16 + (p->n_layers + 2) * D * 16 + p->channels * 16 + TL_ADJ_NW * scw);
    }
    int D = 16;      // x_1 .. x_L and the cotangent ping-pong buffers: rows = the widest LAYER (see ncde_adj_tiled)
    for (int l = 0; l < p->n_layers; ++l) D = std::max(D, p->layer_out[l]);
    if (pk == 16)      // W16: the waves' scratches alias [G0 | G1 | KOA]
        return sizeof(float) * (size_t)(4 * p->hidden * 16 + (p->n_layers + 2) * D * 16 + p->channels * 16);
    return sizeof(float) * (size_t)(4 * p->hidden * 16 + (p->n_layers + 2) * D * 16 + p->channels * 16 + tiled_adj_nwv(p) * scw);
}
// hidden matrices resident, output tiles streamed (RES = 2): small square hidden stacks that do not qualify for RES = 1
bool tiled_adj_res2(const NcdeProblem* p) {
    const int pk = tiled_adj_pk(p);
    bool ok = !tiled_adj_res(p) && pk >= 1 && pk <= 4 && p->hidden == 16 * pk && ncde_dev_env("NCDE_TILED_NO_RES2") == nullptr;
    for (int l = 0; l < p->n_layers; ++l) ok = ok && p->layer_out[l] == 16 * pk && p->layer_in[l] == 16 * pk;
    return ok;
}
// split-bf16 record A / recompute in pass B: streamed weights, last hidden width 32 / 64 / 128, room for the split image of
// x_L in the sweep's LDS, and the caller did not ask for plain fp32-input MFMA
bool tiled_adj_bf(const NcdeProblem* p) {
    const int pk = tiled_adj_pk(p);
    return !(p->flags & NCDE_FLAG_FP32_MFMA) && pk >= 2 && pk <= 8 && !tiled_adj_res(p) && p->field_input == NCDE_INPUT_MATMUL &&
           tiled_adj_lds_base(p) + (size_t)pk * 16 * 16 * 6 <= (size_t)kLdsLimit;
}
// ---- XCD-cooperative output phase (ncde_coop.h) --------------------------------------------------------------------------------------
struct CoopPlan {
    bool ok;
    int M, G;      // members per group, groups of ONE launch
    int chunk;     // sample tiles per launch: all of them, or -- round 6, more tiles than CUs -- the largest multiple of M that is resident at once
};
constexpr int kCoopLdsFloats = 2 * 10 * 2 * 64 * 4 + 2 * 8 * 64 * 4 + 2 * 80 * 16 + 2 * 256 + 2 * 64 + 2 * 4 * 64 + 2 * 16 + 8;      // = COOP_LDS of ncde_adj_tiled
size_t tiled_coop_lds(const NcdeProblem* p) {
    int D = 16;
    for (int l = 0; l < p->n_layers; ++l) D = std::max(D, p->layer_out[l]);
    return sizeof(float) * (size_t)(4 * p->hidden * 16 + (p->n_layers + 2) * D * 16 + p->channels * 16 + kCoopLdsFloats) + (size_t)128 * 16 * 6;
}
int tiled_device_cus() {      // of the CURRENT device (cached per device: ADVICE round 5)
    static std::mutex mu;
    static int cus[64];      // 0 = not asked yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;      // (no device: the host-only queries of the CPU test-suite)
    std::lock_guard<std::mutex> lk(mu);
    if (cus[dev] == 0) {
        int n = 0;
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cus[dev];
}

// ---- run-time gate of the cooperative launches (round 6) -------------------------------------------------------------
// The cooperative kernels spin on each other, so every workgroup of a launch must be resident at once.  What is checked here, at
// launch time, beyond the static plan: (1) the runtime's own occupancy figure for THIS kernel with THIS much LDS says the grid fits
// the device; (2) no cooperative sequence enqueued on ANOTHER stream of this device is still in flight (two such launches could each
// hold CUs the other needs: both would spin until they give up).  A call that fails either test simply runs the per-workgroup kernels.
// What the gate cannot see -- another process's kernels, a CU mask -- ends in the kernels' bounded spin, the call's status word and
// the re-execution by the per-workgroup kernels enqueued behind (never a hang, never a wrong result).
struct CoopFlight {
    hipEvent_t ev;
    hipStream_t stream;
    bool valid;
};
std::mutex g_coop_mu;
CoopFlight g_coop_flight[64];
bool coop_runtime_ok(const void* fn, int threads, size_t lds, int grid, hipStream_t st) {
    int dev = 0, nb = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, threads, lds) != hipSuccess || nb < 1) { (void)hipGetLastError(); return false; }
    if ((long long)nb * tiled_device_cus() < grid) return false;
    std::lock_guard<std::mutex> lk(g_coop_mu);
    const CoopFlight& f = g_coop_flight[dev];
    if (f.valid && f.stream != st && hipEventQuery(f.ev) == hipErrorNotReady) return false;
    (void)hipGetLastError();
    return true;
}
void coop_mark_in_flight(hipStream_t st) {      // after the last cooperative launch of a call
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
    std::lock_guard<std::mutex> lk(g_coop_mu);
    CoopFlight& f = g_coop_flight[dev];
    if (!f.valid) {
        if (hipEventCreateWithFlags(&f.ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return; }
        f.valid = true;
    }
    f.stream = st;
    if (hipEventRecord(f.ev, st) != hipSuccess) (void)hipGetLastError();
}
// The sample tiles of the launch split into G groups of M workgroups; each member keeps COOP_RPM = 20 row tiles of Wo (whole
// state-unit blocks: C/4 in {5, 10, 20}) in registers.  Original field, matmul input, last hidden width 128, H <= 128, and every
// workgroup resident at once (one per CU).
CoopPlan tiled_coop_plan(const NcdeProblem* p) {
    CoopPlan c{false, 0, 0, 0};
    if (p->flags & (NCDE_FLAG_NO_COOP | NCDE_FLAG_FP32_MFMA | NCDE_FLAG_DEBUG_PROFILE)) return c;
    if (p->field_kind != NCDE_FIELD_ORIGINAL || p->field_input != NCDE_INPUT_MATMUL || p->n_layers < 1) return c;      // (any time axis: round 6)
    if (p->layer_out[p->n_layers - 1] != 128 || p->hidden > 128 || p->hidden % 16 || p->channels % 4) return c;
    const int ncq = p->channels / 4;
    if (ncq != 5 && ncq != 10 && ncq != 20) return c;
    const int row_tiles = (p->hidden / 4) * ncq;
    if (row_tiles % COOP_RPM) return c;
    c.M = row_tiles / COOP_RPM;
    const int n_tiles = (p->batch + 15) / 16;
    if (c.M < 4 || c.M % 4 || n_tiles % c.M || c.M > tiled_device_cus()) return c;
    // every workgroup of a launch must be resident at once (one per CU): a batch of more tiles runs as several launches over CHUNKS of the
    // batch, one after the other on the stream (samples are independent; the parameter-gradient partials add up)
    c.chunk = std::min(n_tiles, tiled_device_cus() / c.M * c.M);
    c.G = c.chunk / c.M;
    if (tiled_coop_lds(p) > (size_t)kLdsLimit) return c;
    CoopDims d{p->hidden, p->channels, 128, c.M, c.G};
    if (d.per_tile() * c.chunk * 4 > 0x7ff00000LL) return c;      // the exchange area is addressed with 32-bit byte offsets
    c.ok = true;
    return c;
}

size_t tiled_adj_lds(const NcdeProblem* p) {
    return tiled_adj_lds_base(p) + (tiled_adj_bf(p) ? (size_t)tiled_adj_pk(p) * 16 * 16 * 6 : 0);
}

// The forward's cooperative output phase (ncde_fwd_tiled<.., COOP>): same groups, same packed weight images, an exchange area without
// the partials.  Workspace regions (float offsets) behind the split-fp16 forward's own [h2 copy | bf16 copy | fault words].
struct FwdCoopPlan {
    bool ok;
    int M, G, chunk;
    long long img, x, scale, sync, end;
};
constexpr int kFwdCoopLdsFloats = 2 * 2 * 2048 + 2 * 2 * 1280 + 2 * 2 * 64 + 2 * 8 * 64 + 8 * 16 + 64 + 320 + 4 * 4 * 256 + 8;      // = CBX .. CFL of ncde_fwd_tiled
FwdCoopPlan tiled_fwd_coop_plan(const NcdeProblem* p) {
    FwdCoopPlan f{};
    if (p->field_input != NCDE_INPUT_MATMUL || tiled_fwd_split(p) != 2 || tiled_fwd_ns(p) != 1) return f;
    const CoopPlan c = tiled_coop_plan(p);
    if (!c.ok) return f;
    const size_t lds = sizeof(float) * ((size_t)16 * (size_t)(2 * p->hidden + 2 * tiled_dmax(p) + p->channels) + (size_t)kFwdCoopLdsFloats);
    if (lds > (size_t)kLdsLimit) return f;
    const int n_tiles = (p->batch + 15) / 16;
    const CoopDims d{p->hidden, p->channels, 128, c.M, c.G};
    long long off = 64 + tiled_pack_floats(p, false) + tiled_pack_floats(p, true) + tiled_fault_floats(p);
    off = (off + 63) & ~63LL;
    f.img = off; off += (long long)c.M * (coop_p_words() + coop_t_words());
    f.x = off; off += d.per_tile_fwd() * c.chunk;
    f.scale = off; off += 64;      // [0] sw, [1] 1 / sw, [8] max |Wo| bits, [16] the call's status word (KArgs.coop_status)
    f.sync = off; off += 64 + coop_sync_words(c.G, c.chunk);
    f.end = off + 64;
    f.ok = true; f.M = c.M; f.G = c.G; f.chunk = c.chunk;
    (void)n_tiles;
    return f;
}

bool tiled_adj_ok(const NcdeProblem* p) {
    const bool direct = p->field_input != NCDE_INPUT_MATMUL;
    const bool bigh = tiled_adj_bigh(p);
    // (round 5: any channel count -- beyond 80 / 160 the sweep reads dX/dt in its bookkeeping phase instead of a stage ahead; hidden
    // sizes up to 256 on the BIGH instantiation: original field, matmul input, streamed weights, last width 16 .. 256)
    if ((!direct && tiled_adj_pk(p) == 0) || p->hidden * 16 > 4096 || p->channels > 4095) return false;
    if (bigh && (direct || p->field_kind != NCDE_FIELD_ORIGINAL || tiled_adj_pk(p) < 1)) return false;
    int l1 = -1;
    if (direct && (p->hidden / 16) * (p->layer_out[p->n_layers - 1] / 16) > 64) return false;      // direct heads: [H][dlast] tiles
    // hidden-dW accumulator tiles per matrix (registers of the sweep; W16 keeps them in its global partial: no limit)
    const int max_dw_tiles = tiled_adj_pk(p) == 16 ? 1 << 30 : (bigh ? 128 : 64);
    if (tiled_adj_pk(p) == 16) {      // the 4 scratches of 16 x 132 floats fit [G0 | G1 | KOA] (D = 256 here: always)
        if (2 * 256 * 16 + p->hidden * 16 < 4 * 16 * 132) return false;
    }
    for (int l = 0; l < p->n_layers; ++l) {
        if (p->layer_out[l] > 64 * tiled_adj_nwv(p) || (p->layer_out[l] / 16) * ((p->layer_in[l] + 15) / 16) > max_dw_tiles) return false;
        for (int q = 0; q < l; ++q)
            if ((p->layer_W[l] == p->layer_W[q]) != (p->layer_b[l] == p->layer_b[q])) return false;
        if (l >= 1 && p->layer_W[l] != p->layer_W[0]) {   // at most two distinct matrices: layer 0's and ONE other
            if (l1 < 0) l1 = l;
            else if (p->layer_W[l] != p->layer_W[l1]) return false;
        }
    }
    return tiled_adj_lds(p) <= (size_t)kLdsLimit;
}

struct TiledAdjPlan {
    int n_st, n_sc, gstride, parts, parts_pw, window, S, nrt;      // parts_pw: part-groups of the per-workgroup sequence's pass B
    long long recS;      // cooperative sweep: per (stage, tile) scales for ncde_dwo_h2
    long long recA, recB, recC, recD, gpartA, gpartB, carry, pack, pack_bf, total;   // float offsets into the workspace
    long long theta_o;
    // cooperative output phase: packed weight images, exchange area, {absmax bits, sw, 1/sw}, sync words
    bool coop;
    int coop_M, coop_G, coop_chunk;
    long long coop_img, coop_x, coop_scale, coop_sync, coop_state;
};

// Record budget of one time window.  The continuous adjoint exists to be O(1) in memory (torchcde README: "slower but more
// memory efficient"), so the per-stage records pass B consumes are kept for a WINDOW of steps only: the sweep (pass A) runs W
// steps, the output-layer gradient pass (pass B) folds those W steps into its accumulators, and the record is reused.  The
// default budget is sized to stay resident in the 256 MB Infinity Cache between the two passes (NCDE_FLAG_TILED_WINDOW_STEPS(n) overrides).
// (round 6: the cooperative sweep's pass B, ncde_dwo_h2, reads every record ONCE per XCD -- its workgroups share them through LDS and
// the L2 --, so its records need not stay cache-resident: 768 MB, windows of 20 steps at cfg5, measured 223 against 231 us per stage)
long long tiled_window_budget_bytes(bool coop) {
    const char* e = ncde_dev_env("NCDE_TILED_WINDOW_MB");
    if (e && atof(e) > 0.0) return (long long)(atof(e) * (double)(1LL << 20));
    return coop ? 768LL << 20 : 192LL << 20;
}

TiledAdjPlan tiled_adj_plan(const NcdeProblem* p, const Layout& y) {
    TiledAdjPlan t{};
    const int S = p->method == NCDE_RK4_38 ? 4 : (p->method == NCDE_MIDPOINT ? 2 : 1);
    const long long dlast = y.dlast;
    t.S = S;
    t.n_st = (p->batch + 15) / 16;
    t.gstride = y.gWo_off;
    const bool bf = tiled_adj_bf(p);
    if (p->field_input != NCDE_INPUT_MATMUL) {      // direct modes: no records, no pass B; partials carry every parameter
        t.nrt = 1; t.parts = 1;
        t.gstride = y.theta_size;
        t.window = p->output == NCDE_OUT_TIMES ? std::max(p->n_steps_fwd, p->n_steps_adj) : p->n_knots - 1;
        t.n_sc = t.window * S;
        long long off = 64;
        t.recA = t.recB = t.recC = t.recD = off;
        t.gpartA = off; off += (long long)t.n_st * t.gstride;
        t.gpartB = off;
        t.carry = off; off += 2LL * t.n_st * p->hidden * 16;
        t.pack = off; off += (long long)p->layer_out[0] * ((p->layer_in[0] + 15) & ~15);      // layer 0, columns padded
        t.pack_bf = off;
        t.total = off + 64;
        return t;
    }
    const long long recA_tile = bf ? dlast * 24 : dlast * 16;      // floats per (stage, sample tile) of record A
    const long long per_step = (long long)S * t.n_st * (2 * recA_tile + (p->hidden + p->channels) * 16) * (long long)sizeof(float);
    const int steps = p->output == NCDE_OUT_TIMES ? std::max(p->n_steps_fwd, p->n_steps_adj) : p->n_knots - 1;
    t.window = (int)std::max<long long>(1, std::min<long long>(steps, tiled_window_budget_bytes(bf && tiled_coop_plan(p).ok) / per_step));
    const int forced = (int)((p->flags >> 16) & 0xFFu);      // NCDE_FLAG_TILED_WINDOW_STEPS(n)
    if (forced > 0) t.window = (int)std::min<long long>(steps, forced);
    t.n_sc = t.window * S;
    const long long tiles = (long long)t.n_sc * t.n_st;
    long long off = 64;
    t.recA = off; off += tiles * recA_tile;
    t.recB = off; off += bf ? (long long)t.n_sc * ((t.n_st + 1) / 2) * dlast * 48 : tiles * dlast * 16;   // bf: one block per tile PAIR
    t.recC = off; off += tiles * p->hidden * 16;
    t.recD = off; off += tiles * p->channels * 16;
    t.gpartA = off; off += (long long)t.n_st * t.gstride;
    t.theta_o = (long long)p->hidden * p->channels * dlast + (long long)p->hidden * p->channels;
    // pass B wants >= 4096 waves (four rounds of one wave per SIMD) for balance; each part-group = 4 waves of one row tile
    const int row_tiles = p->hidden * p->channels / 16;
    // row tiles per wave of pass B (original field; the gated heads keep one: two weight panels per tile)
    // (PK = 8: four tiles no longer fit the register file -- measured at cfg5: 1 tile 1382 ms, 2 tiles 1335 ms, 4 tiles 1725 ms)
    t.nrt = p->field_kind == NCDE_FIELD_MINIMAL ? 1 : (row_tiles % 4 == 0 && row_tiles >= 64 && dlast < 128 ? 4 : (row_tiles % 2 == 0 && row_tiles >= 32 ? 2 : 1));
    if (dlast == 256) t.nrt = 1;      // (one tile's weights, accumulators and two record fragments of 256 columns are the register file)
    t.parts = 1;
    while (t.parts < 64 && (row_tiles / t.nrt) * 4 * t.parts < 4096 && 4 * t.parts * 2 <= t.n_st) t.parts *= 2;
    t.parts_pw = t.parts;
    const CoopPlan cp = bf ? tiled_coop_plan(p) : CoopPlan{false, 0, 0, 0};
    if (cp.ok) {      // ncde_dwo_h2: 16 row tiles per workgroup, the sample-tile pairs of a stage split over `parts` workgroups
        int pp = 1;
        while (pp * 2 <= 32 && pp * 2 <= cp.chunk / 2) pp *= 2;
        t.parts = pp;
    }
    t.gpartB = off; off += (long long)std::max(t.parts, t.parts_pw) * t.theta_o * (p->field_kind == NCDE_FIELD_MINIMAL ? 2 : 1);
    t.carry = off; off += 2LL * t.n_st * p->hidden * 16;
    t.pack = off; off += tiled_pack_floats(p, false, true);
    t.pack_bf = off; off += (bf && p->field_kind != NCDE_FIELD_MINIMAL) ? tiled_pack_floats(p, true) : 0;
    t.coop = cp.ok;
    if (cp.ok) {
        t.coop_M = cp.M; t.coop_G = cp.G; t.coop_chunk = cp.chunk;
        off = (off + 63) & ~63LL;
        t.recS = off; off += tiles * 32 + 512;      // (+ the over-read of the last 1 KB chunk ncde_dwo_h2 fetches)
        const CoopDims d{p->hidden, p->channels, 128, cp.M, cp.G};
        off = (off + 63) & ~63LL;
        t.coop_img = off; off += (long long)cp.M * (coop_p_words() + coop_t_words());
        t.coop_x = off; off += d.per_tile() * cp.chunk;
        t.coop_scale = off; off += 64;      // [0] sw, [1] 1 / sw, [8] max |Wo| bits, [16] the call's status word (KArgs.coop_status)
        t.coop_state = off; off += (long long)t.n_st * 2 * 8 * 512 * 4;      // the sweep's hidden-dW accumulators while its registers hold Wo: 2 x TL_DWT x NT float4 per workgroup
        t.coop_sync = off; off += 64 + coop_sync_words(cp.G, cp.chunk);
    }
    t.total = off + 64;
    return t;
}

}  // namespace

bool ncde_tiled_supported(const NcdeProblem* p, int pass) {
    if (p->n_layers < 1 || p->hidden % 16 || p->channels % 4) return false;
    auto aligned = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    if (p->field_kind == NCDE_FIELD_GRU) return false;
    if (p->field_input != NCDE_INPUT_MATMUL) {
        // evaluate / derivative inputs (one sample tile per workgroup; layer 0 is re-laid out with its H + C columns padded to a
        // multiple of 16, so its own alignment does not matter)
        if (p->layer_in[0] != p->hidden + p->channels) return false;
        for (int l = 0; l < p->n_layers; ++l)
            if (p->layer_out[l] % 16 || (l > 0 && (p->layer_in[l] % 16 || !aligned(p->layer_W[l]))) || !aligned(p->layer_b[l])) return false;
        if (!aligned(p->Wo) || !aligned(p->bo)) return false;
        if (p->field_kind == NCDE_FIELD_MINIMAL && (!aligned(p->Wg) || !aligned(p->bg))) return false;
        if (pass != 0) return tiled_adj_ok(p);
        return tiled_fwd_lds(p, 1) <= (size_t)kLdsLimit && p->hidden * 16 <= TL_EMAX * TL_THREADS;
    }
    // field variants: the minimal-gated field with the matmul input (a second head on the same activations)
    if (p->field_kind == NCDE_FIELD_MINIMAL) {
        const int nkb = p->layer_out[p->n_layers - 1] / 16;
        if (!aligned(p->Wg) || !aligned(p->bg) || !(nkb == 1 || nkb == 2 || nkb == 4 || nkb == 8)) return false;
    }
    for (int l = 0; l < p->n_layers; ++l)
        if (p->layer_out[l] % 16 || p->layer_in[l] % 16 || !aligned(p->layer_W[l]) || !aligned(p->layer_b[l])) return false;
    if (!aligned(p->Wo) || !aligned(p->bo)) return false;
    if (pass != 0) return tiled_adj_ok(p);
    return tiled_fwd_ns(p) > 0;
}

// Is the tiled family the better choice?  Measured on MI355X it is wherever it applies: forward cfg5 0.51 s vs 2.54 s
// generic; backward cfg5 1.38 s vs 13.5 s, cfg4 14.1 ms vs 24.0 ms (the generic kernel keeps its per-workgroup
// gradient partial in global memory once it no longer fits LDS).  The hook stays for shapes that measure otherwise.
bool ncde_tiled_preferred(const NcdeProblem* p, int pass) {
    (void)p; (void)pass;
    return true;
}

const char* ncde_tiled_kernel_name(const NcdeProblem* p, int pass) {
    if (!ncde_tiled_supported(p, pass)) return nullptr;
    const bool gated = p->field_kind == NCDE_FIELD_MINIMAL;
    if (pass >= 1 && p->field_input != NCDE_INPUT_MATMUL)
        return pass == 1 ? (gated ? "ncde_adj_tiled<gated,direct>" : "ncde_adj_tiled<direct>") : (gated ? "ncde_adj_tiled<gated,direct,discrete>" : "ncde_adj_tiled<direct,discrete>");
    if (pass >= 1 && tiled_adj_bigh(p)) {      // 128 < H <= 256: the one-wave-per-SIMD instantiation of the sweep
        if (tiled_adj_bf(p)) return pass == 1 ? "ncde_adj_tiled<wide,bf16>+ncde_dwo_pair" : "ncde_adj_tiled<wide,discrete,bf16>+ncde_dwo_pair";
        return pass == 1 ? "ncde_adj_tiled<wide>+ncde_dwo_tiled" : "ncde_adj_tiled<wide,discrete>+ncde_dwo_tiled";
    }
    if (pass >= 1 && tiled_adj_bf(p) && !gated && tiled_coop_plan(p).ok)      // weight-stationary output phase across the workgroups of an XCD
        return pass == 1 ? "ncde_adj_tiled<coop,fp16x2>+ncde_dwo_h2<fp16x2 records>" : "ncde_adj_tiled<coop,discrete,fp16x2>+ncde_dwo_h2<fp16x2 records>";
    if (pass >= 1 && tiled_adj_bf(p)) {      // split-bf16 records: the pair kernel is pass B
        if (pass == 1) return gated ? "ncde_adj_tiled<gated,bf16>+ncde_dwo_pair" : "ncde_adj_tiled<bf16>+ncde_dwo_pair";
        return gated ? "ncde_adj_tiled<gated,discrete,bf16>+ncde_dwo_pair" : "ncde_adj_tiled<discrete,bf16>+ncde_dwo_pair";
    }
    if (pass == 1) return gated ? "ncde_adj_tiled<gated>+ncde_dwo_tiled" : "ncde_adj_tiled+ncde_dwo_tiled";
    if (pass == 2) return gated ? "ncde_adj_tiled<gated,discrete>+ncde_dwo_tiled" : "ncde_adj_tiled<discrete>+ncde_dwo_tiled";
    if (p->field_input != NCDE_INPUT_MATMUL) return gated ? "ncde_fwd_tiled<NS1,gated,direct>" : "ncde_fwd_tiled<NS1,direct>";
    const int ns = tiled_fwd_ns(p);
    if (!gated && tiled_fwd_coop_plan(p).ok) return "ncde_fwd_tiled<NS1,coop,fp16x2>";
    if (tiled_fwd_split(p) == 2) return gated ? "ncde_fwd_tiled<NS1,gated,fp16x2>" : "ncde_fwd_tiled<NS1,fp16x2>";
    if (tiled_fwd_bf(p)) return gated ? "ncde_fwd_tiled<NS1,gated,bf16>" : "ncde_fwd_tiled<NS1,bf16>";
    if (gated) return ns == 4 ? "ncde_fwd_tiled<NS4,gated>" : (ns == 2 ? "ncde_fwd_tiled<NS2,gated>" : "ncde_fwd_tiled<NS1,gated>");
    return ns == 4 ? "ncde_fwd_tiled<NS4>" : (ns == 2 ? "ncde_fwd_tiled<NS2>" : "ncde_fwd_tiled<NS1>");
}

int64_t ncde_tiled_workspace_bytes(const NcdeProblem* p, int pass) {
    if (!ncde_tiled_supported(p, pass)) return NCDE_ERR_UNSUPPORTED;
    if (pass == 0 && p->field_input != NCDE_INPUT_MATMUL) return 256 + (int64_t)sizeof(float) * p->layer_out[0] * ((p->layer_in[0] + 15) & ~15);
    if (pass == 0) {
        const int split = tiled_fwd_split(p);      // split-fp16: its own copy + the split-bf16 copy of the re-execution launch + fault words
        const FwdCoopPlan fc = tiled_fwd_coop_plan(p);
        if (fc.ok) return (int64_t)sizeof(float) * fc.end;
        if (split == 2) return 256 + (tiled_pack_floats(p, false) + tiled_pack_floats(p, true) + tiled_fault_floats(p)) * (int64_t)sizeof(float);
        return 256 + tiled_pack_floats(p, split == 1) * (int64_t)sizeof(float);
    }
    const Layout y = make_layout(p);
    return (int64_t)sizeof(float) * tiled_adj_plan(p, y).total;
}

int64_t ncde_tiled_status_offset(const NcdeProblem* p, int pass) {
    if (!ncde_tiled_supported(p, pass)) return -1;
    if (pass == 0) {
        if (p->field_input != NCDE_INPUT_MATMUL) return -1;
        const FwdCoopPlan fc = tiled_fwd_coop_plan(p);
        return fc.ok ? (int64_t)sizeof(float) * (fc.scale + 16) : -1;
    }
    if (p->field_input != NCDE_INPUT_MATMUL) return -1;
    const Layout y = make_layout(p);
    const TiledAdjPlan t = tiled_adj_plan(p, y);
    return t.coop ? (int64_t)sizeof(float) * (t.coop_scale + 16) : -1;
}

// forward instantiation with split output tiles: BF = 1 split-bf16, 2 split-fp16; resh = resident hidden fragments (0 / 2 / 4)
template <int BF>
static void (*tiled_fwd_split_fn(bool g, bool small, int resh))(KArgs) {
    if (resh == 2) return g ? ncde_fwd_tiled<1, TL_NW, 4, 1, BF, 2> : ncde_fwd_tiled<1, TL_NW, 4, 0, BF, 2>;
    if (resh == 4) return g ? ncde_fwd_tiled<1, TL_NW, 4, 1, BF, 4> : ncde_fwd_tiled<1, TL_NW, 4, 0, BF, 4>;
    if (g) return small ? ncde_fwd_tiled<1, TL_NW, 4, 1, BF> : ncde_fwd_tiled<1, TL_NW, 16, 1, BF>;
    return small ? ncde_fwd_tiled<1, TL_NW, 4, 0, BF> : ncde_fwd_tiled<1, TL_NW, 16, 0, BF>;
}

int ncde_tiled_forward(const NcdeProblem* p, float* out, float* stages, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!ncde_tiled_supported(p, 0)) return NCDE_ERR_UNSUPPORTED;
    if ((int64_t)ws_bytes < ncde_tiled_workspace_bytes(p, 0)) return NCDE_ERR_WORKSPACE;
    const Layout y = make_layout(p);
    KArgs a;
    fill_kargs(p, y, &a);
    a.out = out;
    a.stages = stages;
    const bool direct = p->field_input != NCDE_INPUT_MATMUL;
    const bool bf = !direct && tiled_fwd_bf(p);
    const int split = direct ? 0 : tiled_fwd_split(p);
    float* wsf = (float*)ws + 64;
    float* pack_bf = wsf + (split == 2 ? tiled_pack_floats(p, false) : 0);      // split-fp16: [h2 copy | bf16 copy | fault words]
    int* fault = split == 2 ? reinterpret_cast<int*>(pack_bf + tiled_pack_floats(p, true)) : nullptr;
    if (direct) {      // layer 0 with its H + C columns zero-padded to a multiple of 16
        const int d0p = (p->layer_in[0] + 15) & ~15, n0 = p->layer_out[0];
        float* w0 = (float*)ws + 64;
        hipLaunchKernelGGL(ncde_pad_columns, dim3((n0 * d0p + 255) / 256), dim3(256), 0, st, a.W[0], w0, n0, p->layer_in[0], d0p);
        a.W[0] = w0;
        a.din[0] = d0p;
    } else if (split == 2) {
        const int n_tiles = (p->batch + 15) / 16;
        if (hipMemsetAsync(fault + n_tiles, 0, sizeof(int), st) != hipSuccess) return NCDE_ERR_HIP;
        tiled_pack_launch(p, &a, wsf, false, st, 2, fault + n_tiles);
        a.fault = fault;
    } else if (tiled_pack_floats(p, bf) > 0) tiled_pack_launch(p, &a, wsf, bf, st);
    const int ns = direct ? 1 : tiled_fwd_ns(p);
    const size_t lds = direct ? tiled_direct_residency(p, &a, tiled_fwd_lds(p, ns)) : tiled_fwd_lds(p, ns);
    const bool small = p->hidden * ns * 16 <= 4 * TL_THREADS;   // state slice of <= 4 elements per thread: fewer live registers
    void (*fn)(KArgs) = ns == 4 ? (small ? ncde_fwd_tiled<4, TL_NW, 4> : ncde_fwd_tiled<4, TL_NW, 16>)
                                : (ns == 2 ? (small ? ncde_fwd_tiled<2, TL_NW, 4> : ncde_fwd_tiled<2, TL_NW, 16>)
                                           : (small ? ncde_fwd_tiled<1, TL_NW, 4> : ncde_fwd_tiled<1, TL_NW, 16>));
    if (p->field_kind == NCDE_FIELD_MINIMAL)
        fn = ns == 4 ? ncde_fwd_tiled<4, TL_NW, 16, 1> : (ns == 2 ? ncde_fwd_tiled<2, TL_NW, 16, 1> : ncde_fwd_tiled<1, TL_NW, 16, 1>);
    void (*fx)(KArgs) = nullptr;      // split-fp16: the split-bf16 instantiation of the same configuration re-executes range-faulted tiles
    if (bf) {
        // small square hidden stack (H = every width = 32 or 64): hidden fragments resident
        bool sq = (p->hidden == 32 || p->hidden == 64) && ncde_dev_env("NCDE_TILED_NO_RES2") == nullptr;
        for (int l = 0; l < p->n_layers; ++l) sq = sq && p->layer_out[l] == p->hidden && p->layer_in[l] == p->hidden;
        const bool g = p->field_kind == NCDE_FIELD_MINIMAL;
        const int resh = !sq ? 0 : (p->hidden == 32 ? 2 : 4);
        fx = tiled_fwd_split_fn<1>(g, small, resh);
        fn = split == 2 ? tiled_fwd_split_fn<2>(g, small, resh) : fx;
    }
    if (direct) {
        if (p->field_kind == NCDE_FIELD_MINIMAL) fn = small ? ncde_fwd_tiled<1, TL_NW, 4, 1, 0, 0, 1> : ncde_fwd_tiled<1, TL_NW, 16, 1, 0, 0, 1>;
        else fn = small ? ncde_fwd_tiled<1, TL_NW, 4, 0, 0, 0, 1> : ncde_fwd_tiled<1, TL_NW, 16, 0, 0, 0, 1>;
    }
    const FwdCoopPlan fc = direct ? FwdCoopPlan{} : tiled_fwd_coop_plan(p);
    const int nwg = (p->batch + ns * 16 - 1) / (ns * 16);
    if (fc.ok) {      // XCD-cooperative, weight-stationary output phase (ncde_coop.h): the reverse sweep's groups and weight images
        void (*fc_fn)(KArgs) = ncde_fwd_tiled<1, TL_NW, 4, 0, 2, 0, 0, 1>;
        const size_t lds_coop = sizeof(float) * ((size_t)16 * (size_t)(2 * p->hidden + 2 * tiled_dmax(p) + p->channels) + (size_t)kFwdCoopLdsFloats);
        if (ncde_lds_optin((const void*)fc_fn, lds_coop) != hipSuccess) return NCDE_ERR_HIP;
        if (coop_runtime_ok((const void*)fc_fn, TL_THREADS, lds_coop, fc.chunk, st)) {      // (else: the per-workgroup kernels below, unconditionally)
            float* w = (float*)ws;
            unsigned* amax = reinterpret_cast<unsigned*>(w + fc.scale + 8);
            unsigned* status = reinterpret_cast<unsigned*>(w + fc.scale + 16);
            if (hipMemsetAsync(amax, 0, sizeof(unsigned), st) != hipSuccess) return NCDE_ERR_HIP;
            if (hipMemsetAsync(status, 0, sizeof(unsigned), st) != hipSuccess) return NCDE_ERR_HIP;
            const int n_tiles = (p->batch + 15) / 16;
            const long long nw = (long long)p->hidden * p->channels * 128;
            hipLaunchKernelGGL(ncde_coop_absmax, dim3(512), dim3(256), 0, st, a.Wo, nw, amax);
            hipLaunchKernelGGL(ncde_coop_pack, dim3(1024), dim3(256), 0, st, a.Wo, (const unsigned*)amax, reinterpret_cast<unsigned*>(w + fc.img), w + fc.scale,
                               p->channels, 128, fc.M);
            a.coop_img = reinterpret_cast<const unsigned*>(w + fc.img);
            a.coop_x = w + fc.x;
            a.coop_scale = w + fc.scale;
            a.coop_sync = reinterpret_cast<unsigned*>(w + fc.sync);
            a.coop_M = fc.M;
            a.coop_G = fc.G;
            a.coop_status = status;
            a.coop_inject = (p->flags & NCDE_FLAG_COOP_FAULT_INJECT) ? 1 : 0;
            a.coop_spin = a.coop_inject ? (1u << 12) : (unsigned)COOP_SPIN_LIMIT;
            // one launch per CHUNK of the batch (all of it unless the batch has more sample tiles than the device has CUs)
            for (int t0 = 0; t0 < n_tiles; t0 += fc.chunk) {
                const int tiles_c = std::min(fc.chunk, n_tiles - t0);
                KArgs ac = a;
                ac.B = std::min(p->batch - 16 * t0, 16 * tiles_c);
                ac.coeffs = a.coeffs + (long long)16 * t0 * a.cs_b;
                ac.z0 = a.z0 + (long long)16 * t0 * a.Hr;
                ac.out = a.out + (long long)16 * t0 * a.n_out * a.Hr;
                if (a.stages) { ac.stages = a.stages + (long long)16 * t0 * a.Hr; ac.Brec = p->batch; }
                if (a.fault) ac.fault = a.fault + t0;
                ac.coop_G = tiles_c / fc.M;
                if (hipMemsetAsync(w + fc.sync, 0, sizeof(unsigned) * (size_t)coop_sync_words(ac.coop_G, tiles_c), st) != hipSuccess) return NCDE_ERR_HIP;
                hipLaunchKernelGGL(fc_fn, dim3(tiles_c), dim3(TL_THREADS), lds_coop, st, ac);
            }
            coop_mark_in_flight(st);
            // Behind it, the per-workgroup kernels with run_if = the status word: they return at once unless the cooperative launch gave
            // up (a workgroup that never became resident: another process's kernels, a CU mask), in which case they redo the solve.
            a.run_if = status;
        }
    }
    if (ncde_lds_optin((const void*)fn, lds) != hipSuccess) return NCDE_ERR_HIP;
    hipLaunchKernelGGL(fn, dim3(nwg), dim3(TL_THREADS), lds, st, a);
    if (split == 2) {      // re-execution of range-faulted sample tiles in split-bf16 (normally none: every workgroup exits at once)
        tiled_pack_launch(p, &a, pack_bf, true, st);
        a.only_faulted = 1;
        if (ncde_lds_optin((const void*)fx, lds) != hipSuccess) return NCDE_ERR_HIP;
        hipLaunchKernelGGL(fx, dim3(nwg), dim3(TL_THREADS), lds, st, a);
    }
    return hipGetLastError() == hipSuccess ? NCDE_OK : NCDE_ERR_HIP;
}

int ncde_tiled_adjoint(const NcdeProblem* p, const float* src, const float* grad_out, const NcdeGrads* g, void* ws, size_t ws_bytes,
                       hipStream_t st, bool main_kernel_only, bool discrete) {
    if (!ncde_tiled_supported(p, discrete ? 2 : 1)) return NCDE_ERR_UNSUPPORTED;
    const Layout y = make_layout(p);
    const TiledAdjPlan t = tiled_adj_plan(p, y);
    if (ws_bytes < sizeof(float) * (size_t)t.total) return NCDE_ERR_WORKSPACE;
    float* w = (float*)ws;
    KArgs a;
    fill_kargs(p, y, &a);
    a.grad_out = grad_out; a.grad_z0 = g->grad_z0;
    if (discrete) { a.stages = const_cast<float*>(src); a.discrete = 1; }
    else a.z_out = src;
    a.recA = w + t.recA; a.recB = w + t.recB; a.recC = w + t.recC; a.recD = w + t.recD;
    a.gpart = w + t.gpartA;
    a.gstride = t.gstride;
    if (p->field_input != NCDE_INPUT_MATMUL) {      // evaluate / derivative inputs: one launch of the direct-mode sweep + the reduction
        const int d0p = (p->layer_in[0] + 15) & ~15, n0 = p->layer_out[0];
        hipLaunchKernelGGL(ncde_pad_columns, dim3((n0 * d0p + 255) / 256), dim3(256), 0, st, a.W[0], w + t.pack, n0, p->layer_in[0], d0p);
        a.W[0] = w + t.pack;
        a.din[0] = d0p;
        a.carry = w + t.carry;
        const bool g1 = p->field_kind == NCDE_FIELD_MINIMAL;
        void (*fd)(KArgs) = g1 ? ncde_adj_tiled<1, TL_ADJ_NW, 0, 1, 0, 1> : ncde_adj_tiled<1, TL_ADJ_NW, 0, 0, 0, 1>;
        const size_t ldsd = tiled_direct_residency(p, &a, tiled_adj_lds(p));      // + LDS-resident copies of the small matrices
        if (ncde_lds_optin((const void*)fd, ldsd) != hipSuccess) return NCDE_ERR_HIP;
        const int n_rs = p->output == NCDE_OUT_TIMES ? (discrete ? p->n_steps_fwd : p->n_steps_adj) : p->n_knots - 1;
        a.win_hi = n_rs; a.win_lo = 0; a.resume = 0;
        hipLaunchKernelGGL(fd, dim3(t.n_st), dim3(64 * TL_ADJ_NW), ldsd, st, a);
        if (hipGetLastError() != hipSuccess) return NCDE_ERR_HIP;
        if (main_kernel_only) return NCDE_OK;
        return launch_reduce_partials(p, y, g, (const float*)a.gpart, t.n_st, st);
    }
    tiled_pack_launch(p, &a, w + t.pack, false, st);
    if (tiled_adj_bf(p) && p->field_kind != NCDE_FIELD_MINIMAL) {
        const int dl = p->layer_out[p->n_layers - 1];
        const long long n4 = (long long)p->hidden * p->channels * dl / 4;
        const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
        hipLaunchKernelGGL(ncde_pack_panels_bf, dim3(grid), dim3(256), 0, st, a.Wo, (unsigned*)(w + t.pack_bf), p->hidden, p->channels, dl / 32);
        a.Wo_bf = (const unsigned*)(w + t.pack_bf);
    }
    const int pk = tiled_adj_pk(p);
    const bool gated = p->field_kind == NCDE_FIELD_MINIMAL;
    const bool res = tiled_adj_res(p), bf = tiled_adj_bf(p);
    void (*fa)(KArgs) = pk == 8 ? ncde_adj_tiled<8, TL_ADJ_NW>
                                : (pk == 4 ? (res ? ncde_adj_tiled<4, TL_ADJ_NW, 1> : ncde_adj_tiled<4, TL_ADJ_NW>)
                                           : (pk == 2 ? (res ? ncde_adj_tiled<2, TL_ADJ_NW, 1> : ncde_adj_tiled<2, TL_ADJ_NW>)
                                                      : (res ? ncde_adj_tiled<1, TL_ADJ_NW, 1> : ncde_adj_tiled<1, TL_ADJ_NW>)));
    void (*fb)(KArgs, int, int, float*) = pk == 8 ? ncde_dwo_tiled<8> : (pk == 4 ? ncde_dwo_tiled<4> : (pk == 2 ? ncde_dwo_tiled<2> : ncde_dwo_tiled<1>));
    if (t.nrt == 4) fb = pk == 8 ? ncde_dwo_tiled<8, 0, 4> : (pk == 4 ? ncde_dwo_tiled<4, 0, 4> : (pk == 2 ? ncde_dwo_tiled<2, 0, 4> : ncde_dwo_tiled<1, 0, 4>));
    if (t.nrt == 2) fb = pk == 8 ? ncde_dwo_tiled<8, 0, 2> : (pk == 4 ? ncde_dwo_tiled<4, 0, 2> : (pk == 2 ? ncde_dwo_tiled<2, 0, 2> : ncde_dwo_tiled<1, 0, 2>));
    void (*fb2)(KArgs, int, int, float*) = nullptr;
    if (gated) {
        fa = pk == 8 ? ncde_adj_tiled<8, TL_ADJ_NW, 0, 1> : (pk == 4 ? ncde_adj_tiled<4, TL_ADJ_NW, 0, 1> : (pk == 2 ? ncde_adj_tiled<2, TL_ADJ_NW, 0, 1> : ncde_adj_tiled<1, TL_ADJ_NW, 0, 1>));
        fb = pk == 8 ? ncde_dwo_tiled<8, 1> : (pk == 4 ? ncde_dwo_tiled<4, 1> : (pk == 2 ? ncde_dwo_tiled<2, 1> : ncde_dwo_tiled<1, 1>));
        fb2 = pk == 8 ? ncde_dwo_tiled<8, 2> : (pk == 4 ? ncde_dwo_tiled<4, 2> : (pk == 2 ? ncde_dwo_tiled<2, 2> : ncde_dwo_tiled<1, 2>));
    }
    const bool res2 = tiled_adj_res2(p);
    if (res2 && !bf) {
        if (gated) fa = pk == 4 ? ncde_adj_tiled<4, TL_ADJ_NW, 2, 1> : (pk == 2 ? ncde_adj_tiled<2, TL_ADJ_NW, 2, 1> : ncde_adj_tiled<1, TL_ADJ_NW, 2, 1>);
        else fa = pk == 4 ? ncde_adj_tiled<4, TL_ADJ_NW, 2> : (pk == 2 ? ncde_adj_tiled<2, TL_ADJ_NW, 2> : ncde_adj_tiled<1, TL_ADJ_NW, 2>);
    }
    if (bf) {
        if (gated) {
            fa = pk == 8 ? ncde_adj_tiled<8, TL_ADJ_NW, 0, 1, 1> : (pk == 4 ? ncde_adj_tiled<4, TL_ADJ_NW, 0, 1, 1> : ncde_adj_tiled<2, TL_ADJ_NW, 0, 1, 1>);
            if (res2) fa = pk == 4 ? ncde_adj_tiled<4, TL_ADJ_NW, 2, 1, 1> : ncde_adj_tiled<2, TL_ADJ_NW, 2, 1, 1>;
            // both heads in one pass where the two accumulator sets fit the register file; at 128 columns one pass per head
            if (pk == 8) { fb = ncde_dwo_pair<8, 1, 1>; fb2 = ncde_dwo_pair<8, 2, 1>; }
            else { fb = pk == 4 ? ncde_dwo_pair<4, 3, 1> : ncde_dwo_pair<2, 3, 1>; fb2 = nullptr; }
        } else {
            fa = pk == 8 ? ncde_adj_tiled<8, TL_ADJ_NW, 0, 0, 1> : (pk == 4 ? ncde_adj_tiled<4, TL_ADJ_NW, 0, 0, 1> : ncde_adj_tiled<2, TL_ADJ_NW, 0, 0, 1>);
            if (res2) fa = pk == 4 ? ncde_adj_tiled<4, TL_ADJ_NW, 2, 0, 1> : ncde_adj_tiled<2, TL_ADJ_NW, 2, 0, 1>;
            if (t.nrt == 4) fb = pk == 8 ? ncde_dwo_pair<8, 0, 4> : (pk == 4 ? ncde_dwo_pair<4, 0, 4> : ncde_dwo_pair<2, 0, 4>);
            else if (t.nrt == 2) fb = pk == 8 ? ncde_dwo_pair<8, 0, 2> : (pk == 4 ? ncde_dwo_pair<4, 0, 2> : ncde_dwo_pair<2, 0, 2>);
            else fb = pk == 8 ? ncde_dwo_pair<8, 0, 1> : (pk == 4 ? ncde_dwo_pair<4, 0, 1> : ncde_dwo_pair<2, 0, 1>);
        }
    }
    const int nwv = tiled_adj_nwv(p);
    if (pk == 16) {      // last hidden width 256: fp32 records, hidden dW in the workgroups' global partials (zero before the first window)
        fa = ncde_adj_tiled<16, 4, 0, 0, 0, 0, 1>;
        fb = ncde_dwo_tiled<16, 0, 1>;
        if (hipMemsetAsync(w + t.gpartA, 0, sizeof(float) * (size_t)t.n_st * t.gstride, st) != hipSuccess) return NCDE_ERR_HIP;
    } else if (tiled_adj_bigh(p)) {      // (tiled_adj_ok: original field, matmul input, pk >= 2)
        if (bf) fa = pk == 8 ? ncde_adj_tiled<8, 4, 0, 0, 1, 0, 1> : (pk == 4 ? ncde_adj_tiled<4, 4, 0, 0, 1, 0, 1> : ncde_adj_tiled<2, 4, 0, 0, 1, 0, 1>);
        else fa = pk == 8 ? ncde_adj_tiled<8, 4, 0, 0, 0, 0, 1> : (pk == 4 ? ncde_adj_tiled<4, 4, 0, 0, 0, 0, 1> : (pk == 2 ? ncde_adj_tiled<2, 4, 0, 0, 0, 0, 1> : ncde_adj_tiled<1, 4, 0, 0, 0, 0, 1>));
    }
    int nwv_launch = nwv;
    size_t lds_launch = tiled_adj_lds(p);
    void (*const fa_pw)(KArgs) = fa;      // the per-workgroup sweep of this problem (what a cooperative sequence falls back to)
    const size_t lds_pw = lds_launch;
    bool coop = false;
    if (t.coop) {
        void (*fc_fn)(KArgs) = ncde_adj_tiled<8, 8, 0, 0, 1, 0, 0, 1>;
        if (ncde_lds_optin((const void*)fc_fn, tiled_coop_lds(p)) != hipSuccess) return NCDE_ERR_HIP;
        coop = coop_runtime_ok((const void*)fc_fn, 64 * 8, tiled_coop_lds(p), t.coop_chunk, st);      // (see coop_runtime_ok: else the per-workgroup kernels)
    }
    if (coop) {      // XCD-cooperative output phase: weights resident in registers, activations exchanged through L2 (ncde_coop.h)
        fa = ncde_adj_tiled<8, 8, 0, 0, 1, 0, 0, 1>;
        nwv_launch = 8;
        lds_launch = tiled_coop_lds(p);
        unsigned* amax = reinterpret_cast<unsigned*>(w + t.coop_scale + 8);
        if (hipMemsetAsync(amax, 0, sizeof(unsigned), st) != hipSuccess) return NCDE_ERR_HIP;
        a.coop_status = reinterpret_cast<unsigned*>(w + t.coop_scale + 16);      // zeroed ONCE per call: a time-out in one window stops the later ones
        if (hipMemsetAsync(a.coop_status, 0, sizeof(unsigned), st) != hipSuccess) return NCDE_ERR_HIP;
        a.coop_inject = (p->flags & NCDE_FLAG_COOP_FAULT_INJECT) ? 1 : 0;
        a.coop_spin = a.coop_inject ? (1u << 12) : (unsigned)COOP_SPIN_LIMIT;
        // the hidden-layer weight gradients accumulate in the workgroups' global partials from the first stage on
        if (hipMemsetAsync(w + t.gpartA, 0, sizeof(float) * (size_t)t.n_st * t.gstride, st) != hipSuccess) return NCDE_ERR_HIP;
        const long long nw = (long long)p->hidden * p->channels * 128;
        hipLaunchKernelGGL(ncde_coop_absmax, dim3(512), dim3(256), 0, st, a.Wo, nw, amax);
        hipLaunchKernelGGL(ncde_coop_pack, dim3(1024), dim3(256), 0, st, a.Wo, (const unsigned*)amax, reinterpret_cast<unsigned*>(w + t.coop_img),
                           w + t.coop_scale, p->channels, 128, t.coop_M);
        a.coop_img = reinterpret_cast<const unsigned*>(w + t.coop_img);
        a.coop_x = w + t.coop_x;
        a.coop_scale = w + t.coop_scale;
        a.coop_state = w + t.coop_state;
        a.coop_sync = reinterpret_cast<unsigned*>(w + t.coop_sync);
        a.win_max = a.coop_sync + coop_sync_words(t.coop_G, t.coop_chunk);      // (behind the sync words of a FULL chunk; a smaller last chunk uses fewer)
        a.coop_M = t.coop_M;
        a.coop_G = t.coop_G;
    }
    void (*const fb_pw)(KArgs, int, int, float*) = fb;      // pass B of the per-workgroup sequence, and its grid
    const dim3 gridB_pw(p->hidden * p->channels / 16 / t.nrt, t.parts_pw);
    dim3 gridB = gridB_pw;
    size_t ldsB = 0;
    int threadsB = 256;
    if (coop) {      // the cooperative sweep writes 2-piece fp16 records: ncde_dwo_h2 folds them (16 row tiles per workgroup, records through LDS)
        fb = ncde_dwo_h2;
        threadsB = 512;
        gridB = dim3(p->hidden * p->channels / 16 / 16 * t.parts);
        a.dw2_parts = t.parts;
        ldsB = sizeof(float) * 2 * (size_t)(2 * 2048 + 4096 + 2 * (p->hidden / 16) * 256 + 2 * ((p->channels + 15) / 16) * 256 + 2 * 256);      // (ncde_dwo2.hip: two record buffers)
        if (ncde_lds_optin((const void*)fb, ldsB) != hipSuccess) return NCDE_ERR_HIP;
        a.recS = w + t.recS;
    }
    const size_t lds = lds_launch;
    if (ncde_lds_optin((const void*)fa, lds) != hipSuccess) return NCDE_ERR_HIP;
    a.carry = w + t.carry;
    float* gB = w + t.gpartB;
    float* gB2 = gB + (long long)t.parts * t.theta_o;
    // time windows, newest first: sweep W steps (pass A), fold their records into the output-layer gradient (pass B)
    const int n_rsteps = p->output == NCDE_OUT_TIMES ? (discrete ? p->n_steps_fwd : p->n_steps_adj) : p->n_knots - 1;
    if (coop) {
        // cooperative sequence: one CHUNK of the batch at a time (all of it unless there are more sample tiles than CUs), each chunk through
        // all its time windows; the hidden-layer partials are per workgroup (rows t0 .. of gpartA), ncde_dwo_h2 keeps adding to gB
        for (int t0 = 0, firstc = 1; t0 < t.n_st; t0 += t.coop_chunk, firstc = 0) {
            const int tiles_c = std::min(t.coop_chunk, t.n_st - t0);
            KArgs ac = a;
            ac.B = std::min(p->batch - 16 * t0, 16 * tiles_c);
            ac.coeffs = a.coeffs + (long long)16 * t0 * a.cs_b;
            ac.grad_out = a.grad_out + (long long)16 * t0 * a.n_out * a.Hr;
            ac.grad_z0 = a.grad_z0 + (long long)16 * t0 * a.Hr;
            if (discrete) { ac.stages = a.stages + (long long)16 * t0 * a.Hr; ac.Brec = p->batch; }
            else ac.z_out = a.z_out + (long long)16 * t0 * a.n_out * a.Hr;
            ac.gpart = a.gpart + (long long)t0 * a.gstride;
            ac.coop_G = tiles_c / t.coop_M;
            ac.win_max = a.coop_sync + coop_sync_words(ac.coop_G, tiles_c);
            for (int hi = n_rsteps, first = 1; hi >= 1; hi -= t.window, first = 0) {
                const int lo = std::max(0, hi - t.window);
                ac.win_hi = hi; ac.win_lo = lo; ac.resume = first ? 0 : 1;
                ac.dw2_accum = (first && firstc) ? 0 : 1;
                // (the sync words of the launch and, behind them, the window's cotangent-bound word: KArgs.win_max)
                if (hipMemsetAsync(a.coop_sync, 0, sizeof(unsigned) * (size_t)(coop_sync_words(ac.coop_G, tiles_c) + 1), st) != hipSuccess) return NCDE_ERR_HIP;
                hipLaunchKernelGGL(fa, dim3(tiles_c), dim3(64 * nwv_launch), lds, st, ac);
                hipLaunchKernelGGL(fb, gridB, dim3(threadsB), ldsB, st, ac, (hi - lo) * t.S, tiles_c, gB);
                if (hipGetLastError() != hipSuccess) return NCDE_ERR_HIP;
            }
        }
        coop_mark_in_flight(st);
    } else
    for (int hi = n_rsteps, first = 1; hi >= 1; hi -= t.window, first = 0) {
        const int lo = std::max(0, hi - t.window);
        a.win_hi = hi; a.win_lo = lo; a.resume = first ? 0 : 1;
        hipLaunchKernelGGL(fa, dim3(t.n_st), dim3(64 * nwv_launch), lds, st, a);
        const int n_sc = (hi - lo) * t.S;
        hipLaunchKernelGGL(fb, gridB, dim3(threadsB), ldsB, st, a, n_sc, t.n_st, gB);
        if (fb2) hipLaunchKernelGGL(fb2, gridB, dim3(256), 0, st, a, n_sc, t.n_st, gB2);
        if (hipGetLastError() != hipSuccess) return NCDE_ERR_HIP;
    }
    if (main_kernel_only) return NCDE_OK;
    // deterministic reductions: hidden-layer partials of the sweep, then the part-group partials of pass B
    ReduceSegs segs{};
    int n = 0;
    for (int l = 0; l < p->n_layers; ++l) {
        bool first = true;
        for (int q = 0; q < l; ++q)
            if (p->layer_W[q] == p->layer_W[l]) first = false;
        if (!first) continue;
        if (!g->grad_layer_W[l] || !g->grad_layer_b[l]) return NCDE_ERR_INVALID;
        segs.off[n] = y.gW_off[l]; segs.len[n] = p->layer_out[l] * p->layer_in[l]; segs.dst[n] = g->grad_layer_W[l]; ++n;
        segs.off[n] = y.gb_off[l]; segs.len[n] = p->layer_out[l]; segs.dst[n] = g->grad_layer_b[l]; ++n;
    }
    segs.n = n;
    hipLaunchKernelGGL(ncde_reduce_partials, dim3((t.gstride + 255) / 256), dim3(256), 0, st, (const float*)a.gpart, t.n_st, t.gstride, segs);
    if (!g->grad_Wo || !g->grad_bo) return NCDE_ERR_INVALID;
    ReduceSegs so{};
    const int wo_sz = p->hidden * p->channels * y.dlast;
    so.n = 2;
    so.off[0] = 0; so.len[0] = wo_sz; so.dst[0] = g->grad_Wo;
    so.off[1] = wo_sz; so.len[1] = p->hidden * p->channels; so.dst[1] = g->grad_bo;
    hipLaunchKernelGGL(ncde_reduce_partials, dim3(((int)t.theta_o + 255) / 256), dim3(256), 0, st, (const float*)gB, t.parts, (int)t.theta_o, so);
    if (gated) {      // the gate head's partials: written by fb2, or by the two-head pass behind the tanh head's
        if (!g->grad_Wg || !g->grad_bg) return NCDE_ERR_INVALID;
        so.dst[0] = g->grad_Wg;
        so.dst[1] = g->grad_bg;
        hipLaunchKernelGGL(ncde_reduce_partials, dim3(((int)t.theta_o + 255) / 256), dim3(256), 0, st, (const float*)gB2, t.parts, (int)t.theta_o, so);
    }
    if (coop) {
        // Behind the cooperative sequence: the WHOLE pass again on the per-workgroup kernels, every launch with run_if = the status word
        // -- each returns at once (a few microseconds per window) unless a cooperative launch gave up, and then they overwrite every
        // output of this call: grad_z0, the hidden-layer partials, the output-layer partials, the reductions.  (cooperative: original
        // field only, so there is no second head.)
        a.run_if = a.coop_status;
        if (ncde_lds_optin((const void*)fa_pw, lds_pw) != hipSuccess) return NCDE_ERR_HIP;
        for (int hi = n_rsteps, first = 1; hi >= 1; hi -= t.window, first = 0) {
            const int lo = std::max(0, hi - t.window);
            a.win_hi = hi; a.win_lo = lo; a.resume = first ? 0 : 1;
            hipLaunchKernelGGL(fa_pw, dim3(t.n_st), dim3(64 * nwv), lds_pw, st, a);
            hipLaunchKernelGGL(fb_pw, gridB_pw, dim3(256), 0, st, a, (hi - lo) * t.S, t.n_st, gB);
            if (hipGetLastError() != hipSuccess) return NCDE_ERR_HIP;
        }
        hipLaunchKernelGGL(ncde_reduce_partials_if, dim3((t.gstride + 255) / 256), dim3(256), 0, st, a.run_if, (const float*)a.gpart, t.n_st, t.gstride, segs);
        so.dst[0] = g->grad_Wo;
        so.dst[1] = g->grad_bo;
        hipLaunchKernelGGL(ncde_reduce_partials_if, dim3(((int)t.theta_o + 255) / 256), dim3(256), 0, st, a.run_if, (const float*)gB, t.parts_pw, (int)t.theta_o, so);
    }
    return hipGetLastError() == hipSuccess ? NCDE_OK : NCDE_ERR_HIP;
}
