// In-sweep adjoint for H = HH = 64, C <= 4 (BASELINE cfg4: cubic control path, midpoint, B = 8192) -- no stage records, no
// second pass: the reverse sweep of (y, a) AND every parameter gradient in ONE persistent kernel, weights and gradient
// accumulators resident in the 512-entry register file for the whole solve.
//
// Why a kernel of its own: at H = 64 the weights in MFMA operand form (Wo as forward and as transposed operand, the two hidden
// matrices both ways) plus the 24,960 gradient accumulators are ~260 KB -- they fit ONE workgroup per CU only when the
// accumulators are shared by all the samples that CU works on.  So: workgroup = 4 waves (one per SIMD, 512 registers each),
// NS x 16 samples (NS = 2 at B >= 8192: one workgroup per CU), and a wave owns a SLICE OF UNITS for all samples:
//   wave q: unit tile q (units 16q .. 16q+15) of every hidden layer, the 64 rows of Wo with h in [16q, 16q+16), the state
//   entries h = 16q + 4t + g of every sample, and the gradient accumulators of exactly those rows.
// The two sample tiles are independent instruction streams inside each wave (what hides the latency of the dependent chain),
// every weight operand register feeds two MFMAs, and the weight-gradient GEMMs contract over all NS x 16 samples at once.
// Activations move between waves "lane-aligned": D rows (g, r) of tile q are exactly what lane (s, g) of every other wave needs
// as K slot (q, r) of its next B operand, so an exchange is one ds_write_b128 + four ds_read_b128 per lane, no transposition.
//   forward side (z, x_l, W: O(1) operands)  2-way split-fp16 MFMA (ncde_bf3.h), range faults re-executed by the fp32-MFMA instance
//   cotangent side + weight gradients          fp32-input MFMA (exact fp32; the register file has no room for 3-way split operands)
// Reference semantics restated: adjoint.py:37-145 (continuous adjoint in negated time), fixed_grid.py:6-29 / rk_common.py:106-114
// (Butcher tables), interpolation_cubic.py:331-336 / interpolation_linear.py:212-234 (dX/dt, left piece at a knot); the exact
// discrete backward (DISC) transposes solvers.py:94-119 stage by stage -- same bookkeeping as ncde_adj_fast3.
#include "ncde_fast64.h"

#include "ncde_common.h"
#include "ncde_bf3.h"
#include "ncde_fastdefs.h"
#include "ncde_host.h"

namespace {

// 4 values -> 2-way split-fp16, packed for the lane-aligned exchange: {h(v0,v1), h(v2,v3), l(v0,v1), l(v2,v3)}
__device__ __forceinline__ u32x4 split4h(const float* v, float& mx) {
    h16x2 h[2];
    u32x4 o;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const f32x2 x = {v[2 * p], v[2 * p + 1]};
        h[p] = __builtin_convertvector(x, h16x2);
        asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(mx) : "v"(v[2 * p]), "v"(v[2 * p + 1]));
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        f32x2 r;
        r[0] = __builtin_fmaf((float)h[p][0], -NCDE_H2_SCALE, v[2 * p] * NCDE_H2_SCALE);  // exact: (x - h1) * 2^11
        r[1] = __builtin_fmaf((float)h[p][1], -NCDE_H2_SCALE, v[2 * p + 1] * NCDE_H2_SCALE);
        const h16x2 l = __builtin_convertvector(r, h16x2);
        o[p] = __builtin_bit_cast(unsigned, h[p]);
        o[2 + p] = __builtin_bit_cast(unsigned, l);
    }
    return o;
}

// One 16-row tile x K = 64 x 16 samples.  K slot (qp, r), qp = producing wave, r = its D register, for this lane's k-sub g.
// P = 1: split-fp16 (chunk cc holds slots qp = 2cc, 2cc+1);  P = 2: fp32-input MFMA (k-step ks = 4 qp + r).
template <int P>
struct OpA;
template <>
struct OpA<1> {
    Split2h c[2];
};
template <>
struct OpA<2> {
    float k[16];
};
template <int P>
using OpB = OpA<P>;

template <int P, class F>
__device__ __forceinline__ OpA<P> load_opA(F wf, float& mx) {   // wf(qp, r) -> weight of this lane's A row at K slot (qp, r)
    OpA<P> A;
    if constexpr (P == 1) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            float tmp[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) tmp[j] = wf(2 * cc + (j >> 2), j & 3);
            A.c[cc] = split8h(tmp, mx);
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) A.k[ks] = wf(ks >> 2, ks & 3);
    }
    return A;
}
// B operand from the four waves' exchange pieces of one sample tile (ex: [4 waves][64 lanes][4 words])
template <int P>
__device__ __forceinline__ OpB<P> read_opB(const float* ex, int lane) {
    u32x4 e[4];
#pragma unroll
    for (int qp = 0; qp < 4; ++qp) e[qp] = *reinterpret_cast<const u32x4*>(ex + qp * 256 + lane * 4);
    OpB<P> B;
    if constexpr (P == 1) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            B.c[cc].hi = (u32x4){e[2 * cc][0], e[2 * cc][1], e[2 * cc + 1][0], e[2 * cc + 1][1]};
            B.c[cc].lo = (u32x4){e[2 * cc][2], e[2 * cc][3], e[2 * cc + 1][2], e[2 * cc + 1][3]};
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) B.k[ks] = __uint_as_float(e[ks >> 2][ks & 3]);
    }
    return B;
}
template <int P>
__device__ __forceinline__ f32x4 tile_mac(const OpA<P>& A, const OpB<P>& B, f32x4 bias) {
    if constexpr (P == 1) {
        f32x4 m = bias, x = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) mfma_split2(A.c[cc], B.c[cc], m, x);
        return h2_combine(m, x);
    } else {
        f32x4 acc = bias;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) acc = mfma16(A.k[ks], B.k[ks], acc);
        return acc;
    }
}
// the same for N sample tiles at once, tile index innermost: consecutive MFMAs belong to independent accumulator chains
template <int P, int N>
__device__ __forceinline__ void tile_mac_n(const OpA<P>& A, const OpB<P> (&B)[N], f32x4 bias, f32x4 (&out)[N]) {
    if constexpr (P == 1) {
        f32x4 m[N], x[N];
#pragma unroll
        for (int i = 0; i < N; ++i) { m[i] = bias; x[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
            for (int i = 0; i < N; ++i) m[i] = mfma_h(A.c[cc].hi, B[i].c[cc].hi, m[i]);
#pragma unroll
            for (int i = 0; i < N; ++i) x[i] = mfma_h(A.c[cc].lo, B[i].c[cc].hi, x[i]);
#pragma unroll
            for (int i = 0; i < N; ++i) x[i] = mfma_h(A.c[cc].hi, B[i].c[cc].lo, x[i]);
        }
#pragma unroll
        for (int i = 0; i < N; ++i) out[i] = h2_combine(m[i], x[i]);
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) out[i] = bias;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
#pragma unroll
            for (int i = 0; i < N; ++i) out[i] = mfma16(A.k[ks], B[i].k[ks], out[i]);
    }
}
template <int P>
__device__ __forceinline__ u32x4 make_piece(const float* v, float& mx) {
    if constexpr (P == 1) return split4h(v, mx);
    else return (u32x4){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
}
template <int P>
__device__ __forceinline__ void write_piece(float* ex, int q, int lane, const float* v, float& mx) {
    u32x4 o;
    if constexpr (P == 1) o = split4h(v, mx);
    else o = (u32x4){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
    *reinterpret_cast<u32x4*>(ex + q * 256 + lane * 4) = o;
}

constexpr int kF64MaxLayers = 4;

// development: s_memtime phase counters (cycles per phase, summed over the sweep) -> the first floats of the workgroup's gradient
// partial, which is garbage in such a build.  Slots: 0 hidden forward, 1 output tiles + dL/dx_L, 2 reduce + dWo, 3 hidden backward,
// 4 a^T df/dy + bookkeeping, 7 time spent waiting at the workgroup barriers.
#ifdef NCDE_F64_PROF
#define F64_TICK(k) { const unsigned long long now_ = __builtin_readcyclecounter(); prof[k] += now_ - tlast; tlast = now_; }
#define F64_SYNC(k) { F64_TICK(k) __syncthreads(); F64_TICK(7) }
#else
#define F64_TICK(k)
#define F64_SYNC(k) __syncthreads();
#endif

// plan_stages = S of a PLAN = 1 instance (the control-path region then holds [2][S][SP][4] instead of the ring of three pieces)
__host__ __device__ constexpr int f64_lds_floats(int nl, int ns, bool cubic, int plan_stages = 0) {
    const int SP = 16 * ns, RS = SP + 4;
    const int dxr = plan_stages ? 2 * plan_stages * SP * 4 : 3 * SP * (cubic ? 12 : 4);
    return 2 * 64 * RS + nl * 64 * RS + 256 * RS + 2 * 64 * RS + ns * 16 * 256 + 2 * ns * 4 * 256 + dxr + 384 + 4;
}

// HPF: 1 = forward-side GEMMs 2-way split-fp16 (default), 2 = everything fp32-input MFMA (NCDE_FLAG_FP32_MFMA / _SPLIT_BF16, and the
// instance that re-executes range-faulted workgroups of HPF = 1: `only_faulted`)
// PLAN = 1 (continuous adjoint only): the general time axis -- the reverse steps of a.plan's adjoint table, as in ncde_adj_fast3
template <int METHOD, int NS, int HPF, int DISC, int PLAN = 0>
__global__ __launch_bounds__(256, 1) void ncde_adj_h64(KArgs a) {
    static_assert(PLAN == 0 || DISC == 0, "the planned discrete backward runs on the batch-tiled family");
    constexpr int S = kStages<METHOD>, SP = 16 * NS, RS = SP + 4, KS = 4 * NS;
    constexpr int NLM = kF64MaxLayers;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int NL = a.n_layers;
    const bool cubic = a.interp != NCDE_INTERP_LINEAR;
    const int DXW = cubic ? 12 : 4;
    float* zimg = lds;                         // [2][64][RS]      stage input x_0 = y, [h][sample], by stage parity
    float* ximg = zimg + 2 * 64 * RS;          // [NL][64][RS]     x_1 .. x_NL
    float* dpT = ximg + NL * 64 * RS;          // [256][RS]        w * dP, rows of wave q at 64 q (wave-private)
    float* gimg = dpT + 256 * RS;              // [2][64][RS]      w * dL/dpre_l, rows of wave q at 16 q (wave-private), by layer parity
    float* red = gimg + 2 * 64 * RS;           // [NS][4 src][4 tp][256]  per-wave partials of dL/dx_L
    float* exch = red + NS * 16 * 256;         // [2][NS][4][256]  lane-aligned exchange, double-buffered by phase parity
    float* dxs = exch + 2 * NS * 4 * 256;      // [3][SP][DXW]     ring of control-path pieces  (PLAN: [2][S][SP][4], dX/dt per stage)
    float* biasL = dxs + (PLAN ? 2 * S * SP * 4 : 3 * SP * DXW);         // [256] output-layer biases (prescaled) by row h C' + c, C' = 4;  [2][64] hidden biases
    int* fault_s = reinterpret_cast<int*>(biasL + 384);
    if constexpr (HPF != 1) {
        if (a.only_faulted && a.fault[blockIdx.x] == 0) return;
    }
    if constexpr (PLAN != 0) {
        if (a.plan == nullptr || !plan_header_ok(a, S)) return;      // (uniform: before the first barrier)
    }
    const int pw_ = plan_step_words(S);
    const int* padj = PLAN ? a.plan + plan_off_adj(S, a.n_steps_fwd, a.n_out) : nullptr;
    const int n_rsteps = PLAN ? a.n_steps_adj : a.T - 1;
    float mx = 0.0f;

    const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = lane & 15, g = lane >> 4;
    const int b0 = blockIdx.x * SP;
    const int C = a.C;
    if (tid == 0) *fault_s = 0;

    // ---- weights -> registers ---------------------------------------------------------------------------------------
    const bool has_inner = NL > 1;
    const float* W0 = a.W[0];
    const float* W1 = has_inner ? a.W[1] : a.W[0];
    OpA<HPF> wo_f[4], w_f[2];
    {
        const int h = tid >> 2, c = tid & 3;
        biasL[tid] = c < C ? NCDE_TANH_PRESCALE * a.bo[h * C + c] : 0.0f;
        if (tid < 128) biasL[256 + tid] = (tid < 64 || has_inner) ? a.b[tid >> 6][tid & 63] : 0.0f;
    }
    const float* bo_l = biasL + (16 * q + g) * 4;            // tile t: + 16 t   (row h = 16q + 4t + g, c = r)
    const float* b_l = biasL + 256 + 16 * q + 4 * g;         // layer slot sl: + 64 sl  (unit 16q + 4g + r)
    float woT[4][4][4];      // [t][tp][r] = Wo[(16q+4t+g) C + r][16 tp + s]: A operand of dL/dx_L = Wo^T dP over this wave's rows
    float w_b[2][16];        // transposed hidden tiles: [1]: W1[unit(ks,g)][16q + s]; [0]: W0[unit(ks,g)][h = 16q + 4(s&3) + (s>>2)]
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int hA = 16 * q + 4 * t + (s >> 2), cA = s & 3;
        wo_f[t] = load_opA<HPF>([&](int qp, int r) { return cA < C ? NCDE_TANH_PRESCALE * a.Wo[(long long)(hA * C + cA) * 64 + 16 * qp + 4 * g + r] : 0.0f; }, mx);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) woT[t][tp][r] = r < C ? a.Wo[(long long)((16 * q + 4 * t + g) * C + r) * 64 + 16 * tp + s] : 0.0f;
        }
    }
    w_f[0] = load_opA<HPF>([&](int qp, int r) { return W0[(16 * q + s) * 64 + 16 * qp + 4 * r + g]; }, mx);          // K slot <-> state entry h
    w_f[1] = load_opA<HPF>([&](int qp, int r) { return has_inner ? W1[(16 * q + s) * 64 + 16 * qp + 4 * g + r] : 0.0f; }, mx);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const int u = 16 * (ks >> 2) + 4 * g + (ks & 3);
        w_b[1][ks] = has_inner ? W1[u * 64 + 16 * q + s] : 0.0f;
        w_b[0][ks] = W0[u * 64 + 16 * q + 4 * (s & 3) + (s >> 2)];
    }

    // ---- control-path staging (reverse order; all 256 threads): element e of the [SP][DXW] image of one piece ---------------
    constexpr int EPT = (SP * 12 + 255) / 256;
    const float* eptr[EPT];
    float eprev[EPT], enext[EPT];
    bool eok[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = tid + k * 256;
        const int es = e / DXW, ec = e - es * DXW;
        const int part = ec >> 2, c = ec & 3;      // cubic: part 0..2 = b, 2c, 3d
        eok[k] = e < SP * DXW && c < a.Cc && (b0 + es) < a.B;      // a.Cc: channels of the coefficient tensor (= C unless zero-padded)
        const long long base = (long long)(eok[k] ? b0 + es : 0) * a.cs_b;
        eptr[k] = a.coeffs + base + (cubic ? (part + 1) * a.Cc + c : c);
        eprev[k] = 0.0f;
        enext[k] = 0.0f;
    }
    auto stage_load = [&](int piece) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) enext[k] = eok[k] ? eptr[k][(long long)piece * a.cs_t] : 0.0f;
    };
    auto stage_store = [&](int piece) {
        float* dst = dxs + (piece % 3) * SP * DXW;
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = tid + k * 256;
            if (e < SP * DXW) dst[e] = cubic ? enext[k] : eprev[k] - enext[k];
            eprev[k] = enext[k];
        }
    };
    constexpr int EPQ = PLAN ? (S * SP * 4 + 255) / 256 : 1;
    float qn[EPQ];
    auto plan_load = [&](const int* pstep) {      // element e = (stage j, sample es, channel c) of a reverse step
#pragma unroll
        for (int k = 0; k < EPQ; ++k) {
            const int e = tid + k * 256;
            const int j = e / (SP * 4), rem = e - j * (SP * 4), es = rem >> 2, c = rem & 3;
            float v = 0.0f;
            if (e < S * SP * 4 && c < a.Cc && b0 + es < a.B) {
                const StageDesc sd = plan_stage(pstep, j);
                const float* p = a.coeffs + (long long)(b0 + es) * a.cs_b + (long long)sd.idx * a.cs_t;
                if (!cubic) {
                    v = p[a.cs_t + c] - p[c];
                    if (sd.kdt != 1.0f) v = v / sd.kdt;
                } else {
                    const float bb = p[a.Cc + c], cc = p[2 * a.Cc + c], dd = p[3 * a.Cc + c];
                    const float inner = cc + dd * sd.frac;
                    v = bb + inner * sd.frac;
                }
            }
            qn[k] = v;
        }
    };
    auto plan_store = [&](int buf) {
#pragma unroll
        for (int k = 0; k < EPQ; ++k) {
            const int e = tid + k * 256;
            if (e < S * SP * 4) dxs[buf * (S * SP * 4) + e] = qn[k];
        }
    };
    const int p_hi = a.n_pieces - 1;
    if constexpr (PLAN != 0) {
        plan_load(padj);
        plan_store(0);
    } else {
        if (!cubic) {
#pragma unroll
            for (int k = 0; k < EPT; ++k) eprev[k] = eok[k] ? eptr[k][(long long)(p_hi + 1) * a.cs_t] : 0.0f;
        }
        stage_load(p_hi);
        stage_store(p_hi);
        if (p_hi >= 1) {
            stage_load(p_hi - 1);
            stage_store(p_hi - 1);
        }
    }

    // ---- state: entries h = 16q + 4t + g of sample (hf, s) ---------------------------------------------------------------
    const int last_row = a.n_out - 1;
    const int Hr = a.Hr;      // row width of z_out / grad_out / grad_z0 / the stage record (= 64 unless the problem was zero-padded)
    bool valid[NS];
    long long brow[NS];
    float y0[NS][4], ky1[NS][4], ky2[NS][4], a0[NS][4], ka1[NS][4], ka2[NS][4], as_[NS][4];
    float znext[NS][4];      // DISC: recorded stage input, fetched one stage ahead
    auto rec_fetch = [&](int lin) {
#pragma unroll
        for (int hf = 0; hf < NS; ++hf)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                znext[hf][t] = (valid[hf] && 16 * q + 4 * t + g < Hr) ? a.stages[((long long)lin * a.B + (b0 + 16 * hf + s)) * Hr + 16 * q + 4 * t + g] : 0.0f;
    };
#pragma unroll
    for (int hf = 0; hf < NS; ++hf) {
        const int b = b0 + 16 * hf + s;
        valid[hf] = b < a.B;
        brow[hf] = (long long)(valid[hf] ? b : 0) * a.n_out;
    }
    if constexpr (DISC != 0) rec_fetch((a.T - 1) * S - 1);
    int zp = 0;
#pragma unroll
    for (int hf = 0; hf < NS; ++hf) {
        float zs[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const long long o = (brow[hf] + last_row) * Hr + 16 * q + 4 * t + g;
            const bool live = valid[hf] && 16 * q + 4 * t + g < Hr;
            y0[hf][t] = (DISC == 0 && live) ? a.z_out[o] : 0.0f;
            a0[hf][t] = live ? a.grad_out[o] : 0.0f;
            as_[hf][t] = (DISC != 0 && METHOD == NCDE_RK4_38) ? a0[hf][t] * 0.125f : a0[hf][t];
            ky1[hf][t] = ky2[hf][t] = ka1[hf][t] = ka2[hf][t] = 0.0f;
            zs[t] = DISC != 0 ? znext[hf][t] : y0[hf][t];
            zimg[(zp * 64 + 16 * q + 4 * t + g) * RS + 16 * hf + s] = zs[t];
        }
        write_piece<HPF>(exch + (1 * NS + hf) * 1024, q, lane, zs, mx);      // phase 0 reads buffer 1
    }
    // ---- gradient accumulators ---------------------------------------------------------------------------------------------
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 gWo[4][4], gW[2][4];      // dWo rows (16q+4t+g, c = r) x units 16 ct + s;  dW_slot rows 16q+4g+r x inputs 16 ct + s
    float gbo[4][4], gb[2][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) gWo[t][ct] = zero4;
#pragma unroll
        for (int r = 0; r < 4; ++r) gbo[t][r] = 0.0f;
    }
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) gW[sl][ct] = zero4;
#pragma unroll
        for (int r = 0; r < 4; ++r) gb[sl][r] = 0.0f;
    }
    __syncthreads();

    // hidden-layer weight gradient of layer l: dW_l += (w gpre_l) x_l^T over all SP samples (gimg rows are this wave's own, written
    // one phase earlier): operands first (load), MFMAs later (mac), so that the reads do not queue behind the phase's own writes
    auto dw_hidden_load = [&](int l, float (&av)[KS], float (&bv)[4][KS]) {
        const float* gi = gimg + ((l & 1) * 64 + 16 * q + s) * RS + KS * g;
        const float* xi = (l == 0 ? zimg + zp * 64 * RS : ximg + (l - 1) * 64 * RS) + s * RS + KS * g;
#pragma unroll
        for (int v = 0; v < NS; ++v) *reinterpret_cast<f32x4*>(av + 4 * v) = *reinterpret_cast<const f32x4*>(gi + 4 * v);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int v = 0; v < NS; ++v) *reinterpret_cast<f32x4*>(bv[ct] + 4 * v) = *reinterpret_cast<const f32x4*>(xi + 16 * ct * RS + 4 * v);
    };
    auto dw_hidden_mac = [&](int l, const float (&av)[KS], const float (&bv)[4][KS]) {
        if (l == 0) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) gW[0][ct] = mfma16(av[ks], bv[ct][ks], gW[0][ct]);
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) gW[1][ct] = mfma16(av[ks], bv[ct][ks], gW[1][ct]);
        }
    };

#ifdef NCDE_F64_PROF
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
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
            const float tt = DISC != 0 ? (float)(n - 1) + stage_offset(METHOD, S - 1 - j) : -(-(float)n + stage_offset(METHOD, j));
            const int idx = PLAN ? 0 : piece_index(tt, a.n_pieces);
            const float frac = tt - (float)idx;
            const float w = DISC != 0 ? 1.0f : stage_weight(METHOD, j) * dt;      // (dt = 1 on the default axis)
            const float* dxp = PLAN ? dxs + ((rs & 1) * S + j) * SP * 4 : dxs + (idx % 3) * SP * DXW;
            if constexpr (DISC != 0) {
                const int lin = (n - 1) * S + (S - 1 - j);
                if (lin >= 1) rec_fetch(lin - 1);
            }
            const bool at_knot = PLAN ? (j == S - 1 && reset_row >= 0) : (j == S - 1 && (a.output == NCDE_OUT_KNOTS || n == 1));
            const bool reset_y = PLAN != 0 || a.output == NCDE_OUT_KNOTS;      // at such a point: y back to the stored value
            // Every phase below is written "read everything -> compute -> write everything" with the sample tile innermost: the two
            // tiles are independent, but their LDS reads and writes may alias as far as the compiler can tell, and a tile-by-tile body
            // was scheduled exactly so -- tile 1's reads behind tile 0's writes, every phase twice as long (measured with F64_TICK).
            // ---- forward recompute: phase l = layer l (reads exchange buffer (l+1)&1, writes l&1) ---------------------------
#pragma unroll
            for (int l = 0; l < NLM; ++l) {
                if (l < NL) {
                    OpB<HPF> B[NS];
#pragma unroll
                    for (int hf = 0; hf < NS; ++hf) B[hf] = read_opB<HPF>(exch + (((l + 1) & 1) * NS + hf) * 1024, lane);
                    const f32x4 bias = *reinterpret_cast<const f32x4*>(b_l + (l == 0 ? 0 : 64));
                    f32x4 pre[NS];
                    if (l == 0) tile_mac_n<HPF, NS>(w_f[0], B, bias, pre);
                    else tile_mac_n<HPF, NS>(w_f[1], B, bias, pre);
                    float xv[NS][4];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int hf = 0; hf < NS; ++hf) xv[hf][r] = relu_bits(pre[hf][r]);
                    u32x4 piece[NS];
#pragma unroll
                    for (int hf = 0; hf < NS; ++hf) piece[hf] = make_piece<HPF>(xv[hf], mx);
#pragma unroll
                    for (int hf = 0; hf < NS; ++hf) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) ximg[(l * 64 + 16 * q + 4 * g + r) * RS + 16 * hf + s] = xv[hf][r];
                        *reinterpret_cast<u32x4*>(exch + ((l & 1) * NS + hf) * 1024 + q * 256 + lane * 4) = piece[hf];
                    }
                    F64_SYNC(0)
                }
            }
            // ---- phase NL: output tiles of this wave (P, tanh, f, dP) and its partial of dL/dx_L = Wo^T dP ------------------
            float kout[NS][4];
            {
                OpB<HPF> B[NS];
                f32x4 dx[NS];
                float sdx[NS];
#pragma unroll
                for (int hf = 0; hf < NS; ++hf) {
                    B[hf] = read_opB<HPF>(exch + (((NL + 1) & 1) * NS + hf) * 1024, lane);
                    const float* dp = dxp + (16 * hf + s) * (PLAN ? 4 : DXW);
                    if (!cubic || PLAN != 0) {      // (PLAN: the staged values ARE dX/dt of this stage)
                        dx[hf] = *reinterpret_cast<const f32x4*>(dp);
                    } else {      // b + (2c + 3d fr) fr  (interpolation_cubic.py:331-336)
                        const f32x4 cb = *reinterpret_cast<const f32x4*>(dp);
                        const f32x4 cc = *reinterpret_cast<const f32x4*>(dp + 4);
                        const f32x4 cd = *reinterpret_cast<const f32x4*>(dp + 8);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float inner = cc[r] + cd[r] * frac;
                            dx[hf][r] = cb[r] + inner * frac;
                        }
                    }
                    sdx[hf] = (dx[hf][0] + dx[hf][1]) + (dx[hf][2] + dx[hf][3]);
                }
                float dP[NS][4][4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    f32x4 o[NS];
                    tile_mac_n<HPF, NS>(wo_f[t], B, *reinterpret_cast<const f32x4*>(bo_l + 16 * t), o);
#pragma unroll
                    for (int hf = 0; hf < NS; ++hf) {
                        const float a4 = 4.0f * as_[hf][t];
                        float ko = 0.0f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float rr = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(o[hf][r]) + 1.0f);      // tanh = 1 - 2 rr
                            ko = fmaf(rr, dx[hf][r], ko);
                            dP[hf][t][r] = (a4 * dx[hf][r]) * fmaf(-rr, rr, rr);                                     // a dX (1 - tanh^2)
                        }
                        kout[hf][t] = fmaf(-2.0f, ko, sdx[hf]);
                    }
                }
                if (w != 0.0f) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int hf = 0; hf < NS; ++hf) {
                                const float wd = w * dP[hf][t][r];
                                gbo[t][r] += wd;
                                dpT[(64 * q + 16 * t + 4 * g + r) * RS + 16 * hf + s] = wd;
                            }
                }
                f32x4 acc[NS][4];
#pragma unroll
                for (int hf = 0; hf < NS; ++hf)
#pragma unroll
                    for (int tp = 0; tp < 4; ++tp) acc[hf][tp] = zero4;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int tp = 0; tp < 4; ++tp)
#pragma unroll
                            for (int hf = 0; hf < NS; ++hf) acc[hf][tp] = mfma16(woT[t][tp][r], dP[hf][t][r], acc[hf][tp]);
#pragma unroll
                for (int hf = 0; hf < NS; ++hf)
#pragma unroll
                    for (int tp = 0; tp < 4; ++tp) *reinterpret_cast<f32x4*>(red + ((hf * 4 + q) * 4 + tp) * 256 + lane * 4) = acc[hf][tp];
            }
            F64_SYNC(1)
            // ---- phase NL+1: dL/dpre of the last hidden layer (tile q) = sum of the four partials, masked; dWo of this stage --------
            float gpre[NS][4];
            {
                f32x4 part[NS][4];
                float xl[NS][4];
#pragma unroll
                for (int hf = 0; hf < NS; ++hf) {
#pragma unroll
                    for (int qp = 0; qp < 4; ++qp) part[hf][qp] = *reinterpret_cast<const f32x4*>(red + ((hf * 4 + qp) * 4 + q) * 256 + lane * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) xl[hf][r] = ximg[((NL - 1) * 64 + 16 * q + 4 * g + r) * RS + 16 * hf + s];      // own tile of x_L: the ReLU mask
                }
#pragma unroll
                for (int hf = 0; hf < NS; ++hf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = ((part[hf][0][r] + part[hf][1][r]) + part[hf][2][r]) + part[hf][3][r];
                        gpre[hf][r] = xl[hf][r] > 0.0f ? v : 0.0f;
                    }
#pragma unroll
                for (int hf = 0; hf < NS; ++hf) {
                    float dummy = 0.0f;
                    write_piece<2>(exch + (((NL + 1) & 1) * NS + hf) * 1024, q, lane, gpre[hf], dummy);
                }
                if (w != 0.0f) {
                    const bool sl1 = NL - 1 >= 1;
#pragma unroll
                    for (int hf = 0; hf < NS; ++hf)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float wg = w * gpre[hf][r];
                            gb[1][r] += sl1 ? wg : 0.0f;
                            gb[0][r] += sl1 ? 0.0f : wg;
                            gimg[((((NL - 1) & 1) * 64) + 16 * q + 4 * g + r) * RS + 16 * hf + s] = wg;
                        }
                }
            }
            if (w != 0.0f) {      // dWo: rows of this wave x all 64 units of x_L, K = the SP samples (k-step ks, k-sub g <-> sample KS g + ks)
                float bv[4][KS];
                const float* xi = ximg + ((NL - 1) * 64 + s) * RS + KS * g;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int v = 0; v < NS; ++v) *reinterpret_cast<f32x4*>(bv[ct] + 4 * v) = *reinterpret_cast<const f32x4*>(xi + 16 * ct * RS + 4 * v);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float av[KS];
                    const float* di = dpT + (64 * q + 16 * t + s) * RS + KS * g;
#pragma unroll
                    for (int v = 0; v < NS; ++v) *reinterpret_cast<f32x4*>(av + 4 * v) = *reinterpret_cast<const f32x4*>(di + 4 * v);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) gWo[t][ct] = mfma16(av[ks], bv[ct][ks], gWo[t][ct]);
                }
            }
            F64_SYNC(2)
            // ---- phases NL+2 .. 2NL: hidden layers l = NL-1 .. 1 backwards (tile q of W_l^T gpre_l, masked by x_l) ------------
#pragma unroll
            for (int lr = 0; lr < NLM - 1; ++lr) {
                const int l = NL - 1 - lr;      // layer whose transpose runs in this phase (phase index p = NL + 2 + lr)
                if (l >= 1) {
                    const int p = NL + 2 + lr;
                    OpB<2> B[NS];
                    float xl[NS][4];
#pragma unroll
                    for (int hf = 0; hf < NS; ++hf) {
                        B[hf] = read_opB<2>(exch + (((p + 1) & 1) * NS + hf) * 1024, lane);
#pragma unroll
                        for (int r = 0; r < 4; ++r) xl[hf][r] = ximg[((l - 1) * 64 + 16 * q + 4 * g + r) * RS + 16 * hf + s];      // own tile of x_l
                    }
                    float av[KS], bv[4][KS];      // operands of dW_l (gimg written in the previous phase by this wave), requested before the chain
                    if (w != 0.0f) dw_hidden_load(l, av, bv);
                    f32x4 gx[NS];
#pragma unroll
                    for (int hf = 0; hf < NS; ++hf) gx[hf] = zero4;
#pragma unroll
                    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
                        for (int hf = 0; hf < NS; ++hf) gx[hf] = mfma16(w_b[1][ks], B[hf].k[ks], gx[hf]);
#pragma unroll
                    for (int hf = 0; hf < NS; ++hf)
#pragma unroll
                        for (int r = 0; r < 4; ++r) gpre[hf][r] = xl[hf][r] > 0.0f ? gx[hf][r] : 0.0f;
#pragma unroll
                    for (int hf = 0; hf < NS; ++hf) {
                        float dummy = 0.0f;
                        write_piece<2>(exch + ((p & 1) * NS + hf) * 1024, q, lane, gpre[hf], dummy);
                    }
                    if (w != 0.0f) {
                        dw_hidden_mac(l, av, bv);
                        const bool sl1 = l - 1 >= 1;
#pragma unroll
                        for (int hf = 0; hf < NS; ++hf)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float wg = w * gpre[hf][r];
                                gb[1][r] += sl1 ? wg : 0.0f;
                                gb[0][r] += sl1 ? 0.0f : wg;
                                gimg[((((l - 1) & 1) * 64) + 16 * q + 4 * g + r) * RS + 16 * hf + s] = wg;      // gpre_{l-1}: the other buffer
                            }
                    }
                    F64_SYNC(3)
                }
            }
            // ---- phase 2NL+1: a^T df/dy for the state entries of this wave, Butcher bookkeeping, next stage input -----------------
            {
                const int p = 2 * NL + 1;
                OpB<2> B[NS];
#pragma unroll
                for (int hf = 0; hf < NS; ++hf) B[hf] = read_opB<2>(exch + (((p + 1) & 1) * NS + hf) * 1024, lane);
                float av[KS], bv[4][KS];
                if (w != 0.0f) dw_hidden_load(0, av, bv);
                // sequence outputs: the stored state / cotangent of knot n-1 (requested before the MFMAs below)
                float zk[NS][4], gk[NS][4];
                if (at_knot) {
                    const int row = PLAN ? reset_row : (a.output == NCDE_OUT_KNOTS ? n - 1 : 0);
#pragma unroll
                    for (int hf = 0; hf < NS; ++hf)
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const long long o = (brow[hf] + row) * Hr + 16 * q + 4 * t + g;
                            const bool live = valid[hf] && 16 * q + 4 * t + g < Hr;
                            gk[hf][t] = live ? a.grad_out[o] : 0.0f;
                            zk[hf][t] = (DISC == 0 && reset_y && live) ? a.z_out[o] : 0.0f;
                        }
                }
                f32x4 vy[NS];
#pragma unroll
                for (int hf = 0; hf < NS; ++hf) vy[hf] = zero4;
#pragma unroll
                for (int ks = 0; ks < 16; ++ks)
#pragma unroll
                    for (int hf = 0; hf < NS; ++hf) vy[hf] = mfma16(w_b[0][ks], B[hf].k[ks], vy[hf]);
                if (w != 0.0f) dw_hidden_mac(0, av, bv);
                float ys[NS][4];
#pragma unroll
                for (int hf = 0; hf < NS; ++hf) {
                    if constexpr (DISC != 0) {
                        // transpose of the Butcher step (RK4 3/8: c4 = a/8; c3 = 3c4 + d4; c2 = 3c4 - d4 + d3; c1 = c4 + d4 - d3/3 + d2/3;
                        // a += d4 + d3 + d2 + d1), d = vy = dL/dY of this stage
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const float d = vy[hf][t];
                            if constexpr (METHOD == NCDE_RK4_38) {
                                const float c4 = a0[hf][t] * 0.125f;
                                if (j == 0) { ka1[hf][t] = d; as_[hf][t] = 3.0f * c4 + d; }
                                else if (j == 1) { ka2[hf][t] = d; as_[hf][t] = (3.0f * c4 - ka1[hf][t]) + d; }
                                else if (j == 2) { ky1[hf][t] = d; as_[hf][t] = ((c4 + ka1[hf][t]) - 0.333333343267440796f * ka2[hf][t]) + 0.333333343267440796f * d; }
                                else { a0[hf][t] = (((a0[hf][t] + ka1[hf][t]) + ka2[hf][t]) + ky1[hf][t]) + d; }
                            } else if constexpr (METHOD == NCDE_MIDPOINT) {
                                if (j == 0) { ka1[hf][t] = d; as_[hf][t] = 0.5f * d; }
                                else { a0[hf][t] = (a0[hf][t] + ka1[hf][t]) + d; }
                            } else {
                                a0[hf][t] = a0[hf][t] + d;
                            }
                            if (j == S - 1) {
                                if (at_knot) a0[hf][t] += gk[hf][t];
                                as_[hf][t] = METHOD == NCDE_RK4_38 ? a0[hf][t] * 0.125f : a0[hf][t];
                            }
                            ys[hf][t] = znext[hf][t];
                        }
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            if constexpr (PLAN != 0) {
                                bool last;
                                ys[hf][t] = StageCombine::apply(METHOD, j, -kout[hf][t], dt, y0[hf][t], ky1[hf][t], ky2[hf][t], last);
                                as_[hf][t] = StageCombine::apply(METHOD, j, vy[hf][t], dt, a0[hf][t], ka1[hf][t], ka2[hf][t], last);
                            } else {
                                ys[hf][t] = Combine<METHOD>::apply(j, -kout[hf][t], y0[hf][t], ky1[hf][t], ky2[hf][t]);
                                as_[hf][t] = Combine<METHOD>::apply(j, vy[hf][t], a0[hf][t], ka1[hf][t], ka2[hf][t]);
                            }
                            if (at_knot) {
                                if (reset_y) {      // reset y to the stored value, add dL/dz of that output time
                                    y0[hf][t] = zk[hf][t];
                                    ys[hf][t] = y0[hf][t];
                                }
                                a0[hf][t] += gk[hf][t];
                                as_[hf][t] = a0[hf][t];
                            }
                        }
                    }
                }
                u32x4 piece[NS];
#pragma unroll
                for (int hf = 0; hf < NS; ++hf) piece[hf] = make_piece<HPF>(ys[hf], mx);
#pragma unroll
                for (int hf = 0; hf < NS; ++hf) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) zimg[((zp ^ 1) * 64 + 16 * q + 4 * t + g) * RS + 16 * hf + s] = ys[hf][t];
                    *reinterpret_cast<u32x4*>(exch + ((p & 1) * NS + hf) * 1024 + q * 256 + lane * 4) = piece[hf];
                }
                if constexpr (PLAN != 0) {
                    if (j == S - 1 && n > 1) plan_store((rs + 1) & 1);
                } else {
                    if (j == S - 1 && n - 3 >= 0) stage_store(n - 3);
                }
                zp ^= 1;
                F64_SYNC(4)
            }
        }
    }
    // ---- results ---------------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int hf = 0; hf < NS; ++hf)
        if (valid[hf]) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (16 * q + 4 * t + g < Hr) a.grad_z0[(long long)(b0 + 16 * hf + s) * Hr + 16 * q + 4 * t + g] = a0[hf][t];
        }
    float* gp = a.gpart + (long long)blockIdx.x * a.theta_size;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r < C) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) gp[a.gWo_off + ((16 * q + 4 * t + g) * C + r) * 64 + 16 * ct + s] = gWo[t][ct][r];
            }
            float v = gbo[t][r];
            v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
            if (s == 0 && r < C) gp[a.gbo_off + (16 * q + 4 * t + g) * C + r] = v;
        }
    }
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        if (sl == 1 && !has_inner) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) gp[a.gW_off[sl] + (16 * q + 4 * g + r) * 64 + 16 * ct + s] = gW[sl][ct][r];
            float v = gb[sl][r];
            v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
            if (s == 0) gp[a.gb_off[sl] + 16 * q + 4 * g + r] = v;
        }
    }
#ifdef NCDE_F64_PROF
    if (lane == 0)
        for (int k = 0; k < 8; ++k) gp[q * 8 + k] = (float)prof[k] / (float)((a.T - 1) * S);
#endif
    if constexpr (HPF == 1) {
        if (a.fault != nullptr) {
            if (__builtin_amdgcn_ballot_w64(h2_range_fault(mx)) != 0 && lane == 0) *fault_s = 1;
            __syncthreads();
            if (tid == 0) a.fault[blockIdx.x] = *fault_s;
        }
    }
}

using F64Fn = void (*)(KArgs);

template <int NS, int HPF, int DISC>
F64Fn pick_method(int method) {
    if (method == NCDE_RK4_38) return ncde_adj_h64<NCDE_RK4_38, NS, HPF, DISC>;
    if (method == NCDE_MIDPOINT) return ncde_adj_h64<NCDE_MIDPOINT, NS, HPF, DISC>;
    return ncde_adj_h64<NCDE_EULER, NS, HPF, DISC>;
}
template <int HPF>
F64Fn pick_planned(int method) {      // general time axis: one sample tile per workgroup, continuous adjoint
    if (method == NCDE_RK4_38) return ncde_adj_h64<NCDE_RK4_38, 1, HPF, 0, 1>;
    if (method == NCDE_MIDPOINT) return ncde_adj_h64<NCDE_MIDPOINT, 1, HPF, 0, 1>;
    return ncde_adj_h64<NCDE_EULER, 1, HPF, 0, 1>;
}
F64Fn pick_kernel(int method, int ns, int hpf, bool disc) {
    if (disc) {
        if (ns == 2) return hpf == 1 ? pick_method<2, 1, 1>(method) : pick_method<2, 2, 1>(method);
        return hpf == 1 ? pick_method<1, 1, 1>(method) : pick_method<1, 2, 1>(method);
    }
    if (ns == 2) return hpf == 1 ? pick_method<2, 1, 0>(method) : pick_method<2, 2, 0>(method);
    return hpf == 1 ? pick_method<1, 1, 0>(method) : pick_method<1, 2, 0>(method);
}

size_t f64_lds_bytes(const NcdeProblem* p, int ns) {
    const int S = p->method == NCDE_RK4_38 ? 4 : (p->method == NCDE_MIDPOINT ? 2 : 1);
    return sizeof(float) * (size_t)f64_lds_floats(p->n_layers, ns, p->interp != NCDE_INTERP_LINEAR, p->output == NCDE_OUT_TIMES ? S : 0);
}

// sample tiles per workgroup.  Measured at cfg4 (B = 8192, MI355X): NS = 1 (512 workgroups, two rounds) 3.56 ms, NS = 2 (one workgroup
// per CU, the two tiles interleaved in every wave) 3.7 - 4.8 ms: with both tiles' operands live the 512-register file spills and
// the scheduler serialises the tiles again.  NS = 2 stays selectable (NCDE_FLAG_TILED_NS2; tested).
int f64_ns(const NcdeProblem* p) {
    int ns = 1;
    if ((p->flags & NCDE_FLAG_TILED_NS2) && p->output != NCDE_OUT_TIMES) ns = 2;
    if (ns == 2 && f64_lds_bytes(p, 2) > (size_t)kLdsLimit) ns = 1;
    return ns;
}
int f64_hpf(const NcdeProblem* p) { return (p->flags & (NCDE_FLAG_FP32_MFMA | NCDE_FLAG_SPLIT_BF16)) ? 2 : 1; }
int64_t f64_fault_bytes(int n_wg) { return ((int64_t)n_wg * 4 + 255) & ~(int64_t)255; }

}  // namespace

bool ncde_fast64_supported(const NcdeProblem* p, int pass) {
    if (pass != 1 && pass != 2) return false;
    if (p->hidden != 64 || p->channels > 4 || p->n_layers < 1 || p->n_layers > kF64MaxLayers) return false;
    if (p->field_kind != NCDE_FIELD_ORIGINAL || p->field_input != NCDE_INPUT_MATMUL) return false;
    if (p->output == NCDE_OUT_TIMES && pass != 1) return false;      // general time axis: the continuous adjoint (PLAN instances)
    if (p->flags & (NCDE_FLAG_ADJOINT_V1 | NCDE_FLAG_ADJOINT_V2 | NCDE_FLAG_ADJOINT_V4 | NCDE_FLAG_DEBUG_PROFILE)) return false;
    for (int l = 0; l < p->n_layers; ++l) {
        if (p->layer_in[l] != 64 || p->layer_out[l] != 64) return false;
        if (l >= 1 && (p->layer_W[l] != p->layer_W[1] || p->layer_b[l] != p->layer_b[1])) return false;
    }
    if (p->n_layers > 1 && (p->layer_W[1] == p->layer_W[0] || p->layer_b[1] == p->layer_b[0])) return false;
    return f64_lds_bytes(p, 1) <= (size_t)kLdsLimit;
}

const char* ncde_fast64_kernel_name(const NcdeProblem* p, int pass) {
    if (!ncde_fast64_supported(p, pass)) return nullptr;
    const int ns = f64_ns(p), hpf = f64_hpf(p);
    if (pass == 2) return hpf == 1 ? (ns == 2 ? "ncde_adj_h64<H64,HH64,NS2,in-sweep,fwd-side fp16x2 + fp32,discrete>" : "ncde_adj_h64<H64,HH64,NS1,in-sweep,fwd-side fp16x2 + fp32,discrete>")
                                   : (ns == 2 ? "ncde_adj_h64<H64,HH64,NS2,in-sweep,fp32,discrete>" : "ncde_adj_h64<H64,HH64,NS1,in-sweep,fp32,discrete>");
    return hpf == 1 ? (ns == 2 ? "ncde_adj_h64<H64,HH64,NS2,in-sweep,fwd-side fp16x2 + fp32>" : "ncde_adj_h64<H64,HH64,NS1,in-sweep,fwd-side fp16x2 + fp32>")
                    : (ns == 2 ? "ncde_adj_h64<H64,HH64,NS2,in-sweep,fp32>" : "ncde_adj_h64<H64,HH64,NS1,in-sweep,fp32>");
}

int64_t ncde_fast64_workspace_bytes(const NcdeProblem* p, int pass) {
    if (!ncde_fast64_supported(p, pass)) return NCDE_ERR_UNSUPPORTED;
    const Layout y = make_layout(p);
    const int n_wg = (p->batch + 16 * f64_ns(p) - 1) / (16 * f64_ns(p));
    return (int64_t)sizeof(float) * (int64_t)n_wg * (int64_t)y.theta_size + 256 + f64_fault_bytes(n_wg);
}

int ncde_fast64_adjoint(const NcdeProblem* p, const float* src, const float* grad_out, const NcdeGrads* g, void* ws, size_t ws_bytes,
                        hipStream_t st, bool main_kernel_only, bool discrete) {
    if (!ncde_fast64_supported(p, discrete ? 2 : 1)) return NCDE_ERR_UNSUPPORTED;
    if ((int64_t)ws_bytes < ncde_fast64_workspace_bytes(p, discrete ? 2 : 1)) return NCDE_ERR_WORKSPACE;
    const Layout y = make_layout(p);
    const int ns = f64_ns(p), hpf = f64_hpf(p);
    const int n_wg = (p->batch + 16 * ns - 1) / (16 * ns);
    KArgs a;
    fill_kargs(p, y, &a);
    a.grad_out = grad_out; a.grad_z0 = g->grad_z0;
    if (discrete) { a.stages = const_cast<float*>(src); a.discrete = 1; }
    else a.z_out = src;
    a.gpart = (float*)ws;
    a.fault = hpf == 1 ? reinterpret_cast<int*>(static_cast<char*>(ws) + sizeof(float) * (size_t)n_wg * y.theta_size + 256) : nullptr;
    const size_t lds = f64_lds_bytes(p, ns);
    const bool planned = p->output == NCDE_OUT_TIMES;
    F64Fn fn = planned ? (hpf == 1 ? pick_planned<1>(p->method) : pick_planned<2>(p->method)) : pick_kernel(p->method, ns, hpf, discrete);
    if (ncde_lds_optin((const void*)fn, lds) != hipSuccess) return NCDE_ERR_HIP;
    hipLaunchKernelGGL(fn, dim3(n_wg), dim3(256), lds, st, a);
    if (hipGetLastError() != hipSuccess) return NCDE_ERR_HIP;
    if (hpf == 1) {      // re-execution of range-faulted workgroups with fp32-input MFMA (normally none: every workgroup exits at once)
        F64Fn fx = planned ? pick_planned<2>(p->method) : pick_kernel(p->method, ns, 2, discrete);
        if (ncde_lds_optin((const void*)fx, lds) != hipSuccess) return NCDE_ERR_HIP;
        a.only_faulted = 1;
        hipLaunchKernelGGL(fx, dim3(n_wg), dim3(256), lds, st, a);
        if (hipGetLastError() != hipSuccess) return NCDE_ERR_HIP;
    }
    if (main_kernel_only) return NCDE_OK;
    return launch_reduce_partials(p, y, g, (const float*)ws, n_wg, st);
}
