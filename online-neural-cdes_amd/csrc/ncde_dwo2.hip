// Pass B of the batch-tiled backward for the records of the COOPERATIVE sweep (round 6): its own translation unit (seconds to build).
#include "ncde_common.h"
#include "ncde_bf3.h"
#define NCDE_COOP_NO_KERNELS
#include "ncde_coop.h"
#include "ncde_dwo2.h"

// ------------------------------------------------------------------------------------------------
// pass B for the records of the COOPERATIVE sweep (round 6): 2-way split-fp16, record stream shared through LDS
// ------------------------------------------------------------------------------------------------
// ncde_dwo_pair above re-reads every record once per PAIR of row tiles (320 readers at cfg5: 16 TB/s out of the L2, which is what bounds
// it), multiplies in 3-way split-bf16 (6 MFMAs per product) and turns every dP tile through an LDS patch.  Here:
//   * one workgroup = 4 waves (one per SIMD, 512 registers) x 4 row tiles of Wo each = 16 row tiles; the records of a (stage, sample-tile
//     PAIR) are fetched ONCE per workgroup -- global -> LDS directly (global_load_lds_dwordx4), double-buffered a pair-stage ahead -- and
//     read from LDS by the four waves: 40 readers of the record stream instead of 320;
//   * every product is 2-way split-fp16 (3 MFMAs): Wo from the sweep's packed register images (scaled by sw), x_L for P from record A
//     (the image the owners published: one power-of-two scale per sample), x_L^T for dWo from record B (one scale per tile, u_T), dP
//     scaled by sigma / u_T with ONE sigma per time window (the samples of a pair are the K dimension of the dWo products: a per-sample
//     scale would not factor out) -- the accumulators therefore hold sigma x the window's sum and are divided once, exactly, when the
//     window's partial is folded into the workspace;
//   * P is formed TRANSPOSED, P^T = x_L^T Wo^T (the operand registers of an MFMA are symmetric in A / B): its D layout -- lane = (row of
//     the tile, four samples) -- IS the A operand of dWo = dP x_L^T, so dP goes from the tanh epilogue straight into the next MFMA.
// Exactness of the scaling: powers of two throughout.  An entry far below its window's largest bound keeps an absolute error of 2^-36 of
// that bound -- below the fp32 rounding of the sum it enters.  Same job, same result layout (gpartB[part][theta_o]) as ncde_dwo_pair.
template <int NRT>
__device__ __forceinline__ void dwo_h2_body(const KArgs& a, int n_sc, int n_st, float* gpartB) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int dlast = 128, PK = 8, NCH = 4, NWV = 16 / NRT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int C = a.C, H = a.H, ncq = C >> 2;
    // grid: 1-D, row groups x parts workgroups.  The `nrg` workgroups that walk the SAME sample-tile pairs (one part) should share an L2:
    // consecutive workgroup ids go round the 8 XCDs, so with parts a multiple of 8 the id is read as (XCD, slot on it) and every part
    // lives on one XCD, its row groups in consecutive slots (they stream the same records at about the same time: fetched from the
    // Infinity Cache once per XCD instead of once per row group)
    const int n_pair = n_st >> 1, parts = a.dw2_parts, nrg = gridDim.x / parts;
    int bx_, by_;
    if ((parts & 7) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        by_ = xcd * (parts >> 3) + slot / nrg;
        bx_ = slot % nrg;
    } else {
        by_ = blockIdx.x / nrg;
        bx_ = blockIdx.x % nrg;
    }
    const int bx = __builtin_amdgcn_readfirstlane(bx_), by = __builtin_amdgcn_readfirstlane(by_);
    // LDS: two buffers of { x_L images of the two tiles | x_L^T of the pair | w a of the two tiles | dX/dt of the two tiles | scales }
    const int nRC = H >> 4, nRD = (C + 15) >> 4;      // 1 KB chunks per tile of the w a / dX/dt records
    const int oXP = 2 * 2048, oRC = oXP + 4096, oRD = oRC + 2 * nRC * 256, oS = oRD + 2 * nRD * 256, per_buf = oS + 2 * 256;
    // the four row tiles of this wave: weights (from the sweep's packed images: member, P-role wave, fragment), bias, accumulators
    int hb[NRT], cq[NRT];
    u32x4 W[NRT][NCH][2];
    float bias[NRT];
    f32x4 gW[NRT][PK];
    float gb[NRT];
#pragma unroll
    for (int q = 0; q < NRT; ++q) {
        const int rt = (bx * NWV + wave) * NRT + q;
        hb[q] = rt / ncq;
        cq[q] = rt - hb[q] * ncq;
        const int mem = rt / COOP_RPM, rw = (rt % COOP_RPM) / COOP_NRT, qq = rt % COOP_NRT;
        const unsigned* wp = a.coop_img + (long long)mem * (coop_p_words() + coop_t_words()) + rw * (40 * 64 * 4);
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) W[q][c][pc] = *reinterpret_cast<const u32x4*>(wp + ((((qq * 4 + c) * 2 + pc) * 64) + lane) * 4);
        bias[q] = a.bo[(4 * hb[q] + (li >> 2)) * C + 4 * cq[q] + (li & 3)];
#pragma unroll
        for (int jt = 0; jt < PK; ++jt) gW[q][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        gb[q] = 0.0f;
    }
    const float sigma = coop_pow2_scale(__uint_as_float(*a.win_max)), inv_sigma = coop_pow2_inv(sigma);
    const int my_n = by < n_pair ? (n_pair - by + parts - 1) / parts : 0;      // pairs by, by + parts, ...
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    // chunk ch (1 KB: one wave instruction) of the pair-stage (sc, pr) -> buffer `buf`; the waves take chunks wave, wave + NWV, ...
    // Where a chunk comes from is a table built ONCE per wave (float offsets from recA: constant part, stride per stage, stride per pair,
    // LDS offset), so that issuing a pair-stage's loads is straight-line scalar arithmetic -- a decision tree per chunk inside the
    // pair-stage loop cost 330 cycles per load instruction (measured: 2.6k of 9k cycles per pair-stage).
    const int n_chunk = 2 * 8 + 16 + 2 * nRC + 2 * nRD + 2;
#ifndef DW2_NIW
#define DW2_NIW 4
#endif
    // (only the first NIW waves fetch: with two waves per SIMD the older wave of each pair has the priority and ends every pair-stage
    // waiting at the barrier for the younger one -- the fetch instructions ride in that slack)
    constexpr int NIW = DW2_NIW < NWV ? DW2_NIW : NWV;
    constexpr int NSLOT = (60 + NIW - 1) / NIW;      // (H <= 128, C <= 80: at most 60 chunks)
    long long s_const[NSLOT];
    int s_sc[NSLOT], s_pr[NSLOT], s_lds[NSLOT];
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
        int k = wave + NIW * j;
        if (k >= n_chunk) k = wave < NIW ? wave : 0;      // (a slot past the end fetches this wave's first chunk again: harmless)
        long long cst;
        int ssc, spr, sl;
        if (k < 16) { cst = (long long)(k >> 3) * 2048 + (k & 7) * 256; ssc = n_st * 2048; spr = 2 * 2048; sl = k * 256; }
        else if ((k -= 16) < 16) { cst = (a.recB - a.recA) + (long long)k * 256; ssc = n_pair * 4096; spr = 4096; sl = oXP + k * 256; }
        else if ((k -= 16) < 2 * nRC) { const int t2 = k >= nRC ? 1 : 0, kk = k - t2 * nRC; cst = (a.recC - a.recA) + (long long)t2 * (H * 16) + kk * 256; ssc = n_st * H * 16; spr = 2 * H * 16; sl = oRC + k * 256; }
        else if ((k -= 2 * nRC) < 2 * nRD) { const int t2 = k >= nRD ? 1 : 0, kk = k - t2 * nRD; cst = (a.recD - a.recA) + (long long)t2 * (C * 16) + kk * 256; ssc = n_st * C * 16; spr = 2 * C * 16; sl = oRD + k * 256; }
        else { k -= 2 * nRD; cst = (a.recS - a.recA) + (long long)k * 32; ssc = n_st * 32; spr = 64; sl = oS + k * 256; }
        s_const[j] = cst; s_sc[j] = ssc; s_pr[j] = spr; s_lds[j] = sl;
    }
    // (the chunk offsets RUN: pair-stage after pair-stage a slot's offset grows by `parts` pairs, or -- at the last pair of a stage -- by
    // one stage minus the pairs walked: two scalar adds per chunk instead of two 64-bit multiply-adds)
    const int my_n_ = by < n_pair ? (n_pair - by + parts - 1) / parts : 0;
    long long s_off[NSLOT];
    int s_dpr[NSLOT], s_dsc[NSLOT];
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
        s_off[j] = s_const[j] + (long long)by * s_pr[j];
        s_dpr[j] = parts * s_pr[j];
        s_dsc[j] = s_sc[j] - (my_n_ - 1) * parts * s_pr[j];
    }
    auto issue = [&](bool wrap, int buf) {      // fetch the pair-stage the running offsets point at, then advance them (wrap: to the next stage)
        if (wave >= NIW) return;
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            __builtin_amdgcn_global_load_lds((gptr_t)(a.recA + s_off[j] + lane * 4), (lptr_t)(lds + buf * per_buf + s_lds[j]), 16, 0, 0);
            s_off[j] += wrap ? s_dsc[j] : s_dpr[j];
        }
    };
#ifdef NCDE_DW2_PROF
    unsigned long long dprof[5] = {0, 0, 0, 0, 0}, dlast_ = __builtin_readcyclecounter();      // issue | P + epilogue | dWo | DMA wait | barrier
#define DW2_TICK(k) { const unsigned long long n_ = __builtin_readcyclecounter(); dprof[k] += n_ - dlast_; dlast_ = n_; }
#else
#define DW2_TICK(k)
#endif
    u32x4 Ap[NRT][2];      // A operand of dWo: lane (row li, k-group lk) = samples 4 lk .. + 3 of tile a (dwords 0, 1), of tile b (2, 3)
    // P^T = x_L^T Wo^T of both tiles of the pair in buffer `buf`, tanh, dP (scaled by sigma / u_T, split) -> Ap; bias-gradient sums.
    // (every LDS operand is requested one use ahead)
    auto p_and_epilogue = [&](int buf, int lane, int li, int lk) {
        const float* B_ = lds + buf * per_buf;
        const unsigned* XS = reinterpret_cast<const unsigned*>(B_);
        auto ldx = [&](int t2, int c, int pc) { return *reinterpret_cast<const u32x4*>(XS + t2 * 2048 + ((c * 2 + pc) * 64 + lane) * 4); };
        constexpr bool AHEAD = NRT == 4;      // one wave per SIMD: request every LDS operand one use ahead (two waves cover each other)
        u32x4 nx0, nx1;
        if constexpr (AHEAD) { nx0 = ldx(0, 0, 0); nx1 = ldx(0, 0, 1); }
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const float* S_ = B_ + oS + t2 * 256;
            const f32x4 isx4 = *reinterpret_cast<const f32x4*>(S_ + 4 * lk);
            f32x4 isx2;      // (powers of two times the tanh pre-scale: the same bits as scaling first)
#pragma unroll
            for (int r = 0; r < 4; ++r) isx2[r] = isx4[r] * NCDE_TANH_PRESCALE;
            const float fT = sigma * S_[17];      // sigma / u_T
            auto lda = [&](int i) { return *reinterpret_cast<const f32x4*>(B_ + oRC + t2 * nRC * 256 + (4 * hb[i] + (li >> 2)) * 16 + 4 * lk); };
            auto ldd = [&](int i) { return *reinterpret_cast<const f32x4*>(B_ + oRD + t2 * nRD * 256 + (4 * cq[i] + (li & 3)) * 16 + 4 * lk); };
            f32x4 na4, nd4;
            if constexpr (AHEAD) { na4 = lda(0); nd4 = ldd(0); }
            // (main and cross accumulator per tile; the order below keeps the two cross products of a tile 2 NRT MFMAs apart)
            f32x4 pm[NRT], px[NRT];
#pragma unroll
            for (int i = 0; i < NRT; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pm[i][r] = bias[i] * coop_pow2_inv(isx4[r]);      // sx sw: the bias joins the scaled accumulator exactly
                px[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                u32x4 x0, x1;
                if constexpr (AHEAD) {
                    x0 = nx0; x1 = nx1;
                    if (c + 1 < NCH) { nx0 = ldx(t2, c + 1, 0); nx1 = ldx(t2, c + 1, 1); }
                    else if (t2 == 0) { nx0 = ldx(1, 0, 0); nx1 = ldx(1, 0, 1); }
                } else { x0 = ldx(t2, c, 0); x1 = ldx(t2, c, 1); }
#pragma unroll
                for (int i = 0; i < NRT; ++i) px[i] = mfma_h(x0, W[i][c][1], px[i]);
#pragma unroll
                for (int i = 0; i < NRT; ++i) pm[i] = mfma_h(x0, W[i][c][0], pm[i]);
#pragma unroll
                for (int i = 0; i < NRT; ++i) px[i] = mfma_h(x1, W[i][c][0], px[i]);
            }
#pragma unroll
            for (int i = 0; i < NRT; ++i) {
                const f32x4 pc = h2_combine(pm[i], px[i]);
                f32x4 a4, d4;
                if constexpr (AHEAD) {
                    a4 = na4; d4 = nd4;
                    if (i + 1 < NRT) { na4 = lda(i + 1); nd4 = ldd(i + 1); }
                } else { a4 = lda(i); d4 = ldd(i); }
                float dps[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float th = tanh_prescaled(pc[r] * isx2[r]);
                    const float dp = (a4[r] * d4[r]) * (1.0f - th * th);
                    gb[i] += dp;
                    dps[r] = dp * fT;
                }
                unsigned h0, l0, h1, l1;
                coop_split2(dps[0], dps[1], h0, l0);
                coop_split2(dps[2], dps[3], h1, l1);
                Ap[i][0][2 * t2] = h0; Ap[i][0][2 * t2 + 1] = h1;
                Ap[i][1][2 * t2] = l0; Ap[i][1][2 * t2 + 1] = l1;
            }
        }
    };
    // dWo += dP x_L^T from Ap and the x_L^T of buffer `buf`: the 32 samples of the pair are K; main product into the accumulator, the two cross
    // products through a temporary that carries 2^11 and is folded at once (the accumulators live for the whole launch)
    auto dwo_products = [&](int buf, int lane) {
        const unsigned* XP = reinterpret_cast<const unsigned*>(lds + buf * per_buf + oXP);
        constexpr bool AHEAD = NRT == 4;
        auto ldb = [&](int jt, int pc) { return *reinterpret_cast<const u32x4*>(XP + ((jt * 2 + pc) * 64 + lane) * 4); };
        u32x4 nb0, nb1;
        if constexpr (AHEAD) { nb0 = ldb(0, 0); nb1 = ldb(0, 1); }
        f32x4 ptx[NRT];
#pragma unroll
        for (int jt = 0; jt < PK; ++jt) {
            u32x4 b0, b1;
            if constexpr (AHEAD) {
                b0 = nb0; b1 = nb1;
                if (jt + 1 < PK) { nb0 = ldb(jt + 1, 0); nb1 = ldb(jt + 1, 1); }
            } else { b0 = ldb(jt, 0); b1 = ldb(jt, 1); }
            f32x4 tx[NRT];      // (the two cross products of a tile are 2 NRT MFMAs apart: neither waits for the other)
#pragma unroll
            for (int i = 0; i < NRT; ++i) tx[i] = mfma_h(Ap[i][0], b1, (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int i = 0; i < NRT; ++i) gW[i][jt] = mfma_h(Ap[i][0], b0, gW[i][jt]);
#pragma unroll
            for (int i = 0; i < NRT; ++i) tx[i] = mfma_h(Ap[i][1], b0, tx[i]);
            if (jt > 0) {      // fold the previous column tile's cross products while this one's MFMAs run
#pragma unroll
                for (int i = 0; i < NRT; ++i) gW[i][jt - 1] = h2_combine(gW[i][jt - 1], ptx[i]);
            }
#pragma unroll
            for (int i = 0; i < NRT; ++i) ptx[i] = tx[i];
        }
#pragma unroll
        for (int i = 0; i < NRT; ++i) gW[i][PK - 1] = h2_combine(gW[i][PK - 1], ptx[i]);
    };
    // (Two waves share each SIMD at NRT = 2.  Running the upper half of the waves one pair-stage late -- so that one wave of a SIMD
    // multiplies while the other does its tanh / split work -- was built and measured SLOWER, 8.6k against 7.4k cycles per pair-stage:
    // with two row tiles a wave has two short dependent MFMA chains per column tile and is latency-bound on its own; in lock step the
    // two waves fill each other's gaps.)
    if (my_n > 0 && n_sc > 0) {
        const int nq = n_sc * my_n;
        int k_n = 0;      // index, within its stage, of the pair-stage to request next
        issue(k_n + 1 == my_n, 0);
        if (++k_n == my_n) k_n = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int q = 0; q < nq; ++q) {
            const int buf = q & 1;
            if (q + 1 < nq) {
                issue(k_n + 1 == my_n, buf ^ 1);
                if (++k_n == my_n) k_n = 0;
            }
            __builtin_amdgcn_sched_barrier(0);
            DW2_TICK(0)
            // (lane-derived offsets are re-derived from an opaque copy of the thread id every pair-stage: kept across the loop they were
            // spilled, and a scratch reload waits -- vmcnt is in order -- for the record loads issued just above)
            int t_ = tid;
            asm volatile("" : "+v"(t_));
            const int lane_ = t_ & 63, li_ = t_ & 15, lk_ = (t_ >> 4) & 3;
            p_and_epilogue(buf, lane_, li_, lk_);
            DW2_TICK(1)
            dwo_products(buf, lane_);
            DW2_TICK(2)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the next pair-stage's records: requested at the top)
            DW2_TICK(3)
            __syncthreads();
            DW2_TICK(4)
        }
#ifdef NCDE_DW2_PROF
        if (lane == 0)
            for (int k = 0; k < 5; ++k) a.grad_z0[((long long)blockIdx.x * NWV + wave) * 8 + k] = (float)dprof[k] / (float)nq;
#endif
    }
    // ---- this workgroup's 16 row tiles of the part's partial: sigma divided out exactly, added to what the earlier windows left ------
    const long long wo_sz = (long long)H * C * dlast, theta_o = wo_sz + (long long)H * C;
    float* gp = gpartB + (long long)by * theta_o;
#pragma unroll
    for (int q = 0; q < NRT; ++q) {
#pragma unroll
        for (int jt = 0; jt < PK; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {      // D row 4 lk + r of the tile = (state unit 4 hb + lk, channel 4 cq + r), column 16 jt + li
                float* dst = gp + ((long long)(4 * hb[q] + lk) * C + 4 * cq[q] + r) * dlast + 16 * jt + li;
                const float v = gW[q][jt][r] * inv_sigma;
                *dst = a.dw2_accum ? *dst + v : v;
            }
        float v = gb[q];      // this lane: row li of the tile, its samples; the other three k-groups hold the rest
        v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        if (lk == 0) {
            float* dst = gp + wo_sz + (4 * hb[q] + (li >> 2)) * C + 4 * cq[q] + (li & 3);
            *dst = a.dw2_accum ? *dst + v : v;
        }
    }
}

// NRT = 2 row tiles per wave x 8 waves (two per SIMD) = 16 row tiles per workgroup.  The other split -- 4 x 4, one wave per SIMD with the
// 512-register file, half the LDS operand reads -- was built and measured: 9.8k against 7.4k cycles per pair-stage (the compiler parks
// the weights in AGPRs and copies them, and nothing covers one wave's LDS / MFMA latencies), DESIGN.md section 5.5b.
extern "C" __global__ __launch_bounds__(512) void ncde_dwo_h2(KArgs a, int n_sc, int n_st, float* gpartB) { dwo_h2_body<2>(a, n_sc, n_st, gpartB); }
