// Register-resident adjoint, version 4: the two dependency chains of a reverse stage run in DIFFERENT waves.
//
// A stage of the continuous adjoint (adjoint.py:73-106, one evaluation of augmented_dynamics) is two chains that only
// touch at a hand-over:
//   * the re-integration of y -- a forward stage: hidden layers -> P = Wo x_L + bo -> tanh -> f = tanh(P) . dX -> Butcher step
//     of y.  It depends on y alone.
//   * the cotangent chain -- dP = a (x) dX * (1 - tanh^2 P) -> dL/dx_L = Wo^T dP -> hidden layers backwards -> a^T df/dy ->
//     Butcher step of a -- and the parameter-gradient accumulation.  It depends on a and on what the forward stage produced
//     (1 - tanh^2 P, the ReLU masks, the layer inputs).
// ncde_adj_fast3 ran both in one "chain wave": stage time = their SUM (9.9k of 13.0k cycles).  Here a workgroup (one
// 16-sample tile, 8 waves, two per SIMD) is four PAIRS, pair p owning h-blocks 2p, 2p+1 (160 rows of Wo):
//   * Y wave p (waves 0-3) does the forward stage of reverse stage sc+1 while
//   * A wave p (waves 4-7) does the cotangent chain and ALL gradient accumulation of stage sc,
// so the stage time is max(forward stage, cotangent stage), one workgroup barrier per stage.
// Hand-over Y -> A, per stage and pair: t = 4 dX r (1 - r) with r = 1/(exp(2P) + 1) for the pair's 10 output tiles in the
// MFMA D-register layout (40 floats per lane through LDS, block = one channel quad, consumed-flags A -> Y so the buffer is
// single), one dword of ReLU mask bits per lane, and the [unit][sample] images of (z, x_1..x_L) for the weight gradients.
// Every GEMM is 3-way split-bf16 (ncde_bf3.h), including dL/dx_L = Wo^T dP: dP = a * t is ALREADY the B operand
// (lane (s, g) holds rows 4g + r of both tiles of a block = k index 8g + 4nb + r), Wo^T lives as split A fragments in the A
// wave (hi / mid in registers, lo in LDS).  dWo: dP goes through a wave-private 2 KB LDS patch to come back with the samples
// as K (32x32x16 blocks, 80 accumulators per lane for the whole solve); hidden-layer dW: fp32 MFMA from a private patch.
// DISC = 1: exact discrete backward (ncde_backward): stage inputs come from the forward's record, cotangent bookkeeping is the
// transpose of the Butcher step.
#include "ncde_fast4.h"

#include <type_traits>

#include "ncde_common.h"
#include "ncde_bf3.h"
#include "ncde_fastdefs.h"

namespace {

template <int NL, int C, int INTERP>
struct F4Lds {   // offsets in 4-byte words
    static constexpr int H = 32, HH = 32, NW = 4, HT = 2;
    static constexpr int CP = (C + 3) & ~3, CQ = CP / 4, NB = 2, NTILE = NB * CQ;
    static constexpr int DXW = INTERP == NCDE_INTERP_LINEAR ? CP : 3 * CP;
    static constexpr int XROWS = H + NL * HH;
    static constexpr int zx = 0;                               // [2][H*16]            stage input of y, by stage parity
    static constexpr int dxs = zx + 2 * H * 16;                // [3][16*DXW]          control-path ring
    static constexpr int scr = dxs + 3 * 16 * DXW;             // [NW][32][16]         dP block, [row][sample] (A wave private);
                                                               //   later in the stage the same 2 KB hold the pair's dL/dx_L partial
    static constexpr int boL = scr + NW * 512;                 // [NW][NTILE][4 g][4 r]
    static constexpr int rbuf = boL + NW * NTILE * 16;         // [NW][CQ][2][64][4]   t blocks Y -> A
    static constexpr int ximg = rbuf + NW * CQ * 512;          // [2][XROWS][16]       by stage parity
    static constexpr int gscr = ximg + 2 * XROWS * 16;         // [NW][16][16]         dL/dpre rows of the pair's dW tile (private)
    static constexpr int masks = gscr + NW * 256;              // [2][NW][64]          ReLU mask bits, by stage parity
    static constexpr int flags = masks + 2 * NW * 64;          // [16]                 0..3 reduction arrivals, 4..7 consumed counters
    static constexpr int biasL = flags + 16;                   // [2][HT][4 g][4 r]
    static constexpr int w1T3 = biasL + 2 * HT * 16;           // [HT][3][64][4]       split W1^T
    static constexpr int w0T3 = w1T3 + HT * 3 * 256;           // [2][3][64][4]        split W0^T, tile q = state rows 16q..16q+15
    static constexpr int woTLo = w0T3 + 2 * 3 * 256;           // [NW][CQ][HT][64][4]  lo pieces of Wo^T
    static constexpr int w1S3 = woTLo + NW * CQ * HT * 256;    // [2][HT][3][64][4]    split W0 / W1 (forward)
    static constexpr int total = w1S3 + 2 * HT * 3 * 256;
};

template <int NL, int C, int INTERP, int METHOD, int PROF = 0, int DISC = 0>
#ifndef F4_YB
#define F4_YB 2
#endif
#ifndef F4_LB
#define F4_LB 2
#endif
#ifdef F4_NV
__attribute__((amdgpu_num_vgpr(F4_NV)))
#endif
__global__ __launch_bounds__(512, F4_LB) void ncde_adj_fast4(KArgs a) {
    unsigned long long prof[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;
#define NCDE_TICK(k)                                                \
    if constexpr (PROF != 0) {                                      \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        prof[k] += now_ - tlast;                                    \
        tlast = now_;                                               \
    }
    using L = F4Lds<NL, C, INTERP>;
    constexpr int H = 32, HH = 32, NW = 4, HT = 2;
    constexpr int CP = L::CP, CQ = L::CQ, NB = 2, NTILE = L::NTILE, DXW = L::DXW, XROWS = L::XROWS;
    constexpr int S = kStages<METHOD>;
    constexpr int YB = CQ >= 3 ? F4_YB : 0;                // dWo blocks cq >= CQ - YB are accumulated by the Y wave (register balance: §fast4)
    constexpr int AB = CQ - YB;
    constexpr int NTY = 64 * NW;                       // threads of the Y role (they stage the control path)
    constexpr int EPT = (16 * DXW + NTY - 1) / NTY;
    static_assert(NL >= 1 && NL <= 3, "mask word holds three layers");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* zx = lds + L::zx;
    float* dxs = lds + L::dxs;
    float* scr = lds + L::scr;
    float* boL = lds + L::boL;
    float* rbuf = lds + L::rbuf;
    float* ximg = lds + L::ximg;
    float* gscr = lds + L::gscr;
    unsigned* masks = reinterpret_cast<unsigned*>(lds + L::masks);
    int* flags = reinterpret_cast<int*>(lds + L::flags);
    float* biasL = lds + L::biasL;
    unsigned* w1T3 = reinterpret_cast<unsigned*>(lds + L::w1T3);
    unsigned* w0T3 = reinterpret_cast<unsigned*>(lds + L::w0T3);
    unsigned* woTLo = reinterpret_cast<unsigned*>(lds + L::woTLo);
    unsigned* w1S3 = reinterpret_cast<unsigned*>(lds + L::w1S3);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifndef F4_FORCE_ROLE
#define F4_FORCE_ROLE 0   // development: 1 / 2 compile only the Y / A role (register accounting per role)
#endif
    const bool is_y = F4_FORCE_ROLE == 1 ? true : (F4_FORCE_ROLE == 2 ? false : wave < NW);
    const int pw = wave & (NW - 1);
    const int s = lane & 15, g = lane >> 4;
    const int b0 = blockIdx.x * NCDE_TILE;
    const int bs = b0 + s;
    const bool valid = bs < a.B;
    volatile __attribute__((address_space(3))) int* vflags = (volatile __attribute__((address_space(3))) int*)flags;
    const int NS = (a.T - 1) * S;                      // reverse stages; the loop runs NS + 1 intervals (Y one stage ahead of A)

    // ---- shared images ---------------------------------------------------------------------------------------------------
    if (tid < 16) flags[tid] = tid >= 4 && tid < 8 ? 8 : 0;   // consumed counters start "everything of stage 0 consumed"
    for (int e = tid; e < NW * NTILE * 16; e += 512) {
        const int r = e & 3, gg = (e >> 2) & 3, rest = e >> 4;
        const int t2 = rest % NTILE, wv = rest / NTILE;
        const int nb = t2 / CQ, cq = t2 - nb * CQ;
        const int h = 4 * (wv * NB + nb) + gg, c = 4 * cq + r;
        boL[e] = c < C ? NCDE_TANH_PRESCALE * a.bo[h * C + c] : 0.0f;
    }
    for (int e = tid; e < 2 * HT * 16; e += 512) {
        const int r = e & 3, gg = (e >> 2) & 3, t = (e >> 4) % HT, layer = e / (16 * HT);
        biasL[e] = a.b[layer < a.n_layers ? layer : 0][8 * gg + 4 * t + r];
    }
    if (wave < 2) {               // split W1^T: A row i <-> output unit 8(i>>2)+4t+(i&3), k = 8kg + jj
        const int t = wave;
        const int unit_out = 8 * (s >> 2) + 4 * t + (s & 3);
        float tmp[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) tmp[jj] = NL > 1 ? a.W[NL > 1 ? 1 : 0][(8 * g + jj) * HH + unit_out] : 0.0f;
        const Split3 sp = split8(tmp);
        *reinterpret_cast<u32x4*>(w1T3 + ((t * 3 + 0) * 64 + lane) * 4) = sp.hi;
        *reinterpret_cast<u32x4*>(w1T3 + ((t * 3 + 1) * 64 + lane) * 4) = sp.mid;
        *reinterpret_cast<u32x4*>(w1T3 + ((t * 3 + 2) * 64 + lane) * 4) = sp.lo;
    } else if (wave < 4) {        // split W0^T: tile q, A row i <-> state row h = 16q + 4(i&3) + (i>>2), k = hidden unit 8kg + jj
        const int q = wave - 2;
        const int hrow = 16 * q + 4 * (s & 3) + (s >> 2);
        float tmp[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) tmp[jj] = a.W[0][(8 * g + jj) * H + hrow];
        const Split3 sp = split8(tmp);
        *reinterpret_cast<u32x4*>(w0T3 + ((q * 3 + 0) * 64 + lane) * 4) = sp.hi;
        *reinterpret_cast<u32x4*>(w0T3 + ((q * 3 + 1) * 64 + lane) * 4) = sp.mid;
        *reinterpret_cast<u32x4*>(w0T3 + ((q * 3 + 2) * 64 + lane) * 4) = sp.lo;
    } else if (wave < 6) {        // split W0 and W1 (forward), A row i <-> unit 8(i>>2)+4t+(i&3), k = input 8kg + jj
        const int t = wave - 4;
        const int unitA = 8 * (s >> 2) + 4 * t + (s & 3);
#pragma unroll
        for (int layer = 0; layer < 2; ++layer) {
            float tmp[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) tmp[jj] = (layer == 0 || NL > 1) ? a.W[layer][unitA * HH + 8 * g + jj] : 0.0f;   // H == HH
            const Split3 sp = split8(tmp);
            *reinterpret_cast<u32x4*>(w1S3 + (((layer * HT + t) * 3 + 0) * 64 + lane) * 4) = sp.hi;
            *reinterpret_cast<u32x4*>(w1S3 + (((layer * HT + t) * 3 + 1) * 64 + lane) * 4) = sp.mid;
            *reinterpret_cast<u32x4*>(w1S3 + (((layer * HT + t) * 3 + 2) * 64 + lane) * 4) = sp.lo;
        }
    }

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // write-out of one row tile of a dWo block: D tile lane (unit n = s, g) register r <-> row 4g + r of row tile nb <->
    // (h = 4 (pw NB + nb) + g, c = 4 blk + r); bias: lane (row s of the tile, g) holds the sum over samples 4g..4g+3
    auto write_dwo_tiles = [&](float* gp, int blk, int nb, const f32x4* acc, float bsum) {
        const int h = 4 * (pw * NB + nb) + g;
#pragma unroll
        for (int ut = 0; ut < HT; ++ut)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * blk + r < C) gp[a.gWo_off + (h * C + 4 * blk + r) * HH + 16 * ut + s] = acc[ut][r];
        float v = bsum;
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        const int hrow = 4 * (pw * NB + nb) + (s >> 2), crow = 4 * blk + (s & 3);
        if (g == 0 && crow < C) gp[a.gbo_off + hrow * C + crow] = v;
    };
    if (is_y) {
        // =================================================================================================
        // Y wave: forward stage of reverse stage sc = it + 1
        // =================================================================================================
        u32x4 woHi[NB][CQ], woMid[NB][CQ], woLo[NB][CQ];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int cq = 0; cq < CQ; ++cq) {
                const int hA = 4 * (pw * NB + nb) + (s >> 2), cA = 4 * cq + (s & 3);
                float tmp[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) tmp[jj] = cA < C ? NCDE_TANH_PRESCALE * a.Wo[(hA * C + cA) * HH + 8 * g + jj] : 0.0f;
                const Split3 sp = split8(tmp);
                woHi[nb][cq] = sp.hi;
                woMid[nb][cq] = sp.mid;
                woLo[nb][cq] = sp.lo;
            }
        auto fwd_weights = [&](int layer, int tt) {
            Split3 As;
            As.hi = *reinterpret_cast<const u32x4*>(w1S3 + (((layer * HT + tt) * 3 + 0) * 64 + lane) * 4);
            As.mid = *reinterpret_cast<const u32x4*>(w1S3 + (((layer * HT + tt) * 3 + 1) * 64 + lane) * 4);
            As.lo = *reinterpret_cast<const u32x4*>(w1S3 + (((layer * HT + tt) * 3 + 2) * 64 + lane) * 4);
            return As;
        };
        // control-path staging (reverse order), by the 256 Y threads
        const float* eptr[EPT];
        float eprev[EPT], enext[EPT];
        bool eok[EPT];
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = tid + q * NTY;
            const int es = e / DXW, ec = e - es * DXW;
            const int part = ec / CP, c = ec - part * CP;
            eok[q] = e < 16 * DXW && c < C && (b0 + es) < a.B;
            const long long base = (long long)(eok[q] ? b0 + es : 0) * a.cs_b;
            eptr[q] = a.coeffs + base + (INTERP == NCDE_INTERP_LINEAR ? c : (part + 1) * C + c);
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
                const int e = tid + q * NTY;
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
        f32x4 gWoY[YB > 0 ? YB : 1][NB][HT];
        float gboY[YB > 0 ? YB : 1][NB];
#pragma unroll
        for (int i = 0; i < (YB > 0 ? YB : 1); ++i)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                gboY[i][nb] = 0.0f;
#pragma unroll
                for (int ut = 0; ut < HT; ++ut) gWoY[i][nb][ut] = zero4;
            }
        float wprev = 0.0f;     // quadrature weight of the stage whose dP blocks come back in this interval
        const int last_row = a.n_out - 1;
        float y0[NB], ky1[NB], ky2[NB], zreg[8];
        f32x4 znext[2];     // DISC: the next stage input, fetched a stage ahead; KNOTS: the stored state of the next knot
        float yk[NB];
        auto rec_fetch = [&](int lin) {
            const float* rp = a.stages + ((long long)lin * a.B + (valid ? bs : 0)) * H + 8 * g;
            znext[0] = *reinterpret_cast<const f32x4*>(rp);
            znext[1] = *reinterpret_cast<const f32x4*>(rp + 4);
        };
        if constexpr (DISC != 0) {
            rec_fetch((a.T - 1) * S - 1);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) zreg[jj] = valid ? znext[jj >> 2][jj & 3] : 0.0f;
        } else {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) zreg[jj] = valid ? a.z_out[((long long)bs * a.n_out + last_row) * H + 8 * g + jj] : 0.0f;
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const long long o = ((long long)bs * a.n_out + last_row) * H + 4 * (pw * NB + nb) + g;
            y0[nb] = (DISC == 0 && valid) ? a.z_out[o] : 0.0f;
            ky1[nb] = ky2[nb] = 0.0f;
            yk[nb] = 0.0f;
        }
        __syncthreads();

        int n = a.T - 1, j = 0;
        if constexpr (PROF != 0) tlast = __builtin_readcyclecounter();
        for (int it = 0; it <= NS; ++it) {
            const bool fwd = it < NS;       // forward stage sc = it + 1
            const int sc = it + 1;
            const int par = sc & 1;
            float t = 0.0f, frac = 0.0f;
            int idx = 0;
            Split3 xb;
            if (fwd) {
                if (j == 0) {
                    if (n - 3 >= 0) stage_load(n - 3);
                    if constexpr (DISC == 0) {
                        if (a.output == NCDE_OUT_KNOTS) {      // the stored knot value this step ends on
                            const float* rp = a.z_out + ((long long)(valid ? bs : 0) * a.n_out + (n - 1)) * H;
                            znext[0] = *reinterpret_cast<const f32x4*>(rp + 8 * g);
                            znext[1] = *reinterpret_cast<const f32x4*>(rp + 8 * g + 4);
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb) yk[nb] = rp[4 * (pw * NB + nb) + g];
                        }
                    }
                }
                t = DISC != 0 ? (float)(n - 1) + stage_offset(METHOD, S - 1 - j) : -(-(float)n + stage_offset(METHOD, j));
                idx = piece_index(t, a.n_pieces);
                frac = t - (float)idx;
                if constexpr (DISC != 0) {
                    const int lin = (n - 1) * S + (S - 1 - j);
                    if (lin >= 1) rec_fetch(lin - 1);
                }
                // ---- hidden layers (split-bf16); x[l][4t+r] <-> unit 8g + 4t + r ---------------------------------------
                float x[NL][8];
                {
                    f32x4 acc[HT];
                    xb = split8(zreg);
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) acc[tt] = mfma_split(fwd_weights(0, tt), xb, *reinterpret_cast<const f32x4*>(biasL + (tt * 4 + g) * 4));
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) x[0][4 * tt + r] = relu_bits(acc[tt][r]);
#pragma unroll
                    for (int l = 1; l < NL; ++l) {
                        xb = split8(x[l - 1]);
#pragma unroll
                        for (int tt = 0; tt < HT; ++tt) acc[tt] = mfma_split(fwd_weights(1, tt), xb, *reinterpret_cast<const f32x4*>(biasL + ((HT + tt) * 4 + g) * 4));
#pragma unroll
                        for (int tt = 0; tt < HT; ++tt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) x[l][4 * tt + r] = relu_bits(acc[tt][r]);
                    }
                    xb = split8(x[NL - 1]);
                }
                {   // ReLU masks for the pair's A wave: bit 8l + jj <-> x_l[8g + jj] > 0 (relu output: positive <=> bits != 0)
                    unsigned m = 0;
#pragma unroll
                    for (int l = 0; l < NL; ++l)
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) {
                            const unsigned bit = __builtin_bit_cast(unsigned, x[l][jj]) != 0u ? 1u : 0u;
                            m |= bit << (8 * l + jj);
                        }
                    masks[(par * NW + pw) * 64 + lane] = m;
                }
                {  // [unit][sample] images for the weight gradients: Y waves hold identical copies, wave pw writes image pw.
                   // Written (and consumed by the A waves) on EVERY stage: a stage of quadrature weight 0 (midpoint, j = 0)
                   // contributes exact zeros -- no stage-dependent control flow in either role.
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
                        }
                }
                NCDE_TICK(0)
            }
            const float* dxp = dxs + (idx % 3) * 16 * DXW + s * DXW;
            f32x4 bx[HT];       // w * x_L of stage it as the fp32 B operand of the dWo tiles this wave accumulates
            if constexpr (YB > 0) {
                const float* xp = ximg + (it & 1) * XROWS * 16;
#pragma unroll
                for (int ut = 0; ut < HT; ++ut) {
                    bx[ut] = *reinterpret_cast<const f32x4*>(xp + (H + (NL - 1) * HH + 16 * ut + s) * 16 + 4 * g);
#pragma unroll
                    for (int q = 0; q < 4; ++q) bx[ut][q] *= wprev;
                }
            }
            // ---- output tiles: P, r = 1/(exp(2P)+1), f, t = 4 dX r (1 - r) -> LDS block; dP blocks of stage it coming back ----
            float kout[NB];
            float sdx = 0.0f;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) kout[nb] = 0.0f;
#pragma unroll
            for (int cq = 0; cq < CQ; ++cq) {
                f32x4 tv[NB];
                if (fwd) {
                    f32x4 o[NB];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        Split3 As;
                        As.hi = woHi[nb][cq];
                        As.mid = woMid[nb][cq];
                        As.lo = woLo[nb][cq];
                        o[nb] = mfma_split(As, xb, *reinterpret_cast<const f32x4*>(boL + ((pw * NTILE + nb * CQ + cq) * 4 + g) * 4));
                    }
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
                    for (int r = 0; r < 4; ++r) sdx += dx[r];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float rr = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(o[nb][r]) + 1.0f);
                            if constexpr (DISC == 0) kout[nb] = fmaf(rr, dx[r], kout[nb]);
                            tv[nb][r] = (4.0f * dx[r]) * fmaf(-rr, rr, rr);
                        }
                }
                // the pair's A wave must be done with block cq of stage it (its t values taken; for cq >= AB its dP tile put back)
                {
                    const int want = it * 8 + cq + 1;
                    while (__builtin_amdgcn_readfirstlane(vflags[4 + pw]) < want) __builtin_amdgcn_s_sleep(1);
                    wave_lds_order();
                }
                float* slot = rbuf + (pw * CQ + cq) * 512;
                if constexpr (YB > 0) {
                    if (cq >= AB && it >= 1) {     // dWo of block cq, stage it: the tile is [row][sample], samples are K (fp32 MFMA)
                        f32x4 av[NB];
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) av[nb] = *reinterpret_cast<const f32x4*>(slot + (nb * 16 + s) * 16 + 4 * g);
                        wave_lds_order();
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            gboY[cq >= AB ? cq - AB : 0][nb] += wprev * ((av[nb][0] + av[nb][1]) + (av[nb][2] + av[nb][3]));
#pragma unroll
                            for (int q = 0; q < 4; ++q)
#pragma unroll
                                for (int ut = 0; ut < HT; ++ut)
                                    gWoY[cq >= AB ? cq - AB : 0][nb][ut] = mfma16(av[nb][q], bx[ut][q], gWoY[cq >= AB ? cq - AB : 0][nb][ut]);
                        }
                    }
                }
                if (fwd) {
                    *reinterpret_cast<f32x4*>(slot + lane * 4) = tv[0];
                    *reinterpret_cast<f32x4*>(slot + 256 + lane * 4) = tv[1];
                }
#ifdef F4_YSB
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
            if (fwd) {
                NCDE_TICK(1)
                if constexpr (DISC == 0) {
                    float ys[NB];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        kout[nb] = fmaf(-2.0f, kout[nb], sdx);
                        ys[nb] = Combine<METHOD>::apply(j, -kout[nb], y0[nb], ky1[nb], ky2[nb]);
                    }
                    if (j == S - 1 && a.output == NCDE_OUT_KNOTS) {   // reset y to the stored knot value
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) y0[nb] = valid ? yk[nb] : 0.0f;
                    } else {
                        float* zw = zx + par * H * 16;
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) zw[(4 * (pw * NB + nb) + g) * 16 + s] = ys[nb];
                    }
                }
                if (j == S - 1 && n - 3 >= 0) stage_store(n - 3);
                wprev = DISC != 0 ? 1.0f : stage_weight(METHOD, j);
                NCDE_TICK(2)
            }
            __syncthreads();   // E: stage sc + 1 of y done, stage sc of a done
            NCDE_TICK(3)
            if (it < NS) {
                const int par = (it + 1) & 1;
                if constexpr (DISC != 0) {
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) zreg[jj] = valid ? znext[jj >> 2][jj & 3] : 0.0f;
                } else {
                    if (j == S - 1 && a.output == NCDE_OUT_KNOTS) {
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) zreg[jj] = valid ? znext[jj >> 2][jj & 3] : 0.0f;
                    } else {
                        const float* zw = zx + par * H * 16;
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) zreg[jj] = zw[(8 * g + jj) * 16 + s];
                    }
                }
                if (++j == S) { j = 0; --n; }
            }
        }
        if constexpr (PROF != 0) {
            if (lane == 0) {
                unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.out) + ((long long)blockIdx.x * 8 + wave) * 6;
                for (int k = 0; k < 6; ++k) dst[k] = prof[k];
            }
        }
        if constexpr (YB > 0) {
            float* gp = a.gpart + (long long)blockIdx.x * a.theta_size;
#pragma unroll
            for (int i = 0; i < YB; ++i)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) write_dwo_tiles(gp, AB + i, nb, gWoY[i][nb], gboY[i][nb]);
        }
    } else {
        // =================================================================================================
        // A wave: cotangent chain + parameter gradients of reverse stage sc = it
        // =================================================================================================
        u32x4 wtHi[CQ][HT], wtMid[CQ][HT];
        unsigned* my_lo = woTLo + pw * CQ * HT * 256;
#pragma unroll
        for (int cq = 0; cq < CQ; ++cq)
#pragma unroll
            for (int tt = 0; tt < HT; ++tt) {
                // A row i = s <-> output unit 8(i>>2) + 4tt + (i&3); k = 8g + jj <-> row (nb = jj>>2, g, r = jj&3) of block cq
                const int unit = 8 * (s >> 2) + 4 * tt + (s & 3);
                float tmp[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int h = 4 * (pw * NB + (jj >> 2)) + g, c = 4 * cq + (jj & 3);
                    tmp[jj] = c < C ? a.Wo[(h * C + c) * HH + unit] : 0.0f;
                }
                const Split3 sp = split8(tmp);
                wtHi[cq][tt] = sp.hi;
                wtMid[cq][tt] = sp.mid;
                *reinterpret_cast<u32x4*>(my_lo + ((cq * HT + tt) * 64 + lane) * 4) = sp.lo;
            }
        f32x4 gWo[AB][NB][HT];               // dWo of block cq < AB: row tile nb x unit tile ut, fp32 MFMA accumulators (samples are K)
        f32x4 gW1 = zero4, gW0 = zero4;      // one 16x16 tile of dW1 / dW0 per pair: tile (tr, tc) = (pw >> 1, pw & 1)
        float gbo[AB][NB], gb1 = 0.0f, gb0 = 0.0f;
#pragma unroll
        for (int i = 0; i < AB; ++i)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                gbo[i][nb] = 0.0f;
#pragma unroll
                for (int ut = 0; ut < HT; ++ut) gWo[i][nb][ut] = zero4;
            }
        const int last_row = a.n_out - 1;
        float a0[NB], ka1[NB], ka2[NB], ka3[NB], as_[NB], gk[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const long long o = ((long long)bs * a.n_out + last_row) * H + 4 * (pw * NB + nb) + g;
            a0[nb] = valid ? a.grad_out[o] : 0.0f;
            as_[nb] = (DISC != 0 && METHOD == NCDE_RK4_38) ? a0[nb] * 0.125f : a0[nb];
            ka1[nb] = ka2[nb] = ka3[nb] = 0.0f;
            gk[nb] = 0.0f;
        }
        const int tr = pw >> 1, tc = pw & 1;
        float* my_scr = scr + pw * 512;
        float* my_gs = gscr + pw * 256;
        __syncthreads();

        int n = a.T - 1, j = 0;
        if constexpr (PROF != 0) tlast = __builtin_readcyclecounter();
        for (int it = 0; it <= NS; ++it) {
            if (it >= 1) {
                const int sc = it;
                const int par = sc & 1;
                const float wq = DISC != 0 ? 1.0f : stage_weight(METHOD, j);
                const float* xi = ximg + par * XROWS * 16;
                if (j == 0) {      // cotangent of the knot / of t[0] this step ends on: fetched a step ahead of its use
                    const bool need = a.output == NCDE_OUT_KNOTS || n == 1;
                    const int row = a.output == NCDE_OUT_KNOTS ? n - 1 : 0;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        gk[nb] = (need && valid) ? a.grad_out[((long long)bs * a.n_out + row) * H + 4 * (pw * NB + nb) + g] : 0.0f;
                }
                const unsigned m = masks[(par * NW + pw) * 64 + lane];
                f32x4 bx[HT];       // w * x_L as the fp32 B operand of the dWo tiles: lane (unit 16 ut + s, kk = g) holds samples 4g..4g+3
#pragma unroll
                for (int ut = 0; ut < HT; ++ut) {
                    bx[ut] = *reinterpret_cast<const f32x4*>(xi + (H + (NL - 1) * HH + 16 * ut + s) * 16 + 4 * g);
#pragma unroll
                    for (int q = 0; q < 4; ++q) bx[ut][q] *= wq;
                }
                f32x4 accJ[HT];
#pragma unroll
                for (int tt = 0; tt < HT; ++tt) accJ[tt] = zero4;
#pragma unroll
                for (int cq = 0; cq < CQ; ++cq) {
                    const float* rb = rbuf + ((pw * CQ + cq) * 2) * 256 + lane * 4;
                    const f32x4 t0 = *reinterpret_cast<const f32x4*>(rb);
                    const f32x4 t1 = *reinterpret_cast<const f32x4*>(rb + 256);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the block is in registers before its slot is released
                    if (cq < AB) {
                        if (lane == 0) vflags[4 + pw] = sc * 8 + cq + 1;
                    }
                    float dP[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { dP[r] = as_[0] * t0[r]; dP[4 + r] = as_[1] * t1[r]; }
                    const Split3 gb = split8(dP);
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) {
                        Split3 As;
                        As.hi = wtHi[cq][tt];
                        As.mid = wtMid[cq][tt];
                        As.lo = *reinterpret_cast<const u32x4*>(my_lo + ((cq * HT + tt) * 64 + lane) * 4);
                        accJ[tt] = mfma_split(As, gb, accJ[tt]);
                    }
                    if (cq < AB) {   // dWo block cq: dP through the wave-private patch to come back with the samples as K; plain fp32
                        // MFMA (16 per block: no second split of dP, no split of x_L -- issue slots, not the matrix pipe, bound the stage)
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) my_scr[((jj >> 2) * 16 + 4 * g + (jj & 3)) * 16 + s] = dP[jj];
                        wave_lds_order();
                        f32x4 av[NB];
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) av[nb] = *reinterpret_cast<const f32x4*>(my_scr + (nb * 16 + s) * 16 + 4 * g);
                        wave_lds_order();
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            gbo[cq < AB ? cq : 0][nb] += wq * ((av[nb][0] + av[nb][1]) + (av[nb][2] + av[nb][3]));
#pragma unroll
                            for (int q = 0; q < 4; ++q)
#pragma unroll
                                for (int ut = 0; ut < HT; ++ut)
                                    gWo[cq < AB ? cq : 0][nb][ut] = mfma16(av[nb][q], bx[ut][q], gWo[cq < AB ? cq : 0][nb][ut]);
                        }
                    } else {         // blocks the Y wave accumulates: the [row][sample] tile goes back into the slot the t values came from
                        float* slot = rbuf + (pw * CQ + cq) * 512;
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) slot[((jj >> 2) * 16 + 4 * g + (jj & 3)) * 16 + s] = dP[jj];
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        if (lane == 0) vflags[4 + pw] = sc * 8 + cq + 1;
                    }
                    __builtin_amdgcn_sched_barrier(0);   // one block at a time: hoisting the next blocks' loads costs registers the role does not have
                }
                NCDE_TICK(0)
                // ---- dL/dx_L: sum of the four pairs' partials (the patch of the dP blocks is free now) ---------------------
                float* redb = scr;      // the dP patches are free now: they carry the pairs' dL/dx_L partials
                *reinterpret_cast<f32x4*>(redb + pw * 512 + lane * 4) = accJ[0];
                *reinterpret_cast<f32x4*>(redb + pw * 512 + 256 + lane * 4) = accJ[1];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the partial is in LDS before it is announced
                if (lane == 0) vflags[pw] = sc;
                while (__builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(vflags[lane & 3] == sc) != ~0ull ? 1 : 0))
                    __builtin_amdgcn_s_sleep(1);
                wave_lds_order();
                NCDE_TICK(1)
                float gpre[8];
                {
                    f32x4 v[HT];
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) {
                        v[tt] = *reinterpret_cast<const f32x4*>(redb + tt * 256 + lane * 4);
#pragma unroll
                        for (int wv = 1; wv < NW; ++wv) v[tt] += *reinterpret_cast<const f32x4*>(redb + wv * 512 + tt * 256 + lane * 4);
                    }
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const int keep = __builtin_amdgcn_sbfe((int)m, 8 * (NL - 1) + jj, 1);     // 0 or -1
                        const float vsum = v[jj >> 2][jj & 3];   // (bit_cast straight on a vector element reads element 0)
                        gpre[jj] = __builtin_bit_cast(float, __builtin_bit_cast(int, vsum) & keep);
                    }
                }
                // hidden-layer dW/db tile of this pair from dL/dpre of layer l (registers) and the layer's input image
                auto dw_tile = [&](int xrow0, f32x4& gW, float& gbias) {
                    if ((g >> 1) == tr) {
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) my_gs[(8 * (g & 1) + jj) * 16 + s] = gpre[jj];
                    }
                    wave_lds_order();
                    const f32x4 av = *reinterpret_cast<const f32x4*>(my_gs + s * 16 + 4 * g);
                    wave_lds_order();
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(xi + (xrow0 + 16 * tc + s) * 16 + 4 * g);
                    if (tc == 0) gbias += wq * ((av[0] + av[1]) + (av[2] + av[3]));
#pragma unroll
                    for (int q = 0; q < 4; ++q) gW = mfma16(av[q], wq * bv[q], gW);
                };
                // ---- hidden layers backward (split-bf16) -----------------------------------------------------------------
#pragma unroll
                for (int l = NL - 1; l >= 1; --l) {
                    dw_tile(H + (l - 1) * HH, gW1, gb1);
                    const Split3 gb = split8(gpre);
                    f32x4 acc[HT];
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) {
                        Split3 As;
                        As.hi = *reinterpret_cast<const u32x4*>(w1T3 + ((tt * 3 + 0) * 64 + lane) * 4);
                        As.mid = *reinterpret_cast<const u32x4*>(w1T3 + ((tt * 3 + 1) * 64 + lane) * 4);
                        As.lo = *reinterpret_cast<const u32x4*>(w1T3 + ((tt * 3 + 2) * 64 + lane) * 4);
                        acc[tt] = mfma_split(As, gb, zero4);
                    }
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const int keep = __builtin_amdgcn_sbfe((int)m, 8 * (l - 1) + jj, 1);
                        const float vacc = acc[jj >> 2][jj & 3];
                        gpre[jj] = __builtin_bit_cast(float, __builtin_bit_cast(int, vacc) & keep);
                    }
                }
                dw_tile(0, gW0, gb0);
                float vy[NB];
                {
                    const Split3 gb = split8(gpre);
                    Split3 As;
                    As.hi = *reinterpret_cast<const u32x4*>(w0T3 + ((tr * 3 + 0) * 64 + lane) * 4);
                    As.mid = *reinterpret_cast<const u32x4*>(w0T3 + ((tr * 3 + 1) * 64 + lane) * 4);
                    As.lo = *reinterpret_cast<const u32x4*>(w0T3 + ((tr * 3 + 2) * 64 + lane) * 4);
                    const f32x4 vv = mfma_split(As, gb, zero4);     // D row (g, r) <-> state row 16 tr + 4 r + g: r = 2 tc + nb
                    vy[0] = tc == 0 ? vv[0] : vv[2];
                    vy[1] = tc == 0 ? vv[1] : vv[3];
                }
                NCDE_TICK(2)
                if constexpr (DISC != 0) {
                    // transpose of the Butcher step (RK4 3/8: c4 = a/8; c3 = 3c4 + d4; c2 = 3c4 - d4 + d3;
                    // c1 = c4 + d4 - d3/3 + d2/3; a += d4 + d3 + d2 + d1), d = vy = dL/dY of this stage
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const float d = vy[nb];
                        if constexpr (METHOD == NCDE_RK4_38) {
                            const float c4 = a0[nb] * 0.125f;
                            if (j == 0) { ka1[nb] = d; as_[nb] = 3.0f * c4 + d; }
                            else if (j == 1) { ka2[nb] = d; as_[nb] = (3.0f * c4 - ka1[nb]) + d; }
                            else if (j == 2) { ka3[nb] = d; as_[nb] = ((c4 + ka1[nb]) - 0.333333343267440796f * ka2[nb]) + 0.333333343267440796f * d; }
                            else { a0[nb] = (((a0[nb] + ka1[nb]) + ka2[nb]) + ka3[nb]) + d; }
                        } else if constexpr (METHOD == NCDE_MIDPOINT) {
                            if (j == 0) { ka1[nb] = d; as_[nb] = 0.5f * d; }
                            else { a0[nb] = (a0[nb] + ka1[nb]) + d; }
                        } else {
                            a0[nb] = a0[nb] + d;
                        }
                    }
                    if (j == S - 1) {
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            a0[nb] += gk[nb];
                            as_[nb] = METHOD == NCDE_RK4_38 ? a0[nb] * 0.125f : a0[nb];
                        }
                    }
                } else {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) as_[nb] = Combine<METHOD>::apply(j, vy[nb], a0[nb], ka1[nb], ka2[nb]);
                    if (j == S - 1) {     // add dL/dz of the knot (every knot with sequence outputs, t[0] otherwise)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            a0[nb] += gk[nb];
                            as_[nb] = a0[nb];
                        }
                    }
                }
                NCDE_TICK(3)
                if (++j == S) { j = 0; --n; }
            }
            __syncthreads();   // E
            NCDE_TICK(4)
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
        // ---- write-out of this workgroup's parameter-gradient partial ------------------------------------------
        float* gp = a.gpart + (long long)blockIdx.x * a.theta_size;
#pragma unroll
        for (int blk = 0; blk < AB; ++blk)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) write_dwo_tiles(gp, blk, nb, gWo[blk][nb], gbo[blk][nb]);
        if constexpr (NL > 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) gp[a.gW_off[1] + (16 * tr + 4 * g + r) * HH + 16 * tc + s] = gW1[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) gp[a.gW_off[0] + (16 * tr + 4 * g + r) * H + 16 * tc + s] = gW0[r];
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
#undef NCDE_TICK
}

template <int NL, int C>
NcdeFast4Kernel pick4(int interp, int method, bool disc, bool prof) {
    if (prof) {
        if (interp == NCDE_INTERP_LINEAR && method == NCDE_RK4_38 && !disc) return ncde_adj_fast4<NL, C, NCDE_INTERP_LINEAR, NCDE_RK4_38, 1, 0>;
        return nullptr;
    }
#define NCDE_PICK(I, M)                                                                                  \
    if (interp == I && method == M) return disc ? ncde_adj_fast4<NL, C, I, M, 0, 1> : ncde_adj_fast4<NL, C, I, M, 0, 0>;
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_RK4_38)
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_MIDPOINT)
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_EULER)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_RK4_38)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_MIDPOINT)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_EULER)
#undef NCDE_PICK
    return nullptr;
}

}  // namespace

NcdeFast4Kernel ncde_fast4_pick(int n_layers, int channels, int interp, int method, bool discrete, bool profile) {
    if (n_layers == 3 && channels == 20) return pick4<3, 20>(interp, method, discrete, profile);
    return nullptr;
}

size_t ncde_fast4_lds_bytes(int n_layers, int channels, int interp) {
    if (n_layers == 3 && channels == 20)
    {
        size_t words = interp == NCDE_INTERP_LINEAR ? F4Lds<3, 20, NCDE_INTERP_LINEAR>::total : F4Lds<3, 20, NCDE_INTERP_CUBIC>::total;
        return sizeof(float) * words;
    }
    return 0;
}
