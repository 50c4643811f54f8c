// Adaptive dopri5, FUSED attempt kernels (round 4): one launch = one whole attempt of the step [t0, t0 + dt] -- the six stage
// evaluations on the register-resident field evaluation of the specialised family (ncde_fast.hip: weights held as split-bf16 MFMA
// operands, hidden layers register-to-register, output tiles + tanh + channel contraction per wave), the embedded error estimate, the
// tentative dense output -- and the CONTROLLER (batch-wide norm, accept / reject, next dt, stage descriptors of the next attempt;
// ncde_dp_defs.h) in the last workgroup to finish.  The per-launch kernels of ncde_adaptive.hip (six stage launches, a controller
// launch and a commit launch per attempt, state in HBM in between, the generic family's stage code) stay as the path for every shape
// these kernels do not cover.  Same algorithm, same controller code, same workspace:
//   /root/reference/modules/torchdiffeq/torchdiffeq/_impl/rk_common.py:41-86, 216-305; dopri5.py:5-36; misc.py:33-103; interp.py:4-61.
// What changes is where the data lives between stages (registers / LDS instead of HBM) and how an accepted step is committed: the
// state (y0, k1 = f(y0), FSAL) is double-buffered and the controller flips `ctrl->cur` on acceptance; the rows of the output that fall
// into the attempted step are written tentatively by the attempt itself (a rejected attempt's rows are rewritten by the accepted step
// that finally covers them -- launches are stream-ordered).
// Shapes: hidden, hidden_hidden <= 32 with C <= 20, or <= 64 with C <= 4 (zero-padded in registers: the weights are read with the real
// extents), layer 0 H -> HH and every further layer one shared HH -> HH (the reference's `[layer] * n` construction).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "ncde_adaptive_fast.h"
#include "ncde_bf3.h"
#include "ncde_dp_defs.h"
#include "ncde_fastdefs.h"

#ifdef NCDE_DPF_PROF      // development build (tools/build_dpfprof.sh): wall-clock stamps (10 ns) of workgroup 0 -> row 0 of sample 0 of `out`
#define DPF_STAMP(k) if (blockIdx.x == 0 && threadIdx.x == 0) stamps_[k] = wall_clock64();
#define DPA_TICK(k) if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); acc_[k] += now_ - last_; last_ = now_; }
#else
#define DPF_STAMP(k)
#define DPA_TICK(k)
#endif

namespace {

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// the last workgroup of the launch to get here runs the controller (threadFenceReduction pattern: every workgroup has published its
// partial sums before it takes a ticket)
__device__ __forceinline__ void dpf_finish_launch(const DpArgs& d, double* sh, int* sh_flags, float* sh_x, int* is_last, unsigned long long* stamps_ = nullptr) {
    // The partial sums were stored with 8-byte agent-scope atomic stores (write-through, `sc1`); once they are drained the ticket may
    // go out, and the last arriver reads them back with agent-scope atomic loads (dp_control_body) -- the "8-byte agent atomics on both
    // sides" form of MI355X_MICROARCH.md, instead of a __threadfence() per workgroup (7.4 us here: it writes the XCD's L2 back).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (stamps_ && blockIdx.x == 0 && threadIdx.x == 0) stamps_[6] = wall_clock64();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(&d.ctrl->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *is_last = t == gridDim.x - 1 ? 1 : 0;
    }
    __syncthreads();
    if (*is_last) {
        if (threadIdx.x == 0) __hip_atomic_store(&d.ctrl->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dp_control_body(d, sh, sh_flags, sh_x);
    }
}

}  // namespace

// Per-lane image of the weights: what lane `lane` of wave `wave` keeps in registers, in the order it loads it (fragments of 8 floats =
// one K chunk of one A-operand row, then the bias quadruples); real extents, zero beyond them.  Written ONCE per solve.
template <int H, int HH, int C>
struct DpfPack {
    static constexpr int NW = 4, CP = (C + 3) & ~3, CQ = CP / 4, HB = H / 4, HT = HH / 16, NB = HB / NW, KC0 = H / 32, KC = HH / 32;
    static constexpr int F8 = HT * KC0 + HT * KC + NB * CQ * KC, F4 = 2 * HT + NB * CQ;
    static constexpr int WAVE = F8 * 512 + F4 * 256;      // floats per wave
    static constexpr int TOTAL = NW * WAVE;
};

template <int H, int HH, int C>
__global__ __launch_bounds__(256) void ncde_dpf_pack(DpArgs d) {
    typedef DpfPack<H, HH, C> PK;
    constexpr int CQ = PK::CQ, HT = PK::HT, NB = PK::NB, KC0 = PK::KC0, KC = PK::KC;
    const KArgs& a = d.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = lane & 15, g = lane >> 4;
    const int Hr = a.H, HHr = a.dout[0], Cr = a.C;
    float* wp8 = d.WP + (long long)wave * PK::WAVE + lane * 8;
    float* wp4 = d.WP + (long long)wave * PK::WAVE + PK::F8 * 512 + lane * 4;
    auto put8 = [&](int f, const float* tmp) {
#pragma unroll
        for (int j = 0; j < 8; ++j) wp8[f * 512 + j] = tmp[j];
    };
    const bool has_inner = a.n_layers > 1;
#pragma unroll
    for (int t = 0; t < HT; ++t) {
        const int unitA = 32 * (t >> 1) + 8 * (s >> 2) + 4 * (t & 1) + (s & 3);
        float tmp[8];
#pragma unroll
        for (int cc = 0; cc < KC0; ++cc) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 32 * cc + 8 * g + j;
                tmp[j] = (unitA < HHr && k < Hr) ? a.W[0][unitA * Hr + k] : 0.0f;
            }
            put8(t * KC0 + cc, tmp);
        }
#pragma unroll
        for (int cc = 0; cc < KC; ++cc) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 32 * cc + 8 * g + j;
                tmp[j] = (has_inner && unitA < HHr && k < HHr) ? a.W[1][unitA * HHr + k] : 0.0f;
            }
            put8(HT * KC0 + t * KC + cc, tmp);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int unitD = 32 * (t >> 1) + 8 * g + 4 * (t & 1) + r;
            wp4[t * 256 + r] = unitD < HHr ? a.b[0][unitD] : 0.0f;
            wp4[(HT + t) * 256 + r] = (has_inner && unitD < HHr) ? a.b[1][unitD] : 0.0f;
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
            for (int cc = 0; cc < KC; ++cc) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 32 * cc + 8 * g + j;
                    tmp[j] = (hA < Hr && cA < Cr && k < HHr) ? NCDE_TANH_PRESCALE * a.Wo[(long long)(hA * Cr + cA) * HHr + k] : 0.0f;
                }
                put8(HT * KC0 + HT * KC + (nb * CQ + cq) * KC + cc, tmp);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int hh = 4 * hb + g, cc = 4 * cq + r;
                wp4[(2 * HT + nb * CQ + cq) * 256 + r] = (hh < Hr && cc < Cr) ? NCDE_TANH_PRESCALE * a.bo[hh * Cr + cc] : 0.0f;
            }
        }
    }

}

// ------------------------------------------------------------------------------------------------------------------
// forward attempt (phases: DP_INIT0 = f0 and the norms of the initial-step rule, DP_INIT1 = its probe evaluation, DP_STEP)
// ------------------------------------------------------------------------------------------------------------------
template <int H, int HH, int C>
__global__ __launch_bounds__(256, 1) void ncde_dpf_fwd(DpArgs d) {
    constexpr int NW = 4, NT = 64 * NW;
    constexpr int CP = (C + 3) & ~3, CQ = CP / 4, HB = H / 4, HT = HH / 16, NB = HB / NW;
    constexpr int KC0 = H / 32, KC = HH / 32;
    static_assert(H % (4 * NW) == 0 && HH % 32 == 0 && H % 32 == 0, "shape not tileable");
    __shared__ __attribute__((aligned(16))) float zx[2][H * 16];
    __shared__ __attribute__((aligned(16))) float dxq[6][16 * CP];
    __shared__ double red[4][NW];
    __shared__ double sh[256];
    __shared__ int sh_flags[2], is_last;
    __shared__ float sh_x;

    DpCtrl* c = d.ctrl;
    const int phase = c->phase;
    if (phase == DP_DONE || c->error != 0) return;      // (uniform)
#ifdef NCDE_DPF_PROF
    unsigned long long stamps_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    DPF_STAMP(0)
    const KArgs& a = d.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = lane & 15, g = lane >> 4;
    const int b0 = blockIdx.x * NCDE_TILE;
    const int bs = b0 + s;
    const bool valid = bs < a.B;
    const int Hr = a.H, Cr = a.C;
    const long long BH = (long long)a.B * Hr;
    const int cur = c->cur;
    const float* Ycur = cur ? d.YC : d.Y0;
    float* Ynxt = cur ? d.Y0 : d.YC;
    float* K1cur = d.KY + (long long)cur * BH;
    float* K1nxt = d.KY + (long long)(cur ^ 1) * BH;
    const float dtf = c->dtf, h0 = c->h0;
    const float rtolf = (float)d.rtol, atolf = (float)d.atol;
    float mx = 0.0f;

    // ---- dX/dt of every stage of the attempt: requested here, BEFORE the weight image (independent loads travel together), stored to
    //      LDS [stage][sample][channel] after it
    const int nst = phase == DP_STEP ? 6 : 1;
    constexpr int EPQ = (6 * 16 * CP + NT - 1) / NT;
    float qn[EPQ];
#pragma unroll
    for (int q = 0; q < EPQ; ++q) {
        const int e = tid + q * NT;
        const int j = e / (16 * CP), rem = e - j * (16 * CP), es = rem / CP, cc = rem - es * CP;
        float v = 0.0f;
        if (e < nst * 16 * CP && cc < Cr && b0 + es < a.B) {
            const StageDesc sd = c->st[phase == DP_STEP ? j + 1 : 0];
            const float* p = a.coeffs + (long long)(b0 + es) * a.cs_b + (long long)sd.idx * a.cs_t;
            if (a.interp == NCDE_INTERP_LINEAR) {
                v = p[a.cs_t + cc] - p[cc];
                if (sd.kdt != 1.0f) v = v / sd.kdt;
            } else {
                const float bb = p[Cr + cc], c2 = p[2 * Cr + cc], dd = p[3 * Cr + cc];
                const float inner = c2 + dd * sd.frac;
                v = bb + inner * sd.frac;
            }
        }
        qn[q] = v;
    }
    // ---- weights -> split-bf16 A operands in registers, from the per-lane image ncde_dpf_pack wrote once per solve ----------------
    typedef SplitOps<0> SO;
    typedef typename SO::T SpT;
    SpT w0[HT][KC0], w1[HT][KC], wo[NB][CQ][KC];
    f32x4 bias0[HT], bias1[HT], biaso[NB][CQ];
    {
        typedef DpfPack<H, HH, C> PK;
        const float* wp8 = d.WP + (long long)wave * PK::WAVE + lane * 8;
        const float* wp4 = d.WP + (long long)wave * PK::WAVE + PK::F8 * 512 + lane * 4;
        auto frag = [&](int f) {
            float tmp[8];
            const f32x4 lo = *reinterpret_cast<const f32x4*>(wp8 + f * 512), hi = *reinterpret_cast<const f32x4*>(wp8 + f * 512 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { tmp[j] = lo[j]; tmp[4 + j] = hi[j]; }
            return SO::split(tmp, mx);
        };
#pragma unroll
        for (int t = 0; t < HT; ++t) {
#pragma unroll
            for (int cc = 0; cc < KC0; ++cc) w0[t][cc] = frag(t * KC0 + cc);
#pragma unroll
            for (int cc = 0; cc < KC; ++cc) w1[t][cc] = frag(HT * KC0 + t * KC + cc);
            bias0[t] = *reinterpret_cast<const f32x4*>(wp4 + t * 256);
            bias1[t] = *reinterpret_cast<const f32x4*>(wp4 + (HT + t) * 256);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int cq = 0; cq < CQ; ++cq) {
#pragma unroll
                for (int cc = 0; cc < KC; ++cc) wo[nb][cq][cc] = frag(HT * KC0 + HT * KC + (nb * CQ + cq) * KC + cc);
                biaso[nb][cq] = *reinterpret_cast<const f32x4*>(wp4 + (2 * HT + nb * CQ + cq) * 256);
            }
    }

    DPF_STAMP(1)
    // ---- dX/dt -> LDS ----------------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int q = 0; q < EPQ; ++q) {
        const int e = tid + q * NT;
        if (e < nst * 16 * CP) dxq[e / (16 * CP)][e % (16 * CP)] = qn[q];
    }

    // ---- state entries this lane owns: u = 4 (wave NB + nb) + g of sample s --------------------------------------------------
    float y0[NB], kk[7][NB];
    long long gi[NB];
    bool own[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int u = 4 * (wave * NB + nb) + g;
        own[nb] = valid && u < Hr;
        gi[nb] = own[nb] ? (long long)bs * Hr + u : 0;
        y0[nb] = own[nb] ? Ycur[gi[nb]] : 0.0f;
        kk[0][nb] = (own[nb] && phase != DP_INIT0) ? K1cur[gi[nb]] : 0.0f;
#pragma unroll
        for (int j = 1; j < 7; ++j) kk[j][nb] = 0.0f;
    }
    const int n_inner = a.n_layers - 1;
    int zpar = 0;
    float zreg[KC0][8];
    auto exchange = [&](const float* ys) {      // stage input: owned entries -> the layer-0 B operand of every wave
        float* zw = zx[zpar];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) zw[(4 * (wave * NB + nb) + g) * 16 + s] = ys[nb];
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < KC0; ++cc)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) zreg[cc][jj] = zw[(32 * cc + 8 * g + jj) * 16 + s];
        zpar ^= 1;
    };
    auto evaluate = [&](int slot, float* kout) {      // kout[nb] = (f_theta(z) dX/dt)[u] with dX/dt of stage `slot`
        const float* dxp = dxq[slot] + s * CP;
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
            for (int cc = 0; cc < KC; ++cc) xb[cc] = SO::split(hv[cc], mx);
        };
        {
            SpT zb[KC0];
#pragma unroll
            for (int cc = 0; cc < KC0; ++cc) zb[cc] = SO::split(zreg[cc], mx);
#pragma unroll
            for (int tt = 0; tt < HT; ++tt) acc[tt] = SO::init(bias0[tt]);
#pragma unroll
            for (int cc = 0; cc < KC0; ++cc)
#pragma unroll
                for (int tt = 0; tt < HT; ++tt) SO::mac(w0[tt][cc], zb[cc], acc[tt]);
        }
        activate();
        for (int rep = 0; rep < n_inner; ++rep) {
#pragma unroll
            for (int tt = 0; tt < HT; ++tt) acc[tt] = SO::init(bias1[tt]);
#pragma unroll
            for (int cc = 0; cc < KC; ++cc)
#pragma unroll
                for (int tt = 0; tt < HT; ++tt) SO::mac(w1[tt][cc], xb[cc], acc[tt]);
            activate();
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) kout[nb] = 0.0f;
#pragma unroll
        for (int cq = 0; cq < CQ; ++cq) {
            typename SO::Acc oa[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) oa[nb] = SO::init(biaso[nb][cq]);
#pragma unroll
            for (int cc = 0; cc < KC; ++cc)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) SO::mac(wo[nb][cq][cc], xb[cc], oa[nb]);
            const f32x4 dx = *reinterpret_cast<const f32x4*>(dxp + 4 * cq);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const f32x4 o = SO::finish(oa[nb]);
#pragma unroll
                for (int r = 0; r < 4; ++r) kout[nb] = fmaf(tanh_prescaled(o[r]), dx[r], kout[nb]);
            }
        }
    };

    double sum0 = 0.0, sum1 = 0.0;
    DPF_STAMP(2)
    if (phase != DP_STEP) {
        float ys[NB], kout[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) ys[nb] = phase == DP_INIT0 ? y0[nb] : y0[nb] + h0 * kk[0][nb];
        exchange(ys);
        evaluate(0, kout);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (!own[nb]) continue;
            const float scale = atolf + fabsf(y0[nb]) * rtolf;
            if (phase == DP_INIT0) {
                K1cur[gi[nb]] = kout[nb];
                const float q0 = y0[nb] / scale, q1 = kout[nb] / scale;
                sum0 += (double)q0 * q0;
                sum1 += (double)q1 * q1;
            } else {
                const float q = (kout[nb] - kk[0][nb]) / scale;
                sum0 += (double)q * q;
            }
        }
    } else {
        float y1[NB];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float ys[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                float acc = 0.0f;
#pragma unroll
                for (int i = 0; i <= j; ++i) acc += kk[i][nb] * (kBeta[j][i] * dtf);
                ys[nb] = y0[nb] + acc;
                if (j == 5) y1[nb] = ys[nb];      // c_sol = (beta[-1], 0): the input of the last stage IS the solution (rk_common.py:76-80)
            }
            exchange(ys);
            evaluate(j, kk[j + 1]);
        }
        DPF_STAMP(3)
        // rows of the output inside (t0, t1]: 4th-order dense output (interp.py:4-61), written tentatively
        const double t0 = c->t0, t1 = c->t1;
        int jrow = c->j_out;
        const int m_tape = c->n_accept;
        const bool any_row = jrow < d.n_t && !(d.t_out[jrow] > t1);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (!own[nb]) continue;
            float err = 0.0f;
#pragma unroll
            for (int i = 0; i < 6; ++i) err += kk[i][nb] * (dtf * kCErr[i]);
            err += kk[6][nb] * (dtf * kCErr[6]);
            const float tol = atolf + rtolf * fmaxf(fabsf(y0[nb]), fabsf(y1[nb]));
            const float q = err / tol;
            sum0 += (double)q * q;
            Ynxt[gi[nb]] = y1[nb];
            K1nxt[gi[nb]] = kk[6][nb];
            if (d.tape_y && m_tape < d.tape_cap) d.tape_y[(long long)m_tape * BH + gi[nb]] = y0[nb];
            if (any_row) {
                float ym = 0.0f;
#pragma unroll
                for (int i = 0; i < 7; ++i) ym += kk[i][nb] * (dtf * kMid[i]);
                ym = y0[nb] + ym;
                const int u = 4 * (wave * NB + nb) + g;
                for (int jr = jrow; jr < d.n_t && !(d.t_out[jr] > t1); ++jr) {
                    const float x = (float)((d.t_out[jr] - t0) / (t1 - t0));
                    d.out[((long long)bs * d.n_t + jr) * Hr + u] = dp_poly(y0[nb], y1[nb], ym, kk[0][nb], kk[6][nb], dtf, x);
                }
            }
        }
    }
    DPF_STAMP(4)
    // ---- this workgroup's partial sums of squares, then the controller in the last workgroup -----------------------------------
    sum0 = wave_sum_d(sum0);
    sum1 = wave_sum_d(sum1);
    if (lane == 0) { red[0][wave] = sum0; red[1][wave] = sum1; }
    __syncthreads();
    if (tid < 4)
        __hip_atomic_store(&d.PN[(long long)blockIdx.x * 4 + tid], tid < 2 ? (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]) : 0.0,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    DPF_STAMP(5)
#ifdef NCDE_DPF_PROF
    dpf_finish_launch(d, sh, sh_flags, &sh_x, &is_last, stamps_);
    DPF_STAMP(7)
    if (blockIdx.x == 0 && threadIdx.x == 0 && phase == DP_STEP && d.a.H >= 8)
        for (int k = 0; k < 8; ++k) d.out[k] = (float)(stamps_[k] - stamps_[0]);
#else
    dpf_finish_launch(d, sh, sh_flags, &sh_x, &is_last);
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// adjoint attempt: the augmented state (vjp_t, y, a, g_theta) of adjoint.py:37-145 in negated time
// ------------------------------------------------------------------------------------------------------------------
// One workgroup = one 16-sample tile, 4 waves (the structure of ncde_adj_fast, ncde_fast.hip: forward recompute register-to-register in
// fp32 MFMA, output tiles P -> tanh -> f, dP; Wo^T dP from an LDS image; hidden backward; weight gradients from wave-private
// [unit][sample] images with the samples as the K dimension).  What an ATTEMPT adds:
//   * seven stage evaluations -- stage 1 is re-evaluated from (y0, a0) at the descriptor the controller kept (ctrl->st_k1) instead of
//     carrying FSAL values: the parameter part of k1 would be a per-workgroup vector of |theta| floats to keep between launches;
//   * the parameter part enters only linearly, so the workgroup accumulates TWO weighted sums over the stages in registers,
//     INC = dt sum_j beta_6j k_j (5th-order increment) and ERR = dt sum_j e_j k_j (embedded error), and writes both once per attempt;
//     ncde_dpf_reduce sums them over the workgroups, forms the mixed error norm (adjoint.py:239-242) and runs the controller;
//   * phases: DP_INIT0 / DP_INIT1 (one evaluation each: f0 and the probe of the initial-step rule, parameter part with weight 1),
//     DP_STEP, and DP_FIN after an accepted attempt that reached the end of the output interval: the same seven evaluations with the
//     weights of the 4th-order dense output at the interval end (interp.py:4-61, linear in the k_j), which also resets (y, a) to the
//     stored solution / adds the next output's cotangent (adjoint.py:116-133).
template <int H, int HH, int C, int NL>
struct DpaPack {
    static constexpr int NW = 4, CP = (C + 3) & ~3, CQ = CP / 4, HB = H / 4, HT = HH / 16, KH = HH / 4, NB = HB / NW, NTILE = NB * CQ;
    // per wave and lane (registers): the A operands of the wave's own output tiles, the hidden-layer biases
    static constexpr int O_WO = 0, LANE = NB * CQ * KH, LANE4 = (LANE + 3) / 4;
    static constexpr int WAVE = LANE4 * 256;                    // floats per wave: [chunk of 4][lane][4]
    // shared operand images (LDS): [tile][chunk of 4 k-steps][lane][4] -- one ds_read_b128 feeds four MFMAs
    static constexpr int WOT = NW * NTILE * HT * 256, BOL = NW * NTILE * 16;
    static constexpr int I_W0 = 0, I_W1 = I_W0 + HT * (HB / 4) * 256, I_W1T = I_W1 + HT * (KH / 4) * 256, I_W0T = I_W1T + HT * (KH / 4) * 256,
                         I_WOT = I_W0T + NW * (KH / 4) * 256, I_BOL = I_WOT + WOT, I_B0 = I_BOL + BOL, I_B1 = I_B0 + HT * 16,
                         IMGS = I_B1 + HT * 16;      // I_B0 / I_B1: hidden-layer biases [t][g][r] (the accumulator a lane starts a tile with)
    static constexpr int TOTAL = NW * WAVE + IMGS;
};

template <int H, int HH, int C, int NL>
__global__ __launch_bounds__(256) void ncde_dpa_pack(DpArgs d) {
    typedef DpaPack<H, HH, C, NL> PK;
    constexpr int CQ = PK::CQ, HB = PK::HB, HT = PK::HT, KH = PK::KH, NB = PK::NB, NTILE = PK::NTILE, NW = 4;
    const KArgs& a = d.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = lane & 15, g = lane >> 4;
    const int Hr = a.H, HHr = a.dout[0], Cr = a.C;
    const bool has_inner = a.n_layers > 1;
    float* wp = d.WP + (long long)wave * PK::WAVE;
    auto put = [&](int i, float v) { wp[((i >> 2) * 64 + lane) * 4 + (i & 3)] = v; };
    for (int nb = 0; nb < NB; ++nb)
        for (int cq = 0; cq < CQ; ++cq) {
            const int hA = 4 * (wave * NB + nb) + (s >> 2), cA = 4 * cq + (s & 3);
            for (int ks = 0; ks < KH; ++ks)
                put(PK::O_WO + (nb * CQ + cq) * KH + ks,
                    (hA < Hr && cA < Cr && 4 * ks + g < HHr) ? NCDE_TANH_PRESCALE * a.Wo[(long long)(hA * Cr + cA) * HHr + 4 * ks + g] : 0.0f);
        }
    float* im = d.WP + (long long)NW * PK::WAVE;
    if (tid < HT * 16) {
        const int r = tid & 3, gg = (tid >> 2) & 3, t = tid >> 4;
        const int unitD = 4 * (4 * t + r) + gg;
        im[PK::I_B0 + tid] = unitD < HHr ? a.b[0][unitD] : 0.0f;
        im[PK::I_B1 + tid] = (has_inner && unitD < HHr) ? a.b[1][unitD] : 0.0f;
    }
    if (wave == 0) {      // hidden-layer A operands: row i = s <-> unit 4 (4t + (s & 3)) + (s >> 2), k-step ks, k = g
        for (int t = 0; t < HT; ++t) {
            const int unitA = 4 * (4 * t + (s & 3)) + (s >> 2);
            for (int ks = 0; ks < HB; ++ks)
                im[PK::I_W0 + ((t * (HB / 4) + (ks >> 2)) * 64 + lane) * 4 + (ks & 3)] = (unitA < HHr && 4 * ks + g < Hr) ? a.W[0][unitA * Hr + 4 * ks + g] : 0.0f;
            for (int ks = 0; ks < KH; ++ks) {
                const bool ok = has_inner && unitA < HHr && 4 * ks + g < HHr;
                im[PK::I_W1 + ((t * (KH / 4) + (ks >> 2)) * 64 + lane) * 4 + (ks & 3)] = ok ? a.W[1][unitA * HHr + 4 * ks + g] : 0.0f;
                im[PK::I_W1T + ((t * (KH / 4) + (ks >> 2)) * 64 + lane) * 4 + (ks & 3)] = ok ? a.W[1][(4 * ks + g) * HHr + unitA] : 0.0f;
            }
        }
    }
    {   // W0^T rows for the state entries a wave owns: tile row i <-> h = 4 (wave NB + (i & 3)) + (i >> 2)
        const int r_own = s & 3;
        const int hrow = 4 * (wave * NB + r_own) + (s >> 2);
        for (int ks = 0; ks < KH; ++ks)
            im[PK::I_W0T + ((wave * (KH / 4) + (ks >> 2)) * 64 + lane) * 4 + (ks & 3)] =
                (r_own < NB && hrow < Hr && 4 * ks + g < HHr) ? a.W[0][(4 * ks + g) * Hr + hrow] : 0.0f;
    }
    for (int e = tid; e < PK::WOT; e += 256) {      // Wo^T image [wave][tile][tp][lane][r]
        const int r = e & 3, l = (e >> 2) & 63, rest = e >> 8;
        const int tp = rest % HT, tau = (rest / HT) % NTILE, wv = rest / (HT * NTILE);
        const int nb = tau / CQ, cq = tau - nb * CQ;
        const int h = 4 * (wv * NB + nb) + (l >> 4), c = 4 * cq + r;
        const int jrow = 4 * (4 * tp + (l & 3)) + ((l & 15) >> 2);
        im[PK::I_WOT + e] = (h < Hr && c < Cr && jrow < HHr) ? a.Wo[(long long)(h * Cr + c) * HHr + jrow] : 0.0f;
    }
    for (int e = tid; e < PK::BOL; e += 256) {      // bo image [wave][tile][g][r]
        const int r = e & 3, gg = (e >> 2) & 3, rest = e >> 4;
        const int tau = rest % NTILE, wv = rest / NTILE;
        const int nb = tau / CQ, cq = tau - nb * CQ;
        const int h = 4 * (wv * NB + nb) + gg, c = 4 * cq + r;
        im[PK::I_BOL + e] = (h < Hr && c < Cr) ? NCDE_TANH_PRESCALE * a.bo[h * Cr + c] : 0.0f;
    }
}

// Where everything lives (the register file is the constraint: 512 per lane at one wave per SIMD, and ncde_adj_fast already fills it
// with ONE gradient sum): registers hold the wave's own Wo rows (A operand of P = Wo x_L, 80), BOTH sums of dWo as plain MFMA
// accumulators (2 x 80; the stage weight rides on the A operand), the small sums and the stage values; every other operand is read
// from LDS images with one ds_read_b128 per four MFMAs -- Wo^T as in ncde_adj_fast, and here also W0, W1, W1^T, W0^T (72 registers
// there).  The wave-private [unit][sample] images keep only the rows the wave's own weight-gradient tiles read.  (Two earlier
// versions: both sums updated by VALU fma spilled 258 dwords per lane, 30 us per stage; the second sum read-modify-written in LDS
// with Wo^T streamed from L2 one tile ahead spilled 152 and waited on L2, 25 us per stage.)
template <int H, int HH, int C, int NL>
__global__ __launch_bounds__(256, 1) void ncde_dpf_adj(DpArgs d) {
    typedef DpaPack<H, HH, C, NL> PK;
    constexpr int NW = 4, NT = 256;
    constexpr int CP = PK::CP, CQ = PK::CQ, HB = PK::HB, HT = PK::HT, KH = PK::KH, NB = PK::NB, NTILE = PK::NTILE;
    constexpr int HT0 = H / 16;
    constexpr int XS = 20;
    constexpr int R_Z = 0, R_X = 16, R_XL = R_X + 16 * (NL - 1), R_DP = R_XL + HH, PRIV = (R_DP + HH) * XS;      // image rows of a wave
    constexpr int EPT = (16 * CP + NT - 1) / NT;
    static_assert(H % (4 * NW) == 0 && HH % 16 == 0 && H % 16 == 0 && NB <= 4 && NTILE <= 16 && KH <= 16, "shape not tileable");
    static_assert(HT * HT == NW && HT * HT0 == NW, "one dW1 tile and one dW0 tile per wave");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* zx = lds;                              // [2][H*16]   stage-state exchange
    float* dxs = zx + 2 * H * 16;                 // [2][16*CP]  dX/dt of the stage, by stage parity
    float* d2s = dxs + 2 * 16 * CP;               // [2][16*CP]  d2X/dt2 (cubic control: the vjp_t component)
    float* red = d2s + 2 * 16 * CP;               // [NW][HH*16] dL/dx_L partials
    float* simg = red + NW * HH * 16;             // operand images (DpaPack::I_*)
    float* privbase = simg + PK::IMGS;            // [NW][PRIV]
    double* sh = reinterpret_cast<double*>(privbase + NW * PRIV);      // [4][NW] partial sums
    __shared__ StageDesc sds[8];

    DpCtrl* c = d.ctrl;
    const int phase = c->phase;
    if (phase == DP_DONE || c->error != 0) return;      // (uniform)
#ifdef NCDE_DPF_PROF
    unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = wall_clock64();
#endif
    const KArgs& a = d.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = lane & 15, g = lane >> 4;
    const int b0 = blockIdx.x * NCDE_TILE;
    const int bs = b0 + s;
    const bool valid = bs < a.B;
    const int Hr = a.H, HHr = a.dout[0], Cr = a.C;
    const int cur = c->cur;
    const float* Ycur = cur ? d.YC : d.Y0;
    float* Ynxt = cur ? d.Y0 : d.YC;
    const float* Acur = cur ? d.AC : d.A0;
    float* Anxt = cur ? d.A0 : d.AC;
    const bool step = phase == DP_STEP || phase == DP_FIN;
    const bool two = phase == DP_STEP;
    const float dtf = phase == DP_FIN ? c->dtf_commit : c->dtf, h0 = c->h0;
    const float rtolf = (float)d.rtol, atolf = (float)d.atol;
    const bool cubic = a.interp == NCDE_INTERP_CUBIC;
    const int nst = step ? 7 : 1;
    float* priv = privbase + wave * PRIV;
    float* zimg = priv + R_Z * XS;                // z, the 16 rows the wave's dW0 tile reads (units 16 tc0 ..)
    float* xLimg = priv + R_XL * XS;              // x_NL, all rows (B operand of dWo)
    float* dpimg = priv + R_DP * XS;              // [HH][XS]   dL/dpre of the current layer
    float* dptile = dpimg;                        // [16][XS]   dP of the current output tile (aliases dpimg: disjoint phases)
    const int tr1 = wave / HT, tc1 = wave - tr1 * HT;        // the wave's dW1 tile
    const int tr0 = wave / HT0, tc0 = wave - tr0 * HT0;      // ... and dW0 tile
    const float* w0i = simg + PK::I_W0 + lane * 4;
    const float* w1i = simg + PK::I_W1 + lane * 4;
    const float* w1ti = simg + PK::I_W1T + lane * 4;
    const float* w0ti = simg + PK::I_W0T + (wave * (KH / 4) * 64 + lane) * 4;
    const float* woTw = simg + PK::I_WOT + wave * NTILE * HT * 256;
    const float* boLw = simg + PK::I_BOL + wave * NTILE * 16;
    if (tid < 7) sds[tid] = step ? (tid == 0 ? c->st_k1 : c->st[tid]) : c->st[0];

    // ---- weights: the wave's Wo rows and the biases -> registers, operand images -> LDS ----------------------------------------------
    float wo[NB][CQ][KH];
    const float* b0i = simg + PK::I_B0 + g * 4;
    const float* b1i = simg + PK::I_B1 + g * 4;
    {
        const float* wp = d.WP + (long long)wave * PK::WAVE + lane * 4;
        float wreg[PK::LANE4 * 4];
#pragma unroll
        for (int i4 = 0; i4 < PK::LANE4; ++i4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(wp + i4 * 256);
#pragma unroll
            for (int q = 0; q < 4; ++q) wreg[4 * i4 + q] = v[q];
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int cq = 0; cq < CQ; ++cq)
#pragma unroll
                for (int ks = 0; ks < KH; ++ks) wo[nb][cq][ks] = wreg[PK::O_WO + (nb * CQ + cq) * KH + ks];
        const f32x4* src = reinterpret_cast<const f32x4*>(d.WP + (long long)NW * PK::WAVE);
        f32x4* dst = reinterpret_cast<f32x4*>(simg);
        constexpr int NV = PK::IMGS / 4;
        int e = tid;
        for (; e + 7 * NT < NV; e += 8 * NT) {      // eight 16-byte loads in flight per thread
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[e + u * NT];
#pragma unroll
            for (int u = 0; u < 8; ++u) dst[e + u * NT] = v[u];
        }
        for (; e < NV; e += NT) dst[e] = src[e];
    }

    // ---- dX/dt (and d2X/dt2) of a stage: loaded into registers early, stored to the parity slot before the stage's last barrier -------
    float dxv[EPT], d2v[EPT];
    auto stage_load = [&](int j) {
        const StageDesc sd = sds[j];
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = tid + q * NT;
            const int es = e / CP, cc = e - es * CP;
            float v = 0.0f, v2 = 0.0f;
            if (e < 16 * CP && cc < Cr && b0 + es < a.B) {
                const float* p = a.coeffs + (long long)(b0 + es) * a.cs_b + (long long)sd.idx * a.cs_t;
                if (!cubic) {
                    v = p[a.cs_t + cc] - p[cc];
                    if (sd.kdt != 1.0f) v = v / sd.kdt;
                } else {
                    const float bb = p[Cr + cc], c2 = p[2 * Cr + cc], dd = p[3 * Cr + cc];
                    const float inner = c2 + dd * sd.frac;
                    v = bb + inner * sd.frac;
                    v2 = inner + dd * sd.frac;
                }
            }
            dxv[q] = v;
            d2v[q] = v2;
        }
    };
    auto stage_store = [&](int j) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = tid + q * NT;
            if (e < 16 * CP) {
                dxs[(j & 1) * 16 * CP + e] = dxv[q];
                d2s[(j & 1) * 16 * CP + e] = d2v[q];
            }
        }
    };

    // ---- owned state entries -----------------------------------------------------------------------------------------------------
    float y0[NB], a0[NB], ky[7][NB], ka[7][NB], ys[NB], as_[NB], y1[NB], a1[NB];
    long long gi[NB];
    bool own[NB];
    const int row_fin = c->row_now;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int u = 4 * (wave * NB + nb) + g;
        own[nb] = valid && u < Hr;
        gi[nb] = own[nb] ? (long long)bs * Hr + u : 0;
        y0[nb] = own[nb] ? Ycur[gi[nb]] : 0.0f;
        a0[nb] = own[nb] ? Acur[gi[nb]] : 0.0f;
#pragma unroll
        for (int j = 0; j < 7; ++j) ky[j][nb] = ka[j][nb] = 0.0f;
        ys[nb] = y0[nb];
        as_[nb] = a0[nb];
        y1[nb] = a1[nb] = 0.0f;
        if (phase == DP_INIT1) {
            const float f0y = own[nb] ? d.KY[gi[nb]] : 0.0f, f0a = own[nb] ? d.KA[gi[nb]] : 0.0f;
            ky[0][nb] = f0y;      // (kept for the norm of the difference below; the single evaluation of this phase lands in slot 1)
            ka[0][nb] = f0a;
            ys[nb] = y0[nb] + h0 * f0y;
            as_[nb] = a0[nb] + h0 * f0a;
        }
    }
    __syncthreads();      // sds, operand images
    stage_load(0);
    stage_store(0);

    // weights of the dense output at the interval end (DP_FIN): g_end - g0 = sum_j wfin_j k_j (see the header)
    const float xe = c->x_end;
    auto fin_weight = [&](int j) {
        const float bj = j < 6 ? kBeta[5][j] : 0.0f, mj = kMid[j];
        const float d1 = j == 0 ? 1.0f : 0.0f, d7 = j == 6 ? 1.0f : 0.0f;
        const float c2 = d7 - 4.0f * d1 - 5.0f * bj + 16.0f * mj;
        const float c3 = 5.0f * d1 - 3.0f * d7 + 14.0f * bj - 32.0f * mj;
        const float c4 = 2.0f * d7 - 2.0f * d1 - 8.0f * bj + 16.0f * mj;
        return dtf * (xe * (d1 + xe * (c2 + xe * (c3 + xe * c4))));
    };

    // ---- gradient accumulators: set 0 = INC (or the single evaluation / the dense-output sum), set 1 = ERR -----------------------------
    // bias sums are reduced over the 16 samples per stage and kept by ONE lane per unit (s == tile for dbo, s == k-step for db1 / db0)
    f32x4 gWo[2][NTILE][HT], gW1[2], gW0[2], gbo[2];
    float gb1[2] = {0.0f, 0.0f}, gb0[2] = {0.0f, 0.0f}, gvt[2] = {0.0f, 0.0f};
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        gbo[q] = gW1[q] = gW0[q] = zero4;
#pragma unroll
        for (int i = 0; i < NTILE; ++i)
#pragma unroll
            for (int t = 0; t < HT; ++t) gWo[q][i][t] = zero4;
    }
    float zreg[HB];
    int zpar = 0;
    auto exchange = [&]() {      // owned stage inputs -> the layer-0 B operand of every wave (publishes the dX slot stored before it, too)
        float* zw = zx + zpar * H * 16;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) zw[(4 * (wave * NB + nb) + g) * 16 + s] = ys[nb];
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < HB; ++ks) zreg[ks] = zw[(4 * ks + g) * 16 + s];
        zpar ^= 1;
    };
    exchange();
    DPA_TICK(0)

#pragma unroll 1
    for (int j = 0; j < nst; ++j) {
        if (j + 1 < nst) stage_load(j + 1);
        float w_inc, w_err;
        if (phase == DP_STEP) {
            w_inc = j < 6 ? kBeta[5][j] * dtf : 0.0f;
            w_err = kCErr[j] * dtf;
        } else if (phase == DP_FIN) {
            w_inc = fin_weight(j);
            w_err = 0.0f;
        } else {
            w_inc = 1.0f;
            w_err = 0.0f;
        }
        const float* dxp = dxs + (j & 1) * 16 * CP + s * CP;
        const float* d2p = d2s + (j & 1) * 16 * CP + s * CP;
        // ---- forward recompute; keep x_1..x_NL (registers) and the image rows the weight-gradient tiles read (LDS) -----------------------
        float xc[KH];               // the layer just computed (x_NL after the loop)
        unsigned relu_mask = 0;     // bit (8 l + ks): x_{l+1}[4 ks + g] > 0, for the layers below the last
        {
            f32x4 acc[HT];
#pragma unroll
            for (int tt = 0; tt < HT; ++tt) acc[tt] = *reinterpret_cast<const f32x4*>(b0i + tt * 16);
#pragma unroll
            for (int q4 = 0; q4 < HB / 4; ++q4) {
                f32x4 a4[HT];
#pragma unroll
                for (int tt = 0; tt < HT; ++tt) a4[tt] = *reinterpret_cast<const f32x4*>(w0i + (tt * (HB / 4) + q4) * 256);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) acc[tt] = mfma16(a4[tt][i], zreg[4 * q4 + i], acc[tt]);
            }
#pragma unroll
            for (int tt = 0; tt < HT; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) xc[4 * tt + r] = relu_dev(acc[tt][r]);
#pragma unroll
            for (int l = 1; l < NL; ++l) {
                // x_l is complete: its image rows for the wave's dW1 tile, its mask bits
#pragma unroll
                for (int q4 = 0; q4 < KH / 4; ++q4)
                    if (q4 == tc1) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) priv[((R_X + 16 * (l - 1)) + 4 * i + g) * XS + s] = xc[4 * q4 + i];
                    }
#pragma unroll
                for (int ks = 0; ks < KH; ++ks) relu_mask |= (xc[ks] > 0.0f ? 1u : 0u) << (8 * (l - 1) + ks);
#pragma unroll
                for (int tt = 0; tt < HT; ++tt) acc[tt] = *reinterpret_cast<const f32x4*>(b1i + tt * 16);
#pragma unroll
                for (int q4 = 0; q4 < KH / 4; ++q4) {
                    f32x4 a4[HT];
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) a4[tt] = *reinterpret_cast<const f32x4*>(w1i + (tt * (KH / 4) + q4) * 256);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int tt = 0; tt < HT; ++tt) acc[tt] = mfma16(a4[tt][i], xc[4 * q4 + i], acc[tt]);
                }
#pragma unroll
                for (int tt = 0; tt < HT; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xc[4 * tt + r] = relu_dev(acc[tt][r]);
            }
        }
#pragma unroll
        for (int q4 = 0; q4 < HB / 4; ++q4)
            if (q4 == tc0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) zimg[(4 * i + g) * XS + s] = zreg[4 * q4 + i];
            }
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) xLimg[(4 * ks + g) * XS + s] = xc[ks];
        wave_lds_order();
        DPA_TICK(1)
        f32x4 xB[HT];      // B operands of the dWo GEMM: x_NL[j = 16t + n][samples 4g..4g+3]
#pragma unroll
        for (int tt = 0; tt < HT; ++tt) xB[tt] = *reinterpret_cast<const f32x4*>(xLimg + (16 * tt + s) * XS + 4 * g);
        // ---- output tiles owned by this wave --------------------------------------------------------------------------------------------
        float kout[NB], vt = 0.0f;
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
                for (int nb = 0; nb < NB; ++nb) o[nb] = mfma16(wo[nb][cq][ks], xc[ks], o[nb]);
            const f32x4 dx = *reinterpret_cast<const f32x4*>(dxp + 4 * cq);
            const f32x4 d2 = *reinterpret_cast<const f32x4*>(d2p + 4 * cq);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int tau = nb * CQ + cq;
                f32x4 dP;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float m = tanh_prescaled(o[nb][r]);
                    kout[nb] = fmaf(m, dx[r], kout[nb]);
                    dP[r] = (as_[nb] * dx[r]) * (1.0f - m * m);
                    vt = fmaf(as_[nb] * m, d2[r], vt);
                }
#pragma unroll
                for (int tt = 0; tt < HT; ++tt) {      // dL/dx_L partial: k-step <-> r
                    const f32x4 av = *reinterpret_cast<const f32x4*>(woTw + ((tau * HT + tt) * 64 + lane) * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) accJ[tt] = mfma16(av[r], dP[r], accJ[tt]);
                }
                {      // bias gradient: sum over the 16 samples now, kept by the lanes with s == tau
                    f32x4 sm;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sm[r] = row16_sum(dP[r]);
                        dptile[(4 * g + r) * XS + s] = dP[r];
                    }
                    if (s == tau) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            gbo[0][r] = fmaf(w_inc, sm[r], gbo[0][r]);
                            gbo[1][r] = fmaf(w_err, sm[r], gbo[1][r]);
                        }
                    }
                }
                wave_lds_order();
                const f32x4 av = *reinterpret_cast<const f32x4*>(dptile + s * XS + 4 * g);
                wave_lds_order();
                // dWo += w dP x_L^T (samples are the K dimension): the stage weight rides on the A operand, so both sums are plain MFMA
                // accumulations
                const f32x4 avi = {w_inc * av[0], w_inc * av[1], w_inc * av[2], w_inc * av[3]};
#pragma unroll
                for (int tt = 0; tt < HT; ++tt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) gWo[0][tau][tt] = mfma16(avi[q], xB[tt][q], gWo[0][tau][tt]);
                if (two) {
                    const f32x4 ave = {w_err * av[0], w_err * av[1], w_err * av[2], w_err * av[3]};
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt)
#pragma unroll
                        for (int q = 0; q < 4; ++q) gWo[1][tau][tt] = mfma16(ave[q], xB[tt][q], gWo[1][tau][tt]);
                }
            }
        }
        DPA_TICK(2)
        gvt[0] = fmaf(w_inc, vt, gvt[0]);
        gvt[1] = fmaf(w_err, vt, gvt[1]);
        // ---- sum the dL/dx_L partials over the waves ------------------------------------------------------------------------------------
        float gpre[KH];
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
            gpre[ks] = xc[ks] > 0.0f ? v : 0.0f;
        }
        DPA_TICK(3)
        // ---- hidden layers backward (shared W1), then W0 ----------------------------------------------------------------------------------
        auto bias_sum = [&](float* gb) {      // unit 4 ks + g: sum over the samples, kept by the lane with s == ks
#pragma unroll
            for (int ks = 0; ks < KH; ++ks) {
                const float sm = row16_sum(gpre[ks]);
                if (s == ks) {
                    gb[0] = fmaf(w_inc, sm, gb[0]);
                    gb[1] = fmaf(w_err, sm, gb[1]);
                }
                dpimg[(4 * ks + g) * XS + s] = gpre[ks];
            }
            wave_lds_order();
        };
        auto grad_tile = [&](f32x4* gw, int tr, const float* bimg) {      // the wave's 16 x 16 tile of dW: rows 16 tr.. of dL/dpre, 16 image rows
            const f32x4 av = *reinterpret_cast<const f32x4*>(dpimg + (16 * tr + s) * XS + 4 * g);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bimg + s * XS + 4 * g);
            const f32x4 avi = {w_inc * av[0], w_inc * av[1], w_inc * av[2], w_inc * av[3]};
#pragma unroll
            for (int q = 0; q < 4; ++q) gw[0] = mfma16(avi[q], bv[q], gw[0]);
            if (two) {
                const f32x4 ave = {w_err * av[0], w_err * av[1], w_err * av[2], w_err * av[3]};
#pragma unroll
                for (int q = 0; q < 4; ++q) gw[1] = mfma16(ave[q], bv[q], gw[1]);
            }
            wave_lds_order();
        };
#pragma unroll
        for (int l = NL - 1; l >= 1; --l) {
            bias_sum(gb1);
            grad_tile(gW1, tr1, priv + (R_X + 16 * (l - 1)) * XS);
            f32x4 acc[HT];
#pragma unroll
            for (int tt = 0; tt < HT; ++tt) acc[tt] = zero4;
#pragma unroll
            for (int q4 = 0; q4 < KH / 4; ++q4) {
                f32x4 a4[HT];
#pragma unroll
                for (int tt = 0; tt < HT; ++tt) a4[tt] = *reinterpret_cast<const f32x4*>(w1ti + (tt * (KH / 4) + q4) * 256);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) acc[tt] = mfma16(a4[tt][i], gpre[4 * q4 + i], acc[tt]);
            }
#pragma unroll
            for (int tt = 0; tt < HT; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) gpre[4 * tt + r] = ((relu_mask >> (8 * (l - 1) + 4 * tt + r)) & 1u) ? acc[tt][r] : 0.0f;
        }
        bias_sum(gb0);
        grad_tile(gW0, tr0, zimg);
        DPA_TICK(4)
        f32x4 vy = zero4;      // a^T df/dy for the state entries this wave owns
#pragma unroll
        for (int q4 = 0; q4 < KH / 4; ++q4) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(w0ti + q4 * 256);
#pragma unroll
            for (int i = 0; i < 4; ++i) vy = mfma16(a4[i], gpre[4 * q4 + i], vy);
        }
        // ---- the stage derivative in negated time (dy/ds = -f, da/ds = +a^T df/dy), next stage input -------------------------------------
        const int slot = step ? j : (phase == DP_INIT1 ? 1 : 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                ky[q][nb] = slot == q ? -kout[nb] : ky[q][nb];
                ka[q][nb] = slot == q ? vy[nb] : ka[q][nb];
            }
        }
        if (j + 1 < nst) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                float accy = 0.0f, acca = 0.0f;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const float bj = kBeta[j][i] * dtf;
                    accy += ky[i][nb] * bj;
                    acca += ka[i][nb] * bj;
                }
                ys[nb] = y0[nb] + accy;
                as_[nb] = a0[nb] + acca;
                if (j == 5) { y1[nb] = ys[nb]; a1[nb] = as_[nb]; }      // the input of the last stage IS the solution (rk_common.py:76-80)
            }
            stage_store(j + 1);
            exchange();
        }
        DPA_TICK(5)
    }

    // ---- epilogue by phase --------------------------------------------------------------------------------------------------------------
    double sum0 = 0.0, sum1 = 0.0, sum2 = 0.0, sum3 = 0.0;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        if (!own[nb]) continue;
        const long long gq = gi[nb];
        if (phase == DP_INIT0) {
            d.KY[gq] = ky[0][nb];
            d.KA[gq] = ka[0][nb];
            const float sy = atolf + fabsf(y0[nb]) * rtolf, sa = atolf + fabsf(a0[nb]) * rtolf;
            const float q0 = y0[nb] / sy, q1 = a0[nb] / sa, q2 = ky[0][nb] / sy, q3 = ka[0][nb] / sa;
            sum0 += (double)q0 * q0; sum1 += (double)q1 * q1; sum2 += (double)q2 * q2; sum3 += (double)q3 * q3;
        } else if (phase == DP_INIT1) {
            const float sy = atolf + fabsf(y0[nb]) * rtolf, sa = atolf + fabsf(a0[nb]) * rtolf;
            const float q0 = (ky[1][nb] - ky[0][nb]) / sy, q1 = (ka[1][nb] - ka[0][nb]) / sa;
            sum0 += (double)q0 * q0; sum1 += (double)q1 * q1;
        } else if (phase == DP_STEP) {
            float ey = 0.0f, ea = 0.0f;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const float cj = dtf * kCErr[i];
                ey += ky[i][nb] * cj;
                ea += ka[i][nb] * cj;
            }
            const float qy = ey / (atolf + rtolf * fmaxf(fabsf(y0[nb]), fabsf(y1[nb])));
            const float qa = ea / (atolf + rtolf * fmaxf(fabsf(a0[nb]), fabsf(a1[nb])));
            sum0 += (double)qy * qy; sum1 += (double)qa * qa;
            Ynxt[gq] = y1[nb];
            Anxt[gq] = a1[nb];
        } else {      // DP_FIN: a at the interval end (dense output) + the next output's cotangent; y reset to the stored solution
            float am = 0.0f;
#pragma unroll
            for (int i = 0; i < 7; ++i) am += ka[i][nb] * (dtf * kMid[i]);
            am = a0[nb] + am;
            const int u = 4 * (wave * NB + nb) + g;
            const long long o = ((long long)bs * d.n_t + row_fin) * Hr + u;
            Anxt[gq] = dp_poly(a0[nb], a1[nb], am, ka[0][nb], ka[6][nb], dtf, xe) + d.grad_out[o];
            Ynxt[gq] = d.z_out[o];
        }
    }
    // ---- this workgroup's parameter-part partial(s): [workgroup][set][theta1] -------------------------------------------------------------
    const int nset = two ? 2 : 1;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (q >= nset) break;
        // partial vector of a workgroup: [dWo in register order: wave, tile, tt, lane, r][theta1 entries in parameter order, the dWo
        // range unused] -- 16-byte stores here; ncde_dpf_reduce maps the first block to parameter order once, on the totals
        float* gn = d.GP + ((long long)blockIdx.x * 2 + q) * (PK::WOT + d.theta1);
        float* gp = gn + PK::WOT;
#pragma unroll
        for (int tau = 0; tau < NTILE; ++tau)
#pragma unroll
            for (int tt = 0; tt < HT; ++tt) *reinterpret_cast<f32x4*>(gn + (((wave * NTILE + tau) * HT + tt) * 64 + lane) * 4) = gWo[q][tau][tt];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int cq = 0; cq < CQ; ++cq) {
                const int tau = nb * CQ + cq;
                const int hh = 4 * (wave * NB + nb) + g;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cc = 4 * cq + r;
                    if (hh < Hr && cc < Cr && s == tau) gp[a.gbo_off + hh * Cr + cc] = gbo[q][r];
                }
            }
        if (NL > 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 * tr1 + 4 * g + r < HHr && 16 * tc1 + s < HHr) gp[a.gW_off[1] + (16 * tr1 + 4 * g + r) * HHr + 16 * tc1 + s] = gW1[q][r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (16 * tr0 + 4 * g + r < HHr && 16 * tc0 + s < Hr) gp[a.gW_off[0] + (16 * tr0 + 4 * g + r) * Hr + 16 * tc0 + s] = gW0[q][r];
        if (wave == 0 && s < KH && 4 * s + g < HHr) {      // the lane with s == ks holds the bias sums of unit 4 ks + g
            if (NL > 1) gp[a.gb_off[1] + 4 * s + g] = gb1[q];
            gp[a.gb_off[0] + 4 * s + g] = gb0[q];
        }
        // the time component vjp_t (cubic control; zero for a linear one): workgroup sum of the per-lane partials
        float v = row16_sum(gvt[q]);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        if (tid == 0) gp[a.theta_size] = (red[0] + red[1]) + (red[2] + red[3]);
    }
    // ---- partial sums of squares of the y / a parts; the controller runs in ncde_dpf_reduce ---------------------------------------------
    sum0 = wave_sum_d(sum0); sum1 = wave_sum_d(sum1); sum2 = wave_sum_d(sum2); sum3 = wave_sum_d(sum3);
    __syncthreads();
    if (lane == 0) { sh[0 * NW + wave] = sum0; sh[1 * NW + wave] = sum1; sh[2 * NW + wave] = sum2; sh[3 * NW + wave] = sum3; }
    __syncthreads();
    if (tid < 4) d.PN[(long long)blockIdx.x * 4 + tid] = (sh[tid * NW] + sh[tid * NW + 1]) + (sh[tid * NW + 2] + sh[tid * NW + 3]);
#ifdef NCDE_DPF_PROF
    DPA_TICK(6)
    if (blockIdx.x == 0 && tid == 0 && phase == DP_FIN && Hr >= 8) {      // the last launch of the solve that does seven stages
        for (int k = 0; k < 7; ++k) Anxt[k] = (float)acc_[k];
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// reverse sweep of a TAPED solve (adjoint=False) on the same stage machinery: ncde_dp_tape_backward (ncde_adaptive.hip) restated
// ------------------------------------------------------------------------------------------------------------------
// Persistent, one workgroup per 16-sample tile, over the accepted steps m = M-1 .. 0.  Per step: the seven stage derivatives are
// recomputed from the recorded start state (forward part of the stage body only), the cotangents of the step's outputs are set up
// (FSAL, transpose of the 4th-order dense output and of its fit, the time partials d/dt0 and d/d(dt_1)), then the six stages are
// transposed in reverse, each a full stage body (forward recompute at the stage input + VJP with the stage's cotangent, ONE gradient
// sum with weight 1).  Lane-local: a lane owns the entries u = 4 (wave NB + nb) + g of sample s for everything elementwise, exactly
// as the attempt kernels do.  Hand-over to ncde_dp_tape_finish (DZ0, F0B, SCB, PN2, the parameter partial) as the per-launch kernel.
template <int H, int HH, int C, int NL>
__global__ __launch_bounds__(256, 1) void ncde_dpf_tape(DpArgs d) {
    typedef DpaPack<H, HH, C, NL> PK;
    constexpr int NW = 4, NT = 256;
    constexpr int CP = PK::CP, CQ = PK::CQ, HB = PK::HB, HT = PK::HT, KH = PK::KH, NB = PK::NB, NTILE = PK::NTILE;
    constexpr int HT0 = H / 16;
    constexpr int XS = 20;
    constexpr int R_Z = 0, R_X = 16, R_XL = R_X + 16 * (NL - 1), R_DP = R_XL + HH, PRIV = (R_DP + HH) * XS;
    constexpr int EPT = (16 * CP + NT - 1) / NT;
    static_assert(HT * HT == NW && HT * HT0 == NW, "one dW1 tile and one dW0 tile per wave");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* zx = lds;
    float* dxs = zx + 2 * H * 16;
    float* d2s = dxs + 2 * 16 * CP;
    float* red = d2s + 2 * 16 * CP;
    float* simg = red + NW * HH * 16;
    float* privbase = simg + PK::IMGS;
    double* sh = reinterpret_cast<double*>(privbase + NW * PRIV);      // [2][NW]
    __shared__ StageDesc sds[8];
    __shared__ double sh_dt;
    __shared__ int sh_jb, sh_je;

    const KArgs& a = d.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = lane & 15, g = lane >> 4;
    const int b0 = blockIdx.x * NCDE_TILE;
    const int bs = b0 + s;
    const bool valid = bs < a.B;
    const int Hr = a.H, HHr = a.dout[0], Cr = a.C;
    const long long BH = (long long)a.B * Hr;
    const bool cubic = a.interp == NCDE_INTERP_CUBIC;
    float* priv = privbase + wave * PRIV;
    float* zimg = priv + R_Z * XS;
    float* xLimg = priv + R_XL * XS;
    float* dpimg = priv + R_DP * XS;
    float* dptile = dpimg;
    const int tr1 = wave / HT, tc1 = wave - tr1 * HT;
    const int tr0 = wave / HT0, tc0 = wave - tr0 * HT0;
    const float* w0i = simg + PK::I_W0 + lane * 4;
    const float* w1i = simg + PK::I_W1 + lane * 4;
    const float* w1ti = simg + PK::I_W1T + lane * 4;
    const float* w0ti = simg + PK::I_W0T + (wave * (KH / 4) * 64 + lane) * 4;
    const float* woTw = simg + PK::I_WOT + wave * NTILE * HT * 256;
    const float* boLw = simg + PK::I_BOL + wave * NTILE * 16;
    const float* b0i = simg + PK::I_B0 + g * 4;
    const float* b1i = simg + PK::I_B1 + g * 4;

    float wo[NB][CQ][KH];
    {
        const float* wp = d.WP + (long long)wave * PK::WAVE + lane * 4;
        float wreg[PK::LANE4 * 4];
#pragma unroll
        for (int i4 = 0; i4 < PK::LANE4; ++i4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(wp + i4 * 256);
#pragma unroll
            for (int q = 0; q < 4; ++q) wreg[4 * i4 + q] = v[q];
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int cq = 0; cq < CQ; ++cq)
#pragma unroll
                for (int ks = 0; ks < KH; ++ks) wo[nb][cq][ks] = wreg[PK::O_WO + (nb * CQ + cq) * KH + ks];
        const f32x4* src = reinterpret_cast<const f32x4*>(d.WP + (long long)NW * PK::WAVE);
        f32x4* dst = reinterpret_cast<f32x4*>(simg);
        for (int e = tid; e < PK::IMGS / 4; e += NT) dst[e] = src[e];
    }
    float dxv[EPT], d2v[EPT];
    auto stage_load = [&](int j) {
        const StageDesc sd = sds[j];
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = tid + q * NT;
            const int es = e / CP, cc = e - es * CP;
            float v = 0.0f, v2 = 0.0f;
            if (e < 16 * CP && cc < Cr && b0 + es < a.B) {
                const float* p = a.coeffs + (long long)(b0 + es) * a.cs_b + (long long)sd.idx * a.cs_t;
                if (!cubic) {
                    v = p[a.cs_t + cc] - p[cc];
                    if (sd.kdt != 1.0f) v = v / sd.kdt;
                } else {
                    const float bb = p[Cr + cc], c2 = p[2 * Cr + cc], dd = p[3 * Cr + cc];
                    const float inner = c2 + dd * sd.frac;
                    v = bb + inner * sd.frac;
                    v2 = inner + dd * sd.frac;
                }
            }
            dxv[q] = v;
            d2v[q] = v2;
        }
    };
    auto stage_store = [&](int slot) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = tid + q * NT;
            if (e < 16 * CP) {
                dxs[(slot & 1) * 16 * CP + e] = dxv[q];
                d2s[(slot & 1) * 16 * CP + e] = d2v[q];
            }
        }
    };

    // ---- gradient accumulators (one sum, weight 1) ---------------------------------------------------------------------------------
    f32x4 gWo[NTILE][HT], gW1, gW0, gbo;
    float gb1 = 0.0f, gb0 = 0.0f;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    gW1 = gW0 = gbo = zero4;
#pragma unroll
    for (int i = 0; i < NTILE; ++i)
#pragma unroll
        for (int t = 0; t < HT; ++t) gWo[i][t] = zero4;

    float zreg[HB], ys[NB], as_[NB];
    int zpar = 0, dpar = 0;
    auto exchange = [&]() {
        float* zw = zx + zpar * H * 16;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) zw[(4 * (wave * NB + nb) + g) * 16 + s] = ys[nb];
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < HB; ++ks) zreg[ks] = zw[(4 * ks + g) * 16 + s];
        zpar ^= 1;
    };
    // One stage evaluation at the stage input now in zreg, with dX/dt in parity slot `dslot`: kout = f dX/dt for the owned entries; with
    // `vjp`: also vy = as_^T df/dy, the parameter gradients (+= as_^T df/dtheta) and vt = as_^T (d/dt of f dX/dt) per lane.
    float kout[NB], vt;
    f32x4 vy;
    auto stage = [&](int dslot, bool vjp) {
        const float* dxp = dxs + (dslot & 1) * 16 * CP + s * CP;
        const float* d2p = d2s + (dslot & 1) * 16 * CP + s * CP;
        float xc[KH];
        unsigned relu_mask = 0;
        {
            f32x4 acc[HT];
#pragma unroll
            for (int tt = 0; tt < HT; ++tt) acc[tt] = *reinterpret_cast<const f32x4*>(b0i + tt * 16);
#pragma unroll
            for (int q4 = 0; q4 < HB / 4; ++q4) {
                f32x4 a4[HT];
#pragma unroll
                for (int tt = 0; tt < HT; ++tt) a4[tt] = *reinterpret_cast<const f32x4*>(w0i + (tt * (HB / 4) + q4) * 256);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) acc[tt] = mfma16(a4[tt][i], zreg[4 * q4 + i], acc[tt]);
            }
#pragma unroll
            for (int tt = 0; tt < HT; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) xc[4 * tt + r] = relu_dev(acc[tt][r]);
#pragma unroll
            for (int l = 1; l < NL; ++l) {
                if (vjp) {
#pragma unroll
                    for (int q4 = 0; q4 < KH / 4; ++q4)
                        if (q4 == tc1) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) priv[((R_X + 16 * (l - 1)) + 4 * i + g) * XS + s] = xc[4 * q4 + i];
                        }
                }
#pragma unroll
                for (int ks = 0; ks < KH; ++ks) relu_mask |= (xc[ks] > 0.0f ? 1u : 0u) << (8 * (l - 1) + ks);
#pragma unroll
                for (int tt = 0; tt < HT; ++tt) acc[tt] = *reinterpret_cast<const f32x4*>(b1i + tt * 16);
#pragma unroll
                for (int q4 = 0; q4 < KH / 4; ++q4) {
                    f32x4 a4[HT];
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) a4[tt] = *reinterpret_cast<const f32x4*>(w1i + (tt * (KH / 4) + q4) * 256);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int tt = 0; tt < HT; ++tt) acc[tt] = mfma16(a4[tt][i], xc[4 * q4 + i], acc[tt]);
                }
#pragma unroll
                for (int tt = 0; tt < HT; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xc[4 * tt + r] = relu_dev(acc[tt][r]);
            }
        }
        f32x4 xB[HT];
        if (vjp) {
#pragma unroll
            for (int q4 = 0; q4 < HB / 4; ++q4)
                if (q4 == tc0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) zimg[(4 * i + g) * XS + s] = zreg[4 * q4 + i];
                }
#pragma unroll
            for (int ks = 0; ks < KH; ++ks) xLimg[(4 * ks + g) * XS + s] = xc[ks];
            wave_lds_order();
#pragma unroll
            for (int tt = 0; tt < HT; ++tt) xB[tt] = *reinterpret_cast<const f32x4*>(xLimg + (16 * tt + s) * XS + 4 * g);
        }
        f32x4 accJ[HT];
#pragma unroll
        for (int tt = 0; tt < HT; ++tt) accJ[tt] = zero4;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) kout[nb] = 0.0f;
        vt = 0.0f;
#pragma unroll
        for (int cq = 0; cq < CQ; ++cq) {
            f32x4 o[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) o[nb] = *reinterpret_cast<const f32x4*>(boLw + ((nb * CQ + cq) * 4 + g) * 4);
#pragma unroll
            for (int ks = 0; ks < KH; ++ks)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) o[nb] = mfma16(wo[nb][cq][ks], xc[ks], o[nb]);
            const f32x4 dx = *reinterpret_cast<const f32x4*>(dxp + 4 * cq);
            const f32x4 d2 = *reinterpret_cast<const f32x4*>(d2p + 4 * cq);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int tau = nb * CQ + cq;
                f32x4 dP;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float m = tanh_prescaled(o[nb][r]);
                    kout[nb] = fmaf(m, dx[r], kout[nb]);
                    dP[r] = (as_[nb] * dx[r]) * (1.0f - m * m);
                    vt = fmaf(as_[nb] * m, d2[r], vt);
                }
                if (vjp) {
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) {
                        const f32x4 av = *reinterpret_cast<const f32x4*>(woTw + ((tau * HT + tt) * 64 + lane) * 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) accJ[tt] = mfma16(av[r], dP[r], accJ[tt]);
                    }
                    f32x4 sm;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sm[r] = row16_sum(dP[r]);
                        dptile[(4 * g + r) * XS + s] = dP[r];
                    }
                    if (s == tau) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) gbo[r] += sm[r];
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
        vy = zero4;
        if (!vjp) return;
        float gpre[KH];
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
            gpre[ks] = xc[ks] > 0.0f ? v : 0.0f;
        }
        auto bias_sum = [&](float& gb) {
#pragma unroll
            for (int ks = 0; ks < KH; ++ks) {
                const float sm = row16_sum(gpre[ks]);
                if (s == ks) gb += sm;
                dpimg[(4 * ks + g) * XS + s] = gpre[ks];
            }
            wave_lds_order();
        };
        auto grad_tile = [&](f32x4& gw, int tr, const float* bimg) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(dpimg + (16 * tr + s) * XS + 4 * g);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bimg + s * XS + 4 * g);
#pragma unroll
            for (int q = 0; q < 4; ++q) gw = mfma16(av[q], bv[q], gw);
            wave_lds_order();
        };
#pragma unroll
        for (int l = NL - 1; l >= 1; --l) {
            bias_sum(gb1);
            grad_tile(gW1, tr1, priv + (R_X + 16 * (l - 1)) * XS);
            f32x4 acc[HT];
#pragma unroll
            for (int tt = 0; tt < HT; ++tt) acc[tt] = zero4;
#pragma unroll
            for (int q4 = 0; q4 < KH / 4; ++q4) {
                f32x4 a4[HT];
#pragma unroll
                for (int tt = 0; tt < HT; ++tt) a4[tt] = *reinterpret_cast<const f32x4*>(w1ti + (tt * (KH / 4) + q4) * 256);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt) acc[tt] = mfma16(a4[tt][i], gpre[4 * q4 + i], acc[tt]);
            }
#pragma unroll
            for (int tt = 0; tt < HT; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) gpre[4 * tt + r] = ((relu_mask >> (8 * (l - 1) + 4 * tt + r)) & 1u) ? acc[tt][r] : 0.0f;
        }
        bias_sum(gb0);
        grad_tile(gW0, tr0, zimg);
#pragma unroll
        for (int q4 = 0; q4 < KH / 4; ++q4) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(w0ti + q4 * 256);
#pragma unroll
            for (int i = 0; i < 4; ++i) vy = mfma16(a4[i], gpre[4 * q4 + i], vy);
        }
    };

    // ---- per-lane state of the sweep ---------------------------------------------------------------------------------------------------
    float y0[NB], kf[7][NB], kb[7][NB], yb0[NB], yb1[NB], YB1[NB], KBN[NB];
    bool own[NB];
    long long gi[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int u = 4 * (wave * NB + nb) + g;
        own[nb] = valid && u < Hr;
        gi[nb] = own[nb] ? (long long)bs * Hr + u : 0;
        YB1[nb] = KBN[nb] = 0.0f;
        as_[nb] = 0.0f;
    }
    double Tpart = 0.0, D1part = 0.0;      // per-thread partials of dL/d(t0 of the steps >= 2) and dL/d(dt_1)
    const int M = d.tape->n_steps;
    float k1_later[NB];      // FSAL: k7 of step m IS k1 of step m + 1, evaluated one iteration ago (six evaluations per step, not seven)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) k1_later[nb] = 0.0f;
    __syncthreads();      // operand images

    for (int m = M - 1; m >= 0; --m) {
        if (tid == 0) {
            const DpStepRec r = d.tape_steps[m];
            const double t1 = r.t0 + r.dt;
            const float t0f = (float)r.t0, dtf0 = (float)r.dt, t1f = (float)t1;
            sh_dt = r.dt;
            sh_jb = r.j_begin;
            sh_je = r.j_end;
            sds[0] = dp_stage_desc(m == 0 ? t0f : nextafterf(t0f, -INFINITY), d.knots, d.n_knots);
            for (int i = 0; i < 6; ++i) {
                const float ti = kAlpha[i] == 1.0f ? nextafterf(t1f, -INFINITY) : t0f + kAlpha[i] * dtf0;
                sds[i + 1] = dp_stage_desc(ti, d.knots, d.n_knots);
            }
        }
        __syncthreads();
        const double dt64 = sh_dt;
        const float dtf = (float)dt64;
        const int jb = sh_jb, je = sh_je;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            y0[nb] = own[nb] ? d.tape_y[(long long)m * BH + gi[nb]] : 0.0f;
            ys[nb] = y0[nb];
#pragma unroll
            for (int q = 0; q < 7; ++q) kf[q][nb] = kb[q][nb] = 0.0f;
        }
        // ---- the seven stage derivatives of the step -----------------------------------------------------------------------------------
        stage_load(0);
        stage_store(dpar);
        exchange();
        float y1[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) y1[nb] = 0.0f;
        const int n_eval = m == M - 1 ? 7 : 6;      // the last step has no later step to take k7 from
#pragma unroll 1
        for (int i = 0; i < n_eval; ++i) {
            if (i + 1 < n_eval) stage_load(i + 1);
            stage(dpar, false);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int q = 0; q < 7; ++q) kf[q][nb] = i == q ? kout[nb] : kf[q][nb];
            if (i + 1 < 7) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    float acc = 0.0f;
#pragma unroll
                    for (int jq = 0; jq < 6; ++jq) acc += kf[jq][nb] * (kBeta[i][jq] * dtf);
                    ys[nb] = own[nb] ? y0[nb] + acc : 0.0f;
                    if (i == 5) y1[nb] = ys[nb];      // the input of the last stage = the step's solution
                }
                if (i + 1 < n_eval) {
                    dpar ^= 1;
                    stage_store(dpar);
                    exchange();
                }
            }
        }
        if (m < M - 1) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) kf[6][nb] = k1_later[nb];
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) k1_later[nb] = kf[0][nb];
        // ---- cotangents: FSAL, dense output, interpolation fit -----------------------------------------------------------------------
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            yb0[nb] = 0.0f;
            yb1[nb] = YB1[nb];
            kb[6][nb] = KBN[nb];
        }
        if (je > jb) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                if (!own[nb]) continue;
                const int u = 4 * (wave * NB + nb) + g;
                const float k1 = kf[0][nb], k7 = kf[6][nb];
                float ym = 0.0f, kmid = 0.0f;
#pragma unroll
                for (int jq = 0; jq < 7; ++jq) {
                    ym += kf[jq][nb] * (dtf * kMid[jq]);
                    kmid += kf[jq][nb] * kMid[jq];
                }
                ym = y0[nb] + ym;
                const float ca = 2.0f * dtf * (k7 - k1) - 8.0f * (y1[nb] + y0[nb]) + 16.0f * ym;
                const float cb = dtf * (5.0f * k1 - 3.0f * k7) + 18.0f * y0[nb] + 14.0f * y1[nb] - 32.0f * ym;
                const float cc = dtf * (k7 - 4.0f * k1) - 11.0f * y0[nb] - 5.0f * y1[nb] + 16.0f * ym;
                const float cd = dtf * k1;
                float ab = 0.f, bb = 0.f, cbb = 0.f, db = 0.f, eb = 0.f;
                for (int jr = jb; jr < je; ++jr) {
                    const float x = d.tape_x[jr];
                    const float gq = d.grad_out[((long long)bs * d.n_t + jr) * Hr + u];
                    const float x2 = x * x, x3 = x2 * x, x4 = x3 * x;
                    eb += gq; db += x * gq; cbb += x2 * gq; bb += x3 * gq; ab += x4 * gq;
                    const double xbar = (double)(gq * (cd + 2.0f * x * cc + 3.0f * x2 * cb + 4.0f * x3 * ca));
                    if (m >= 1) Tpart += xbar * (-1.0 / dt64);
                    else D1part += xbar * (-(double)x / dt64);
                }
                const float ymb = 16.0f * ab - 32.0f * bb + 16.0f * cbb;
                yb0[nb] += -8.0f * ab + 18.0f * bb - 11.0f * cbb + eb + ymb;
                yb1[nb] += -8.0f * ab + 14.0f * bb - 5.0f * cbb;
                kb[0][nb] += dtf * (-2.0f * ab + 5.0f * bb - 4.0f * cbb + db);
                kb[6][nb] += dtf * (2.0f * ab - 3.0f * bb + cbb);
#pragma unroll
                for (int jq = 0; jq < 7; ++jq) kb[jq][nb] += (dtf * kMid[jq]) * ymb;
                if (m == 0)
                    D1part += (double)(ab * 2.0f * (k7 - k1) + bb * (5.0f * k1 - 3.0f * k7) + cbb * (k7 - 4.0f * k1) + db * k1) + (double)(ymb * kmid);
            }
        }
        // ---- the six stages in reverse: k_{i+1} = f(t_i, y_i), y_i = y0 + dt sum_j beta_ij k_j ------------------------------------------
#pragma unroll 1
        for (int i = 6; i >= 1; --i) {
            stage_load(i);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                float acc = 0.0f, cot = 0.0f;
#pragma unroll
                for (int jq = 0; jq < 6; ++jq) acc += kf[jq][nb] * (kBeta[i - 1][jq] * dtf);
#pragma unroll
                for (int q = 0; q < 7; ++q) cot = i == q ? kb[q][nb] : cot;
                ys[nb] = own[nb] ? y0[nb] + acc : 0.0f;
                as_[nb] = own[nb] ? cot : 0.0f;
            }
            dpar ^= 1;
            stage_store(dpar);
            exchange();
            stage(dpar, true);
            if (cubic) {      // dL/d(t_i): t_i = t0 + alpha_i dt, t0 = t[0] + dt_1 + constants
                if (m >= 1) Tpart += (double)vt;
                else D1part += (double)kAlpha[i - 1] * (double)vt;
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float ybi = vy[nb] + (i == 6 ? yb1[nb] : 0.0f);
                yb0[nb] += ybi;
                float sbk = 0.0f;
#pragma unroll
                for (int jq = 0; jq < 6; ++jq) {
                    const float bj = kBeta[i - 1][jq];
                    kb[jq][nb] += (bj * dtf) * ybi;
                    sbk += bj * kf[jq][nb];
                }
                if (m == 0) D1part += (double)(ybi * sbk);
            }
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            YB1[nb] = yb0[nb];
            KBN[nb] = kb[0][nb];
        }
        __syncthreads();      // sds is rewritten at the top of the next step
    }
    // ---- hand-over to the finish launches ------------------------------------------------------------------------------------------------
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        if (!own[nb]) continue;
        const int u = 4 * (wave * NB + nb) + g;
        d.DZ0[gi[nb]] = YB1[nb] + d.grad_out[((long long)bs * d.n_t) * Hr + u];      // the solution at t[0] is z0 itself
        d.F0B[gi[nb]] = KBN[nb];
        d.SCB[gi[nb]] = 0.0f;
    }
    const double Tw = wave_sum_d(Tpart), Dw = wave_sum_d(D1part);
    __syncthreads();
    if (lane == 0) { sh[wave] = Tw; sh[NW + wave] = Dw; }
    __syncthreads();
    if (tid == 0) {
        d.PN2[(long long)blockIdx.x * 4 + 0] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        d.PN2[(long long)blockIdx.x * 4 + 1] = (sh[NW] + sh[NW + 1]) + (sh[NW + 2] + sh[NW + 3]);
        d.PN2[(long long)blockIdx.x * 4 + 2] = 0.0;
        d.PN2[(long long)blockIdx.x * 4 + 3] = 0.0;
    }
    // the parameter partial in parameter order (once per solve): ncde_dp_tape_finish adds its own evaluations on top
    float* gp = d.GP + (long long)blockIdx.x * d.theta1;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int cq = 0; cq < CQ; ++cq) {
            const int tau = nb * CQ + cq;
            const int hh = 4 * (wave * NB + nb) + g;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cc = 4 * cq + r;
                if (hh < Hr && cc < Cr) {
#pragma unroll
                    for (int tt = 0; tt < HT; ++tt)
                        if (16 * tt + s < HHr) gp[a.gWo_off + (long long)(hh * Cr + cc) * HHr + 16 * tt + s] = gWo[tau][tt][r];
                    if (s == tau) gp[a.gbo_off + hh * Cr + cc] = gbo[r];
                }
            }
        }
    if (NL > 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (16 * tr1 + 4 * g + r < HHr && 16 * tc1 + s < HHr) gp[a.gW_off[1] + (16 * tr1 + 4 * g + r) * HHr + 16 * tc1 + s] = gW1[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (16 * tr0 + 4 * g + r < HHr && 16 * tc0 + s < Hr) gp[a.gW_off[0] + (16 * tr0 + 4 * g + r) * Hr + 16 * tc0 + s] = gW0[r];
    if (wave == 0 && s < KH && 4 * s + g < HHr) {
        if (NL > 1) gp[a.gb_off[1] + 4 * s + g] = gb1;
        gp[a.gb_off[0] + 4 * s + g] = gb0;
    }
    if (tid == 0) gp[a.theta_size] = 0.0f;
}

// Sum of the per-workgroup parameter-part partials, the parameter part of the mixed norm, and -- in the last workgroup -- the controller.
// A block handles RED_EPB consecutive entries of the PARTIAL vector ([dWo in register order][parameter order], see ncde_dpf_adj); its
// RED_NG thread groups each sum their share of the workgroups (fixed order, coalesced 256-byte rows), the quarters are combined in a fixed order.
#ifndef NCDE_RED_EPB
#define NCDE_RED_EPB 128
#endif
constexpr int RED_EPB = NCDE_RED_EPB, RED_NG = 256 / RED_EPB;      // entries per block, groups of workgroups per block
extern "C" __global__ __launch_bounds__(256) void ncde_dpf_reduce(DpArgs d) {
    typedef DpaPack<32, 32, 20, 3> PK;
    __shared__ double sh[256];
    __shared__ float part[2][RED_NG][RED_EPB];
    __shared__ int sh_flags[2], is_last;
    __shared__ float sh_x;
    DpCtrl* c = d.ctrl;
    const int phase = c->phase;
    if (phase == DP_DONE || c->error != 0) return;
    const KArgs& a = d.a;
    const int tid = threadIdx.x, grp = tid / RED_EPB, kk = tid % RED_EPB;
    const int PL = PK::WOT + d.theta1;
    const int e = blockIdx.x * RED_EPB + kk;
    const float rtolf = (float)d.rtol, atolf = (float)d.atol;
    const int cur = c->cur;
    const float* G0 = cur ? d.GCT : d.G0T;
    float* G1 = cur ? d.G0T : d.GCT;
    const int nset = phase == DP_STEP ? 2 : 1;
    // parameter index of the entry (-1: padding of the register-order block, or the unused dWo range of the parameter-order block)
    int k = -1;
    if (e < PK::WOT) {
        const int r = e & 3, lane = (e >> 2) & 63, rest = e >> 8;
        const int tt = rest % PK::HT, tau = (rest / PK::HT) % PK::NTILE, wv = rest / (PK::HT * PK::NTILE);
        const int nb = tau / PK::CQ, cq = tau - nb * PK::CQ;
        const int hh = 4 * (wv * PK::NB + nb) + (lane >> 4), cc = 4 * cq + r, jj = 16 * tt + (lane & 15);
        if (hh < a.H && cc < a.C && jj < a.dout[0]) k = a.gWo_off + (hh * a.C + cc) * a.dout[0] + jj;
    } else if (e < PL) {
        const int kq = e - PK::WOT;
        if (kq < a.gWo_off || kq >= a.gWo_off + a.H * a.C * a.dout[0]) k = kq;
    }
    for (int q = 0; q < nset; ++q) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (k >= 0) {
            const float* src = d.GP + (long long)q * PL + e;
            const long long row = 2LL * PL;
            int p = grp;
            for (; p + 15 * RED_NG < d.n_wg; p += 16 * RED_NG) {      // sixteen rows in flight; the order of the additions is fixed
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = src[(long long)(p + RED_NG * u) * row];
#pragma unroll
                for (int u = 0; u < 16; u += 4) { s0 += v[u]; s1 += v[u + 1]; s2 += v[u + 2]; s3 += v[u + 3]; }
            }
            for (; p < d.n_wg; p += RED_NG) s0 += src[(long long)p * row];
        }
        part[q][grp][kk] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    float qa = 0.0f, qb = 0.0f;
    int seg = -1;
    if (grp == 0 && k >= 0) {
        float tot[2] = {0.0f, 0.0f};
        for (int q = 0; q < nset; ++q) {
            float t4[RED_NG / 2];
#pragma unroll
            for (int u = 0; u < RED_NG / 2; ++u) t4[u] = part[q][2 * u][kk] + part[q][2 * u + 1][kk];
            float tq = t4[0];
#pragma unroll
            for (int u = 1; u < RED_NG / 2; ++u) tq += t4[u];
            tot[q] = tq;
        }
        const float g0 = G0[k];
        if (phase == DP_INIT0) {
            d.KT[k] = tot[0];
            const float scale = atolf + fabsf(g0) * rtolf;
            qa = g0 / scale;
            qb = tot[0] / scale;
        } else if (phase == DP_INIT1) {
            const float scale = atolf + fabsf(g0) * rtolf;
            qa = (tot[0] - d.KT[k]) / scale;
        } else if (phase == DP_STEP) {
            const float g1 = g0 + tot[0];
            G1[k] = g1;
            qa = tot[1] / (atolf + rtolf * fmaxf(fabsf(g0), fabsf(g1)));
        } else {
            G1[k] = g0 + tot[0];
        }
        for (int sg = 0; sg <= d.nseg; ++sg) {
            const int off = sg < d.nseg ? d.seg_off[sg] : d.theta1 - 1;
            const int len = sg < d.nseg ? d.seg_len[sg] : 1;
            if (k >= off && k < off + len) seg = sg;
        }
    }
    // per-segment sums of squares of this block -> SEGP[block][2][DP_MAXSEG + 1]  (wave 0 holds the terms; a block of RED_EPB entries
    // touches few segments, so: one wave-wide sum per segment that occurs)
    double* segp = d.SEGP + (long long)blockIdx.x * 2 * (DP_MAXSEG + 1);
    constexpr int RED_W0 = (RED_EPB + 63) / 64;      // waves that hold group 0's terms (the other lanes hold seg = -1)
    __shared__ double segw[RED_W0][2][DP_MAXSEG + 1];
    if (tid < 64 * RED_W0) {
        for (int sg = 0; sg <= d.nseg; ++sg) {
            double va = 0.0, vb = 0.0;
            if (__builtin_amdgcn_ballot_w64(seg == sg) != 0) {
                va = wave_sum_d(seg == sg ? (double)qa * qa : 0.0);
                vb = wave_sum_d(seg == sg ? (double)qb * qb : 0.0);
            }
            if ((tid & 63) == 0) { segw[tid >> 6][0][sg] = va; segw[tid >> 6][1][sg] = vb; }
        }
    }
    __syncthreads();
    if (tid <= d.nseg) {
        double va = 0.0, vb = 0.0;
        for (int w = 0; w < RED_W0; ++w) { va += segw[w][0][tid]; vb += segw[w][1][tid]; }
        __hip_atomic_store(&segp[tid], va, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&segp[(DP_MAXSEG + 1) + tid], vb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    dpf_finish_launch(d, sh, sh_flags, &sh_x, &is_last);
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
namespace {
typedef void (*DpfKernel)(DpArgs);

int dpf_shape(const NcdeProblem* p) {      // 0: none, 1: <32, 32, 20>, 2: <64, 64, 4>
    if (p->field_kind != NCDE_FIELD_ORIGINAL || p->field_input != NCDE_INPUT_MATMUL) return 0;
    if (p->n_layers < 1) return 0;
    const int HH = p->layer_out[0];
    if (p->layer_in[0] != p->hidden) return 0;
    for (int l = 1; l < p->n_layers; ++l)
        if (p->layer_out[l] != HH || p->layer_in[l] != HH || p->layer_W[l] != p->layer_W[1] || p->layer_b[l] != p->layer_b[1]) return 0;
    if (p->hidden <= 32 && HH <= 32 && p->channels <= 20) return 1;
    if (p->hidden <= 64 && HH <= 64 && p->channels <= 4) return 2;
    return 0;
}
}  // namespace

namespace {
typedef DpaPack<32, 32, 20, 3> Dpa32;      // (the pack layout does not depend on NL)
size_t dpa_lds_bytes(int nl) {
    const size_t priv = (size_t)(16 + 16 * (nl - 1) + 32 + 32) * 20;
    return 4 * (size_t)(2 * 32 * 16 + 4 * 16 * 20 + 4 * 32 * 16 + Dpa32::IMGS + 4 * priv) + 16 * sizeof(double);
}
template <int NL>
DpfKernel dpa_kernel() { return (DpfKernel)ncde_dpf_adj<32, 32, 20, NL>; }
DpfKernel dpa_pick(int nl) { return nl == 1 ? dpa_kernel<1>() : (nl == 2 ? dpa_kernel<2>() : (nl == 3 ? dpa_kernel<3>() : nullptr)); }      // (NL = 4 does not fit the 160 KB of LDS)
}  // namespace

bool ncde_dpf_supported(const NcdeProblem* p, int adj) {
    if (p->flags & NCDE_FLAG_FORCE_GENERIC) return false;
    const int sh = dpf_shape(p);
    if (adj == 0) return sh != 0;
    if (adj == 1) return sh == 1 && p->n_layers <= 3 && dpa_lds_bytes(p->n_layers) + 512 <= 160 * 1024;      // (32, 32, 20) set, <= 3 layers (include/ncde_hip.h); 512 B: static LDS
    return false;
}

const char* ncde_dpf_kernel_name(const NcdeProblem* p, int adj) {
    if (!ncde_dpf_supported(p, adj)) return nullptr;
    if (adj) return "ncde_dpf_adj<H32,HH32,C20,fp32 MFMA> + ncde_dpf_reduce";
    return dpf_shape(p) == 1 ? "ncde_dpf_fwd<H32,HH32,C20,bf16x3>" : "ncde_dpf_fwd<H64,HH64,C4,bf16x3>";
}

// fused adjoint: floats of one workgroup's partial vector (one weighted sum), and the number of blocks of ncde_dpf_reduce
size_t ncde_dpf_partial_floats(const NcdeProblem* p, int theta1) { return ncde_dpf_supported(p, 1) ? (size_t)Dpa32::WOT + (size_t)theta1 : (size_t)theta1; }
int ncde_dpf_reduce_blocks(const NcdeProblem* p, int theta1) { return (int)((ncde_dpf_partial_floats(p, theta1) + NCDE_RED_EPB - 1) / NCDE_RED_EPB); }

size_t ncde_dpf_pack_floats(const NcdeProblem* p, int adj) {
    if (!ncde_dpf_supported(p, adj)) return 0;
    if (adj) return (size_t)Dpa32::TOTAL;
    return dpf_shape(p) == 1 ? (size_t)DpfPack<32, 32, 20>::TOTAL : (size_t)DpfPack<64, 64, 4>::TOTAL;
}

// once per solve, before the first attempt: the per-lane weight image
int ncde_dpf_prepare(const NcdeProblem* p, const void* dp_args, size_t dp_args_bytes, int adj, hipStream_t st) {
    if (dp_args_bytes != sizeof(DpArgs) || !ncde_dpf_supported(p, adj)) return NCDE_ERR_UNSUPPORTED;
    DpArgs d;
    memcpy(&d, dp_args, sizeof(d));
    if (adj) {
        hipLaunchKernelGGL((ncde_dpa_pack<32, 32, 20, 3>), dim3(1), dim3(256), 0, st, d);
        const size_t lds = dpa_lds_bytes(p->n_layers);
        if (hipFuncSetAttribute((const void*)dpa_pick(p->n_layers), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return NCDE_ERR_HIP;
    } else if (dpf_shape(p) == 1) {
        hipLaunchKernelGGL((ncde_dpf_pack<32, 32, 20>), dim3(1), dim3(256), 0, st, d);
    } else {
        hipLaunchKernelGGL((ncde_dpf_pack<64, 64, 4>), dim3(1), dim3(256), 0, st, d);
    }
    return hipGetLastError() == hipSuccess ? NCDE_OK : NCDE_ERR_HIP;
}

// reverse sweep of a taped solve: weight image + the persistent sweep (the caller launches ncde_dp_tape_finish afterwards, as before)
bool ncde_dpf_tape_supported(const NcdeProblem* p) { return ncde_dpf_supported(p, 1); }
const char* ncde_dpf_tape_kernel_name(const NcdeProblem* p) { return ncde_dpf_tape_supported(p) ? "ncde_dpf_tape<H32,HH32,C20,fp32 MFMA>" : nullptr; }
int ncde_dpf_tape_launch(const NcdeProblem* p, const void* dp_args, size_t dp_args_bytes, hipStream_t st) {
    if (dp_args_bytes != sizeof(DpArgs) || !ncde_dpf_tape_supported(p)) return NCDE_ERR_UNSUPPORTED;
    DpArgs d;
    memcpy(&d, dp_args, sizeof(d));
    const int nl = p->n_layers;
    const DpfKernel k = nl == 1 ? (DpfKernel)ncde_dpf_tape<32, 32, 20, 1> : (nl == 2 ? (DpfKernel)ncde_dpf_tape<32, 32, 20, 2>
                                                                             : (DpfKernel)ncde_dpf_tape<32, 32, 20, 3>);      // nl <= 3 (ncde_dpf_supported)
    const size_t lds = dpa_lds_bytes(nl);
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return NCDE_ERR_HIP;
    hipLaunchKernelGGL((ncde_dpa_pack<32, 32, 20, 3>), dim3(1), dim3(256), 0, st, d);
    hipLaunchKernelGGL(k, dim3(d.n_wg), dim3(256), lds, st, d);
    return hipGetLastError() == hipSuccess ? NCDE_OK : NCDE_ERR_HIP;
}

int ncde_dpf_launch(const NcdeProblem* p, const void* dp_args, size_t dp_args_bytes, int adj, int rounds, hipStream_t st) {
    if (dp_args_bytes != sizeof(DpArgs) || !ncde_dpf_supported(p, adj)) return NCDE_ERR_UNSUPPORTED;
    DpArgs d;
    memcpy(&d, dp_args, sizeof(d));
    if (adj) {
        const DpfKernel k = dpa_pick(p->n_layers);
        const size_t lds = dpa_lds_bytes(p->n_layers);
        for (int r = 0; r < rounds; ++r) {
            hipLaunchKernelGGL(k, dim3(d.n_wg), dim3(256), lds, st, d);
            hipLaunchKernelGGL(ncde_dpf_reduce, dim3(d.n_rblk), dim3(256), 0, st, d);      // n_rblk = ceil(partial length / 64)
        }
    } else {
        const DpfKernel k = dpf_shape(p) == 1 ? (DpfKernel)ncde_dpf_fwd<32, 32, 20> : (DpfKernel)ncde_dpf_fwd<64, 64, 4>;
        for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(k, dim3(d.n_wg), dim3(256), 0, st, d);
    }
    return hipGetLastError() == hipSuccess ? NCDE_OK : NCDE_ERR_HIP;
}
