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
#else
#define DPF_STAMP(k)
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
    // ---- dX/dt of every stage of the attempt -> LDS [stage][sample][channel] ------------------------------------------------
    const int nst = phase == DP_STEP ? 6 : 1;
    for (int e = tid; e < nst * 16 * CP; e += NT) {
        const int j = e / (16 * CP), rem = e - j * (16 * CP), es = rem / CP, cc = rem - es * CP;
        float v = 0.0f;
        if (cc < Cr && b0 + es < a.B) {
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
        dxq[j][es * CP + cc] = v;
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

bool ncde_dpf_supported(const NcdeProblem* p, int adj) {
    if (p->flags & NCDE_FLAG_FORCE_GENERIC) return false;
    if (adj != 0) return false;
    return dpf_shape(p) != 0;
}

const char* ncde_dpf_kernel_name(const NcdeProblem* p, int adj) {
    if (!ncde_dpf_supported(p, adj)) return nullptr;
    return dpf_shape(p) == 1 ? "ncde_dpf_fwd<H32,HH32,C20,bf16x3>" : "ncde_dpf_fwd<H64,HH64,C4,bf16x3>";
}

size_t ncde_dpf_pack_floats(const NcdeProblem* p) {
    const int sh = dpf_shape(p);
    return sh == 1 ? (size_t)DpfPack<32, 32, 20>::TOTAL : (sh == 2 ? (size_t)DpfPack<64, 64, 4>::TOTAL : 0);
}

// once per solve, before the first attempt: the per-lane weight image
int ncde_dpf_prepare(const NcdeProblem* p, const void* dp_args, size_t dp_args_bytes, int adj, hipStream_t st) {
    if (dp_args_bytes != sizeof(DpArgs) || !ncde_dpf_supported(p, adj)) return NCDE_ERR_UNSUPPORTED;
    DpArgs d;
    memcpy(&d, dp_args, sizeof(d));
    if (dpf_shape(p) == 1) hipLaunchKernelGGL((ncde_dpf_pack<32, 32, 20>), dim3(1), dim3(256), 0, st, d);
    else hipLaunchKernelGGL((ncde_dpf_pack<64, 64, 4>), dim3(1), dim3(256), 0, st, d);
    return hipGetLastError() == hipSuccess ? NCDE_OK : NCDE_ERR_HIP;
}

int ncde_dpf_launch(const NcdeProblem* p, const void* dp_args, size_t dp_args_bytes, int adj, int rounds, hipStream_t st) {
    if (dp_args_bytes != sizeof(DpArgs) || !ncde_dpf_supported(p, adj)) return NCDE_ERR_UNSUPPORTED;
    DpArgs d;
    memcpy(&d, dp_args, sizeof(d));
    const DpfKernel k = dpf_shape(p) == 1 ? (DpfKernel)ncde_dpf_fwd<32, 32, 20> : (DpfKernel)ncde_dpf_fwd<64, 64, 4>;
    for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(k, dim3(d.n_wg), dim3(256), 0, st, d);
    return hipGetLastError() == hipSuccess ? NCDE_OK : NCDE_ERR_HIP;
}
