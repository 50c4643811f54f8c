// Adaptive Dormand-Prince 5(4) ("dopri5") for the Neural-CDE path: NeuralCDE(solver="dopri5") (src/ncde/ncde.py:129-134) and
// the default method of cdeint (experiments/sim_bm_toy_example.py:54-57).  Restates, relative to
// /root/reference/modules/torchdiffeq/torchdiffeq/_impl:
//   dopri5.py:5-36            tableau, mid-point weights
//   rk_common.py:41-86        one step + embedded error estimate (stage times in the STATE dtype; alpha = 1 stages just before t1)
//   rk_common.py:216-305      step control: fp64 time, accept iff error ratio <= 1, min_step / max_step overrides
//   misc.py:33-103            initial step, error ratio, next step size
//   interp.py:4-61            4th-order dense output at the requested times (steps are NOT clipped to output times)
//   misc.py:18-19             ONE rms norm over the whole batch state: every sample takes the same steps
//   adjoint.py:37-145, 235-247  one adaptive reverse solve per output interval over (vjp_t, y, a, g_theta), mixed norm
//
// Why this is not the persistent per-tile loop of the fixed-step kernels: accept / reject is a property of the WHOLE batch
// (one error norm), so every attempt needs a grid-wide reduction before anybody may move on.  The solve is therefore a
// sequence of small launches per attempt -- six stage launches (one workgroup per 16-sample tile, the generic family's
// stage code), a controller (one workgroup: norms, accept / reject, next dt, stage descriptors of the next attempt, all
// time arithmetic in fp64 on the device) and a commit (dense output, state roll) -- with the state resident in HBM
// between launches.  The host only counts rounds: it enqueues a batch of rounds (kernels of finished solves exit at
// once) and looks at the controller's phase word between batches -- the trip count is data dependent, exactly as the
// reference's Python `while next_t > self.rk_state.t1` is.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "ncde_adaptive.h"
#include "ncde_adaptive_fast.h"
#include "ncde_generic_stage.h"
#include "ncde_host.h"

#include "ncde_dp_defs.h"

namespace {
// second derivative of the cubic spline for the tile -> D2X[c*16 + s] (the t-dependence autograd sends to vjp_t)
__device__ void dp_load_d2x(const KArgs& a, int b0, const StageDesc& sd, float* D2X, int Cp, int tid) {
    for (int e = tid; e < 16 * Cp; e += GEN_THREADS) {
        const int s = e / Cp, c = e - s * Cp;
        const int b = b0 + s;
        float v = 0.0f;
        if (c < a.C && b < a.B) {
            const float* p = a.coeffs + (long long)b * a.cs_b + (long long)sd.idx * a.cs_t;
            const float cc = p[2 * a.C + c], dd = p[3 * a.C + c];
            const float inner = cc + dd * sd.frac;
            v = inner + dd * sd.frac;
        }
        D2X[c * 16 + s] = v;
    }
}

}  // namespace


// ------------------------------------------------------------------------------------------------------------------
// init: state <- initial condition, controller <- start of the (first) solve
// ------------------------------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(256) void ncde_dp_init(DpArgs d) {
    const long long n = (long long)d.a.B * d.a.H;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int H = d.a.H;
    if (gid < n) {
        const int b = (int)(gid / H), h = (int)(gid - (long long)b * H);
        if (!d.adj) {
            const float v = d.a.z0[gid];
            d.Y0[gid] = v;
            d.out[((long long)b * d.n_t) * H + h] = v;
        } else {
            const long long o = ((long long)b * d.n_t + (d.n_t - 1)) * H + h;
            d.Y0[gid] = d.z_out[o];
            d.A0[gid] = d.grad_out[o];
        }
    }
    if (d.adj)
        for (long long e = gid; e < d.theta1; e += (long long)gridDim.x * 256) d.G0T[e] = 0.0f;
    if (gid == 0) {
        DpCtrl* c = d.ctrl;
        memset(c, 0, sizeof(DpCtrl));
        c->phase = DP_INIT0;
        c->j_out = 1;
        if (!d.adj) {
            c->t0 = d.t_out[0];
        } else {
            c->interval = d.n_t - 1;
            c->t0 = -d.t_out[d.n_t - 1];
            c->t_goal = -d.t_out[d.n_t - 2];
        }
        c->t1 = c->t0;
        c->st[0] = dp_stage_desc(d.adj ? -(float)c->t0 : (float)c->t0, d.knots, d.n_knots);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// one stage evaluation for every 16-sample tile
// ------------------------------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(GEN_THREADS) void ncde_dp_stage(DpArgs d, int slot) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const DpCtrl* c = d.ctrl;
    const int phase = c->phase;
    if (phase == DP_DONE || c->error != 0 || (phase != DP_STEP && slot != 1)) return;
    const KArgs& a = d.a;
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * NCDE_TILE;
    const int H = a.H, Hp = ru16(H), Cp = ru4(a.C), L = a.n_layers;
    int Dp = Hp;
    for (int l = 0; l < L; ++l) Dp = max(Dp, ru16(a.dout[l]));
    const int HS = Hp * 16, DS = Dp * 16;
    const float dtf = c->dtf, h0 = c->h0;
    const StageDesc sd = c->st[phase == DP_STEP ? slot : 0];
    const long long BH = (long long)a.B * H;
    // stage index in the K planes: INIT0 -> plane 0 (f0), INIT1 -> plane 1 (probe, scratch), STEP -> plane `slot`
    const int plane = phase == DP_INIT0 ? 0 : (phase == DP_INIT1 ? 1 : slot);
    const float rtolf = (float)d.rtol, atolf = (float)d.atol;
    double sum0 = 0.0, sum1 = 0.0, sum2 = 0.0, sum3 = 0.0;   // per-thread partial sums of squares (y / a parts)

    if (!d.adj) {
        float* YS = lds;
        float* ACT0 = YS + HS;
        float* ACT1 = ACT0 + DS;
        float* KO = ACT1 + DS;
        float* DX = KO + HS;
        for (int e = tid; e < HS + 2 * DS + HS + Cp * 16; e += GEN_THREADS) lds[e] = 0.0f;
        __syncthreads();
        for (int e = tid; e < HS; e += GEN_THREADS) {
            const int h = e >> 4, s = e & 15, b = b0 + s;
            if (h < H && b < a.B) {
                const long long g = (long long)b * H + h;
                float y = d.Y0[g];
                if (phase == DP_INIT1) y = y + h0 * d.KY[g];
                else if (phase == DP_STEP) {
                    float acc = 0.0f;
                    for (int j = 0; j < slot; ++j) acc += d.KY[j * BH + g] * (kBeta[slot - 1][j] * dtf);
                    y = y + acc;
                }
                YS[e] = y;
            }
        }
        load_dx(a, b0, sd, DX, Cp, tid);
        __syncthreads();
        gen_stage_forward(a, YS, ACT0, ACT1, DX, KO, Hp, Cp, tid);
        for (int e = tid; e < HS; e += GEN_THREADS) {
            const int h = e >> 4, s = e & 15, b = b0 + s;
            if (h < H && b < a.B) {
                const long long g = (long long)b * H + h;
                const float k = KO[e];
                d.KY[plane * BH + g] = k;
                if (phase == DP_INIT0) {
                    const float y0 = d.Y0[g];
                    const float scale = atolf + fabsf(y0) * rtolf;
                    const float q0 = y0 / scale, q1 = k / scale;
                    sum0 += (double)q0 * q0;
                    sum1 += (double)q1 * q1;
                } else if (phase == DP_INIT1) {
                    const float y0 = d.Y0[g];
                    const float scale = atolf + fabsf(y0) * rtolf;
                    const float q = (k - d.KY[g]) / scale;
                    sum0 += (double)q * q;
                } else if (slot == 6) {
                    const float y0 = d.Y0[g], y1 = YS[e];
                    d.YC[g] = y1;     // c_sol = (beta[-1], 0): the input of the last stage IS the solution (rk_common.py:76-80)
                    float err = 0.0f;
                    for (int j = 0; j < 6; ++j) err += d.KY[j * BH + g] * (dtf * kCErr[j]);
                    err += k * (dtf * kCErr[6]);
                    const float tol = atolf + rtolf * fmaxf(fabsf(y0), fabsf(y1));
                    const float q = err / tol;
                    sum0 += (double)q * q;
                }
            }
        }
    } else {
        float* YS = lds;
        float* AS = YS + HS;
        float* KOY = AS + HS;
        float* KOA = KOY + HS;
        float* X = KOA + HS;
        float* G0 = X + (L > 0 ? L : 1) * DS;
        float* G1 = G0 + DS;
        float* PW2 = G1 + DS;
        float* DX = PW2 + 2 * DS;
        float* D2X = DX + Cp * 16;
        float* SC = D2X + Cp * 16;
        float* GL = SC + GEN_NW * 16 * 17;
        const int total = 4 * HS + ((L > 0 ? L : 1) + 4) * DS + 2 * Cp * 16 + GEN_NW * 16 * 17 + (d.gacc_in_lds ? d.theta1 : 0);
        for (int e = tid; e < total; e += GEN_THREADS) lds[e] = 0.0f;
        float* gacc = d.gacc_in_lds ? GL : d.GP + (long long)blockIdx.x * d.theta1;
        if (!d.gacc_in_lds)
            for (int e = tid; e < d.theta1; e += GEN_THREADS) gacc[e] = 0.0f;
        __syncthreads();
        for (int e = tid; e < HS; e += GEN_THREADS) {
            const int h = e >> 4, s = e & 15, b = b0 + s;
            if (h < H && b < a.B) {
                const long long g = (long long)b * H + h;
                float y = d.Y0[g], av = d.A0[g];
                if (phase == DP_INIT1) {
                    y = y + h0 * d.KY[g];
                    av = av + h0 * d.KA[g];
                } else if (phase == DP_STEP) {
                    float accy = 0.0f, acca = 0.0f;
                    for (int j = 0; j < slot; ++j) {
                        const float bj = kBeta[slot - 1][j] * dtf;
                        accy += d.KY[j * BH + g] * bj;
                        acca += d.KA[j * BH + g] * bj;
                    }
                    y = y + accy;
                    av = av + acca;
                }
                YS[e] = y;
                AS[e] = av;
            }
        }
        load_dx(a, b0, sd, DX, Cp, tid);
        const bool cubic = a.interp == NCDE_INTERP_CUBIC;
        if (cubic) dp_load_d2x(a, b0, sd, D2X, Cp, tid);
        __syncthreads();
        gen_stage_vjp(a, YS, AS, DX, cubic ? D2X : nullptr, X, G0, G1, PW2, SC, KOY, KOA, gacc, 1.0f, Hp, Cp, DS, tid);
        __syncthreads();
        if (d.gacc_in_lds) {
            float* dst = d.GP + (long long)blockIdx.x * d.theta1;
            for (int e = tid; e < d.theta1; e += GEN_THREADS) dst[e] = GL[e];
        }
        for (int e = tid; e < HS; e += GEN_THREADS) {
            const int h = e >> 4, s = e & 15, b = b0 + s;
            if (h < H && b < a.B) {
                const long long g = (long long)b * H + h;
                const float ky = -KOY[e], ka = KOA[e];      // negated time: dy/ds = -f, da/ds = +a^T df/dy (misc.py:152-159)
                d.KY[plane * BH + g] = ky;
                d.KA[plane * BH + g] = ka;
                const float y0 = d.Y0[g], a0 = d.A0[g];
                if (phase == DP_INIT0) {
                    const float sy = atolf + fabsf(y0) * rtolf, sa = atolf + fabsf(a0) * rtolf;
                    const float q0 = y0 / sy, q1 = a0 / sa, q2 = ky / sy, q3 = ka / sa;
                    sum0 += (double)q0 * q0; sum1 += (double)q1 * q1; sum2 += (double)q2 * q2; sum3 += (double)q3 * q3;
                } else if (phase == DP_INIT1) {
                    const float sy = atolf + fabsf(y0) * rtolf, sa = atolf + fabsf(a0) * rtolf;
                    const float q0 = (ky - d.KY[g]) / sy, q1 = (ka - d.KA[g]) / sa;
                    sum0 += (double)q0 * q0; sum1 += (double)q1 * q1;
                } else if (slot == 6) {
                    const float y1 = YS[e], a1 = AS[e];
                    d.YC[g] = y1;
                    d.AC[g] = a1;
                    float ey = 0.0f, ea = 0.0f;
                    for (int j = 0; j < 6; ++j) {
                        const float cj = dtf * kCErr[j];
                        ey += d.KY[j * BH + g] * cj;
                        ea += d.KA[j * BH + g] * cj;
                    }
                    ey += ky * (dtf * kCErr[6]);
                    ea += ka * (dtf * kCErr[6]);
                    const float qy = ey / (atolf + rtolf * fmaxf(fabsf(y0), fabsf(y1)));
                    const float qa = ea / (atolf + rtolf * fmaxf(fabsf(a0), fabsf(a1)));
                    sum0 += (double)qy * qy; sum1 += (double)qa * qa;
                }
            }
        }
    }
    // workgroup reduction of the partial sums -> PN[wg][0..3]
    __shared__ double red[4][GEN_THREADS];
    red[0][tid] = sum0; red[1][tid] = sum1; red[2][tid] = sum2; red[3][tid] = sum3;
    __syncthreads();
    for (int off = GEN_THREADS / 2; off > 0; off >>= 1) {
        if (tid < off)
            for (int q = 0; q < 4; ++q) red[q][tid] += red[q][tid + off];
        __syncthreads();
    }
    if (tid < 4) d.PN[(long long)blockIdx.x * 4 + tid] = red[tid][0];
}

// parameter part of the stage derivative: deterministic sum of the per-workgroup partials (K4 of the fixed-step path)
extern "C" __global__ __launch_bounds__(256) void ncde_dp_reduce_theta(DpArgs d, int slot) {
    const DpCtrl* c = d.ctrl;
    const int phase = c->phase;
    if (phase == DP_DONE || c->error != 0 || (phase != DP_STEP && slot != 1)) return;
    const int plane = phase == DP_INIT0 ? 0 : (phase == DP_INIT1 ? 1 : slot);
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= d.theta1) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int p = 0;
    for (; p + 3 < d.n_wg; p += 4) {
        s0 += d.GP[(long long)p * d.theta1 + k];
        s1 += d.GP[(long long)(p + 1) * d.theta1 + k];
        s2 += d.GP[(long long)(p + 2) * d.theta1 + k];
        s3 += d.GP[(long long)(p + 3) * d.theta1 + k];
    }
    for (; p < d.n_wg; ++p) s0 += d.GP[(long long)p * d.theta1 + k];
    d.KT[(long long)plane * d.theta1 + k] = (s0 + s1) + (s2 + s3);
}

// ------------------------------------------------------------------------------------------------------------------
// controller: one workgroup.  Norms over the whole batch, accept / reject, next step, phase transitions.
// ------------------------------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(256) void ncde_dp_control(DpArgs d) {
    __shared__ double sh[256];
    __shared__ int sh_flags[2];
    __shared__ float sh_x;
    dp_control_body(d, sh, sh_flags, &sh_x);
}

// ------------------------------------------------------------------------------------------------------------------
// commit of an accepted step: dense output at the requested times, state roll, FSAL
// ------------------------------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(256) void ncde_dp_commit(DpArgs d) {
    const DpCtrl* c = d.ctrl;
    if (!c->accepted_now) return;
    const long long n = (long long)d.a.B * d.a.H;
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= n) return;
    const int H = d.a.H;
    const int b = (int)(g / H), h = (int)(g - (long long)b * H);
    const float dtc = c->dtf_commit;
    const float y0 = d.Y0[g], y1 = d.YC[g];
    const float f0 = d.KY[g], f1 = d.KY[6 * n + g];
    if (!d.adj) {
        if (d.tape_y && c->n_accept - 1 < d.tape_cap) d.tape_y[(long long)(c->n_accept - 1) * n + g] = y0;
        if (c->j_end > c->j_begin) {
            float ym = 0.0f;
            for (int j = 0; j < 7; ++j) ym += d.KY[j * n + g] * (dtc * kMid[j]);
            ym = y0 + ym;
            for (int j = c->j_begin; j < c->j_end; ++j) d.out[((long long)b * d.n_t + j) * H + h] = dp_poly(y0, y1, ym, f0, f1, dtc, d.XOUT[j]);
        }
        d.Y0[g] = y1;
        d.KY[g] = f1;
        return;
    }
    const float a0 = d.A0[g], a1 = d.AC[g];
    const float fa0 = d.KA[g], fa1 = d.KA[6 * n + g];
    if (c->finish_now) {      // adjoint.py:131-133: a at the interval end (dense output), y reset to the stored solution
        float am = 0.0f;
        for (int j = 0; j < 7; ++j) am += d.KA[j * n + g] * (dtc * kMid[j]);
        am = a0 + am;
        const long long o = ((long long)b * d.n_t + c->row_now) * H + h;
        d.A0[g] = dp_poly(a0, a1, am, fa0, fa1, dtc, c->x_end) + d.grad_out[o];
        d.Y0[g] = d.z_out[o];
    } else {
        d.Y0[g] = y1;
        d.A0[g] = a1;
        d.KY[g] = f1;
        d.KA[g] = fa1;
    }
}


// ------------------------------------------------------------------------------------------------------------------
// reverse sweep of a TAPED solve: cdeint(..., method='dopri5', adjoint=False) -- the way the reference's shipped "interpolation"
// experiments run dopri5 (experiments/configurations/configurations.json5:187-191).  What autograd differentiates there
// (rk_common.py:216-305 under torchdiffeq.odeint): the six stage evaluations of every ACCEPTED step with FSAL (k1 of a step is k7
// of the previous one), the 4th-order dense output at the requested times (interp.py:4-61), and -- _optimal_step_size being
// @torch.no_grad() (misc.py:84-97) -- of all step sizes only the FIRST, _select_initial_step(y0, f0, f(t0 + h0, y0 + h0 f0))
// (misc.py:33-74), which moves the start time of every later step (dense-output abscissae, stage times of a cubic control).
// Samples couple only through that one scalar, so the sweep over the recorded steps is ONE persistent launch, one workgroup per
// 16-sample tile, no grid-wide synchronisation inside: per step the stage derivatives are recomputed from the recorded state
// (7 evaluations), then six stage VJPs in reverse; the scalar partials d/d(t0), d/d(dt_1) are reduced afterwards and pushed
// through the initial-step rule by two small launches (ncde_dp_tape_finish).  Oracle: ncde_oracle.dopri5_discrete_backward.
// ------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int DP_NE = 8;      // elements of an [H <= 128][16] tile array per thread (GEN_THREADS = 256)

struct TapeLds {
    float *YS, *AS, *KOY, *KOA, *X, *G0, *G1, *PW2, *DX, *D2X, *SC, *Y0s, *YB1, *KBN, *GL;
    int total;
};
__device__ TapeLds tape_lds(float* lds, int HS, int DS, int L, int Cp, int theta1, int gacc_in_lds) {
    TapeLds t;
    t.YS = lds; t.AS = t.YS + HS; t.KOY = t.AS + HS; t.KOA = t.KOY + HS;
    t.X = t.KOA + HS; t.G0 = t.X + (L > 0 ? L : 1) * DS; t.G1 = t.G0 + DS; t.PW2 = t.G1 + DS;
    t.DX = t.PW2 + 2 * DS; t.D2X = t.DX + Cp * 16; t.SC = t.D2X + Cp * 16;
    t.Y0s = t.SC + GEN_NW * 16 * 17; t.YB1 = t.Y0s + HS; t.KBN = t.YB1 + HS; t.GL = t.KBN + HS;
    t.total = (int)(t.GL - lds) + (gacc_in_lds ? theta1 : 0);
    return t;
}
// block-wide sum of a double over GEN_THREADS threads; every thread gets the total
__device__ double tape_block_sum(double v, double* sh) {
    const int tid = threadIdx.x;
    __syncthreads();
    sh[tid] = v;
    __syncthreads();
    for (int off = GEN_THREADS / 2; off > 0; off >>= 1) {
        if (tid < off) sh[tid] += sh[tid + off];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}
}  // namespace

extern "C" __global__ __launch_bounds__(GEN_THREADS) void ncde_dp_tape_backward(DpArgs d) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ StageDesc sdesc[7];
    __shared__ double sh_dt;
    __shared__ int sh_jb, sh_je;
    __shared__ double shred[GEN_THREADS];
    const KArgs& a = d.a;
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * NCDE_TILE;
    const int H = a.H, Hp = ru16(H), Cp = ru4(a.C), L = a.n_layers;
    int Dp = Hp;
    for (int l = 0; l < L; ++l) Dp = max(Dp, ru16(a.dout[l]));
    const int HS = Hp * 16, DS = Dp * 16;
    const TapeLds t = tape_lds(lds, HS, DS, L, Cp, d.theta1, d.gacc_in_lds);
    for (int e = tid; e < t.total; e += GEN_THREADS) lds[e] = 0.0f;
    float* gacc = d.gacc_in_lds ? t.GL : d.GP + (long long)blockIdx.x * d.theta1;
    if (!d.gacc_in_lds)
        for (int e = tid; e < d.theta1; e += GEN_THREADS) gacc[e] = 0.0f;
    float* KF = d.KF + (long long)blockIdx.x * 7 * HS;
    float* KB = d.KBR + (long long)blockIdx.x * 7 * HS;
    const bool cubic = a.interp == NCDE_INTERP_CUBIC;
    const long long BH = (long long)a.B * H;
    const int M = d.tape->n_steps;
    double Tpart = 0.0, D1part = 0.0;      // per-thread partials of dL/d(t0 of the steps >= 2) and dL/d(dt_1)
    __syncthreads();

    for (int m = M - 1; m >= 0; --m) {
        if (tid == 0) {
            const DpStepRec r = d.tape_steps[m];
            const double t1 = r.t0 + r.dt;
            const float t0f = (float)r.t0, dtf = (float)r.dt, t1f = (float)t1;
            sh_dt = r.dt;
            sh_jb = r.j_begin;
            sh_je = r.j_end;
            // k1 of the step: f0 = f(t[0], y0) for the first step, else the previous step's last stage, evaluated one ulp before its t1
            sdesc[0] = dp_stage_desc(m == 0 ? t0f : nextafterf(t0f, -INFINITY), d.knots, d.n_knots);
            for (int i = 0; i < 6; ++i) {
                const float ti = kAlpha[i] == 1.0f ? nextafterf(t1f, -INFINITY) : t0f + kAlpha[i] * dtf;
                sdesc[i + 1] = dp_stage_desc(ti, d.knots, d.n_knots);
            }
        }
        __syncthreads();
        const double dt64 = sh_dt;
        const float dtf = (float)dt64;
        const int jb = sh_jb, je = sh_je;
        // ---- recompute the stage derivatives of the step ------------------------------------------------------------
        for (int e = tid; e < HS; e += GEN_THREADS) {
            const int h = e >> 4, sidx = e & 15, b = b0 + sidx;
            t.Y0s[e] = (h < H && b < a.B) ? d.tape_y[(long long)m * BH + (long long)b * H + h] : 0.0f;
        }
        load_dx(a, b0, sdesc[0], t.DX, Cp, tid);
        __syncthreads();
        gen_stage_forward(a, t.Y0s, t.X, t.G0, t.DX, t.KOY, Hp, Cp, tid);
        for (int e = tid; e < HS; e += GEN_THREADS) KF[e] = t.KOY[e];
        for (int i = 1; i <= 6; ++i) {
            for (int e = tid; e < HS; e += GEN_THREADS) {
                const int h = e >> 4, sidx = e & 15, b = b0 + sidx;
                float acc = 0.0f;
                for (int j = 0; j < i; ++j) acc += KF[j * HS + e] * (kBeta[i - 1][j] * dtf);
                t.YS[e] = (h < H && b < a.B) ? t.Y0s[e] + acc : 0.0f;
            }
            load_dx(a, b0, sdesc[i], t.DX, Cp, tid);
            __syncthreads();
            gen_stage_forward(a, t.YS, t.X, t.G0, t.DX, t.KOY, Hp, Cp, tid);
            for (int e = tid; e < HS; e += GEN_THREADS) KF[i * HS + e] = t.KOY[e];
        }
        // ---- cotangents: FSAL, dense output, interpolation fit ------------------------------------------------------
        float yb0[DP_NE], yb1[DP_NE];
#pragma unroll
        for (int q = 0; q < DP_NE; ++q) {
            const int e = tid + q * GEN_THREADS;
            yb0[q] = 0.0f;
            yb1[q] = 0.0f;
            if (e < HS) {
                yb1[q] = t.YB1[e];
                for (int j = 0; j < 6; ++j) KB[j * HS + e] = 0.0f;
                KB[6 * HS + e] = t.KBN[e];
            }
        }
        if (je > jb) {
#pragma unroll
            for (int q = 0; q < DP_NE; ++q) {
                const int e = tid + q * GEN_THREADS;
                const int h = e >> 4, sidx = e & 15, b = b0 + sidx;
                if (e < HS && h < H && b < a.B) {
                    const float y0 = t.Y0s[e], y1 = t.YS[e];      // YS still holds the input of the last stage = the step's solution
                    const float k1 = KF[e], k7 = KF[6 * HS + e];
                    float ym = 0.0f, kmid = 0.0f;
                    for (int j = 0; j < 7; ++j) {
                        ym += KF[j * HS + e] * (dtf * kMid[j]);
                        kmid += KF[j * HS + e] * kMid[j];
                    }
                    ym = y0 + ym;
                    const float ca = 2.0f * dtf * (k7 - k1) - 8.0f * (y1 + y0) + 16.0f * ym;
                    const float cb = dtf * (5.0f * k1 - 3.0f * k7) + 18.0f * y0 + 14.0f * y1 - 32.0f * ym;
                    const float cc = dtf * (k7 - 4.0f * k1) - 11.0f * y0 - 5.0f * y1 + 16.0f * ym;
                    const float cd = dtf * k1;
                    float ab = 0.f, bb = 0.f, cbb = 0.f, db = 0.f, eb = 0.f;
                    for (int j = jb; j < je; ++j) {
                        const float x = d.tape_x[j];
                        const float g = d.grad_out[((long long)b * d.n_t + j) * H + h];
                        const float x2 = x * x, x3 = x2 * x, x4 = x3 * x;
                        eb += g; db += x * g; cbb += x2 * g; bb += x3 * g; ab += x4 * g;
                        const double xbar = (double)(g * (cd + 2.0f * x * cc + 3.0f * x2 * cb + 4.0f * x3 * ca));
                        if (m >= 1) Tpart += xbar * (-1.0 / dt64);             // x = (t - t0) / dt_m, t0 moves with dt_1
                        else D1part += xbar * (-(double)x / dt64);            // x = (t - t[0]) / dt_1
                    }
                    const float ymb = 16.0f * ab - 32.0f * bb + 16.0f * cbb;
                    yb0[q] += -8.0f * ab + 18.0f * bb - 11.0f * cbb + eb + ymb;
                    yb1[q] += -8.0f * ab + 14.0f * bb - 5.0f * cbb;
                    KB[e] += dtf * (-2.0f * ab + 5.0f * bb - 4.0f * cbb + db);
                    KB[6 * HS + e] += dtf * (2.0f * ab - 3.0f * bb + cbb);
                    for (int j = 0; j < 7; ++j) KB[j * HS + e] += (dtf * kMid[j]) * ymb;
                    if (m == 0)
                        D1part += (double)(ab * 2.0f * (k7 - k1) + bb * (5.0f * k1 - 3.0f * k7) + cbb * (k7 - 4.0f * k1) + db * k1) + (double)(ymb * kmid);
                }
            }
        }
        // ---- the six stages in reverse: k_{i+1} = f(t_i, y_i), y_i = y0 + dt sum_j beta_ij k_j -------------------------
        for (int i = 6; i >= 1; --i) {
            for (int e = tid; e < HS; e += GEN_THREADS) {
                const int h = e >> 4, sidx = e & 15, b = b0 + sidx;
                const bool ok = h < H && b < a.B;
                float acc = 0.0f;
                for (int j = 0; j < i; ++j) acc += KF[j * HS + e] * (kBeta[i - 1][j] * dtf);
                t.YS[e] = ok ? t.Y0s[e] + acc : 0.0f;
                t.AS[e] = ok ? KB[i * HS + e] : 0.0f;
            }
            load_dx(a, b0, sdesc[i], t.DX, Cp, tid);
            if (cubic) dp_load_d2x(a, b0, sdesc[i], t.D2X, Cp, tid);
            __syncthreads();
            gen_stage_vjp(a, t.YS, t.AS, t.DX, cubic ? t.D2X : nullptr, t.X, t.G0, t.G1, t.PW2, t.SC, t.KOY, t.KOA, gacc, 1.0f, Hp, Cp, DS, tid);
            __syncthreads();
            const float tbi = cubic ? gacc[a.theta_size] : 0.0f;     // dL/d(t_i), summed over the tile
            __syncthreads();
            if (tid == 0) {
                if (cubic) gacc[a.theta_size] = 0.0f;
                if (m >= 1) Tpart += (double)tbi;                   // t_i = t0 + alpha_i dt, t0 = t[0] + dt_1 + constants
                else D1part += (double)kAlpha[i - 1] * (double)tbi;
            }
#pragma unroll
            for (int q = 0; q < DP_NE; ++q) {
                const int e = tid + q * GEN_THREADS;
                if (e < HS) {
                    const float ybi = t.KOA[e] + (i == 6 ? yb1[q] : 0.0f);
                    yb0[q] += ybi;
                    float sbk = 0.0f;
                    for (int j = 0; j < i; ++j) {
                        KB[j * HS + e] += (kBeta[i - 1][j] * dtf) * ybi;
                        sbk += kBeta[i - 1][j] * KF[j * HS + e];
                    }
                    if (m == 0) D1part += (double)(ybi * sbk);
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < DP_NE; ++q) {
            const int e = tid + q * GEN_THREADS;
            if (e < HS) {
                t.YB1[e] = yb0[q];
                t.KBN[e] = KB[e];
            }
        }
        __syncthreads();
    }
    // ---- hand-over to the finish launches -----------------------------------------------------------------------------
    for (int e = tid; e < HS; e += GEN_THREADS) {
        const int h = e >> 4, sidx = e & 15, b = b0 + sidx;
        if (h < H && b < a.B) {
            const long long g = (long long)b * H + h;
            d.DZ0[g] = t.YB1[e] + d.grad_out[((long long)b * d.n_t) * H + h];      // the solution at t[0] is z0 itself
            d.F0B[g] = t.KBN[e];
            d.SCB[g] = 0.0f;
        }
    }
    const double Tsum = tape_block_sum(Tpart, shred), Dsum = tape_block_sum(D1part, shred);
    if (tid == 0) {
        d.PN2[(long long)blockIdx.x * 4 + 0] = Tsum;
        d.PN2[(long long)blockIdx.x * 4 + 1] = Dsum;
        d.PN2[(long long)blockIdx.x * 4 + 2] = 0.0;
        d.PN2[(long long)blockIdx.x * 4 + 3] = 0.0;
    }
    if (d.gacc_in_lds) {
        float* dst = d.GP + (long long)blockIdx.x * d.theta1;
        for (int e = tid; e < d.theta1; e += GEN_THREADS) dst[e] = t.GL[e];
    }
}

// phase 1: dL/d(dt_1) through the probe evaluation f(t0 + h0, y0 + h0 f0) of the initial-step rule; phase 2: through d0, d1 and
// the scale, and the VJP of f0 = f(t[0], y0) with everything that landed on it (misc.py:33-74; oracle: dopri5_discrete_backward)
extern "C" __global__ __launch_bounds__(GEN_THREADS) void ncde_dp_tape_finish(DpArgs d, int phase) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ StageDesc sd0, sd1;
    __shared__ double shred[GEN_THREADS];
    const KArgs& a = d.a;
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * NCDE_TILE;
    const int H = a.H, Hp = ru16(H), Cp = ru4(a.C), L = a.n_layers;
    int Dp = Hp;
    for (int l = 0; l < L; ++l) Dp = max(Dp, ru16(a.dout[l]));
    const int HS = Hp * 16, DS = Dp * 16;
    const TapeLds t = tape_lds(lds, HS, DS, L, Cp, d.theta1, d.gacc_in_lds);
    for (int e = tid; e < t.total; e += GEN_THREADS) lds[e] = 0.0f;
    float* gpart = d.GP + (long long)blockIdx.x * d.theta1;
    float* gacc = d.gacc_in_lds ? t.GL : gpart;
    const bool cubic = a.interp == NCDE_INTERP_CUBIC;
    const DpTapeHeader hd = *d.tape;
    const bool active = hd.delta_active != 0;
    const float rtolf = (float)d.rtol, atolf = (float)d.atol;
    const double n_el = (double)a.B * (double)H;
    // batch-wide scalars (every workgroup forms them itself: n_wg doubles)
    double p0 = 0.0, p2 = 0.0, p3 = 0.0;
    for (int w = tid; w < d.n_wg; w += GEN_THREADS) {
        p0 += d.PN2[(long long)w * 4 + 0] + d.PN2[(long long)w * 4 + 1];
        p2 += d.PN2[(long long)w * 4 + 2];
        p3 += d.PN2[(long long)w * 4 + 3];
    }
    const double dbar = tape_block_sum(p0, shred);                 // dL/d(dt_1)
    const double s2 = tape_block_sum(p2, shred), s3 = tape_block_sum(p3, shred);
    const double h0 = hd.h0, h1 = hd.h1, d0 = hd.d0, d1 = hd.d1, d2 = hd.d2;
    double h0b = 0.0, h1b = 0.0, d1b = 0.0, d2b = 0.0;
    if (active) {
        if ((float)(100.0f * hd.h0) <= hd.h1) h0b = 100.0 * dbar;
        else h1b = dbar;
        if (h1b != 0.0 && !hd.h1_const) {
            const double mx = fmax(d1, d2);
            const double mxb = -0.2 * h1 / mx * h1b;
            if (d1 >= d2) d1b += mxb;
            else d2b += mxb;
        } else if (h1b != 0.0) {
            if (hd.h0 * 1e-3f > 1e-6f) h0b += 1e-3 * h1b;
        }
    }
    if (tid == 0) {
        const float t0f = (float)hd.t_start;
        sd0 = dp_stage_desc(t0f, d.knots, d.n_knots);
        sd1 = dp_stage_desc(t0f + hd.h0, d.knots, d.n_knots);
    }
    __syncthreads();
    // y0 and f0 = f(t[0], y0) for the tile
    for (int e = tid; e < HS; e += GEN_THREADS) {
        const int h = e >> 4, sidx = e & 15, b = b0 + sidx;
        t.Y0s[e] = (h < H && b < a.B) ? a.z0[(long long)b * H + h] : 0.0f;
    }
    load_dx(a, b0, sd0, t.DX, Cp, tid);
    __syncthreads();
    gen_stage_forward(a, t.Y0s, t.X, t.G0, t.DX, t.KOY, Hp, Cp, tid);
    for (int e = tid; e < HS; e += GEN_THREADS) t.YB1[e] = t.KOY[e];       // f0
    __syncthreads();
    if (phase == 1) {
        if (!(active && d2b != 0.0)) return;
        const double n2 = d2 * h0;
        for (int e = tid; e < HS; e += GEN_THREADS) {
            const int h = e >> 4, sidx = e & 15, b = b0 + sidx;
            t.YS[e] = (h < H && b < a.B) ? t.Y0s[e] + hd.h0 * t.YB1[e] : 0.0f;     // y1' = y0 + h0 f0
        }
        load_dx(a, b0, sd1, t.DX, Cp, tid);
        __syncthreads();
        gen_stage_forward(a, t.YS, t.X, t.G0, t.DX, t.KOY, Hp, Cp, tid);          // f1' = f(t0 + h0, y1')
        for (int e = tid; e < HS; e += GEN_THREADS) {
            const int h = e >> 4, sidx = e & 15, b = b0 + sidx;
            float f1b = 0.0f;
            if (h < H && b < a.B) {
                const long long g = (long long)b * H + h;
                const float scale = atolf + fabsf(t.Y0s[e]) * rtolf;
                const float q = (t.KOY[e] - t.YB1[e]) / scale;
                const float qb = (float)((d2b / h0) * (double)q / (n_el * n2));
                f1b = qb / scale;
                d.F0B[g] -= qb / scale;
                d.SCB[g] -= qb * q / scale;
            }
            t.AS[e] = f1b;
        }
        if (cubic) dp_load_d2x(a, b0, sd1, t.D2X, Cp, tid);
        __syncthreads();
        gen_stage_vjp(a, t.YS, t.AS, t.DX, cubic ? t.D2X : nullptr, t.X, t.G0, t.G1, t.PW2, t.SC, t.KOY, t.KOA, gacc, 1.0f, Hp, Cp, DS, tid);
        __syncthreads();
        double part = 0.0;
        for (int e = tid; e < HS; e += GEN_THREADS) {
            const int h = e >> 4, sidx = e & 15, b = b0 + sidx;
            if (h < H && b < a.B) {
                const long long g = (long long)b * H + h;
                const float dy1 = t.KOA[e];
                d.DZ0[g] += dy1;
                d.F0B[g] += hd.h0 * dy1;
                part += (double)(dy1 * t.YB1[e]);
            }
        }
        const double psum = tape_block_sum(part, shred);
        if (tid == 0) {
            d.PN2[(long long)blockIdx.x * 4 + 2] = psum;
            d.PN2[(long long)blockIdx.x * 4 + 3] = cubic ? (double)gacc[a.theta_size] : 0.0;
        }
    } else {
        double d0b = 0.0;
        if (active) {
            if (d2b != 0.0) h0b += -d2 / h0 * d2b + s2 + s3;
            if (h0b != 0.0 && !hd.h0_const) {
                d0b = 0.01 / d1 * h0b;
                d1b += -h0 / d1 * h0b;
            }
        }
        for (int e = tid; e < HS; e += GEN_THREADS) {
            const int h = e >> 4, sidx = e & 15, b = b0 + sidx;
            float f0b = 0.0f;
            if (h < H && b < a.B) {
                const long long g = (long long)b * H + h;
                const float y0 = t.Y0s[e];
                const float scale = atolf + fabsf(y0) * rtolf;
                f0b = d.F0B[g];
                float scb = d.SCB[g], dz = d.DZ0[g];
                if (d0b != 0.0) {
                    const float q0 = y0 / scale;
                    const float q0b = (float)(d0b * (double)q0 / (n_el * d0));
                    dz += q0b / scale;
                    scb -= q0b * q0 / scale;
                }
                if (d1b != 0.0) {
                    const float q1 = t.YB1[e] / scale;
                    const float q1b = (float)(d1b * (double)q1 / (n_el * d1));
                    f0b += q1b / scale;
                    scb -= q1b * q1 / scale;
                }
                dz += scb * rtolf * (y0 > 0.0f ? 1.0f : (y0 < 0.0f ? -1.0f : 0.0f));
                d.DZ0[g] = dz;
            }
            t.AS[e] = f0b;
        }
        __syncthreads();
        gen_stage_vjp(a, t.Y0s, t.AS, t.DX, nullptr, t.X, t.G0, t.G1, t.PW2, t.SC, t.KOY, t.KOA, gacc, 1.0f, Hp, Cp, DS, tid);
        __syncthreads();
        for (int e = tid; e < HS; e += GEN_THREADS) {
            const int h = e >> 4, sidx = e & 15, b = b0 + sidx;
            if (h < H && b < a.B) {
                const long long g = (long long)b * H + h;
                d.gz0[g] = d.DZ0[g] + t.KOA[e];
            }
        }
    }
    if (d.gacc_in_lds) {       // this launch's parameter part on top of the partial the sweep left
        __syncthreads();
        for (int e = tid; e < d.theta1; e += GEN_THREADS) gpart[e] += t.GL[e];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
namespace {

struct DpPlan {
    size_t off_ctrl, off_t, off_knots, off_xout, off_y0, off_yc, off_ky, off_a0, off_ac, off_ka, off_gp, off_kt, off_g0t, off_gct, off_pn, off_tr, off_wp, off_segp;
    size_t total;
    int theta1, n_wg;
};

size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
constexpr int kTraceCap = 8192;      // attempts the diagnostic trace can hold

DpPlan dp_plan(const NcdeProblem* p, const Layout& y, int n_t, bool adj) {
    DpPlan w{};
    size_t o = 0;
    const size_t BH = (size_t)p->batch * p->hidden;
    w.theta1 = y.theta_size + 1;
    w.n_wg = y.n_wg;
    auto take = [&](size_t bytes) { const size_t at = o; o = al256(o + bytes); return at; };
    w.off_ctrl = take(sizeof(DpCtrl));
    w.off_t = take(sizeof(double) * n_t);
    w.off_knots = take(sizeof(float) * p->n_knots);
    w.off_xout = take(sizeof(float) * n_t);
    w.off_y0 = take(4 * BH);
    w.off_yc = take(4 * BH);
    w.off_ky = take(4 * 7 * BH);
    w.off_pn = take(sizeof(double) * 4 * w.n_wg);
    w.off_tr = take(sizeof(double) * 4 * kTraceCap + sizeof(double) * 2 * kTraceCap);      // trace rows, then the replay list
    w.off_wp = take(4 * ncde_dpf_pack_floats(p, adj ? 1 : 0));      // fused attempt kernels: per-lane weight image (0 bytes where they do not apply)
    if (adj) {
        w.off_a0 = take(4 * BH);
        w.off_ac = take(4 * BH);
        w.off_ka = take(4 * 7 * BH);
        // fused: two weighted sums per workgroup, each [dWo in register order][parameter order]
        w.off_gp = take(ncde_dpf_supported(p, 1) ? 4 * (size_t)w.n_wg * 2 * ncde_dpf_partial_floats(p, w.theta1) : 4 * (size_t)w.n_wg * w.theta1);
        w.off_segp = take(sizeof(double) * 2 * (DP_MAXSEG + 1) * (size_t)ncde_dpf_reduce_blocks(p, w.theta1));
        w.off_kt = take(4 * 7 * (size_t)w.theta1);
        w.off_g0t = take(4 * (size_t)w.theta1);
        w.off_gct = take(4 * (size_t)w.theta1);
    }
    w.total = o;
    return w;
}

}  // namespace

namespace {
// device record of a taped solve: header | steps[cap] | x[n_t] | y[cap][B][H]
struct TapeLayout {
    size_t off_steps, off_x, off_y, total;
    int cap;
};
TapeLayout tape_layout(const NcdeProblem* p, int n_t, int cap) {
    TapeLayout t{};
    const size_t BH = (size_t)p->batch * p->hidden;
    t.cap = cap;
    t.off_steps = sizeof(DpTapeHeader);
    t.off_x = al256(t.off_steps + sizeof(DpStepRec) * (size_t)cap);
    t.off_y = al256(t.off_x + sizeof(float) * (size_t)n_t);
    t.total = t.off_y + 4 * BH * (size_t)cap;
    return t;
}
// accepted steps a solve can take: every step but the first is >= min_step long and starts before t[-1]
int tape_default_cap(const NcdeProblem* p, const NcdeTimeSpec* ts, const NcdeAdaptiveOptions* op) {
    const double span = ts->t[ts->n_t - 1] - ts->t[0];
    // torchdiffeq counts max_num_steps per output interval (rk_common.py:232): no solve that finishes takes more steps than this
    const double by_max = op->max_num_steps > 0 ? (double)op->max_num_steps * (ts->n_t - 1) + 4.0 : 1.0e6;
    if (op->min_step > 0.0) return (int)std::min(std::min(1.0e6, by_max), std::ceil(span / op->min_step) + 4.0);
    // no lower bound on the step: a default of about two steps per knot; a solve that needs more returns NCDE_ERR_WORKSPACE ("pass a
    // larger record") and the caller repeats it with a larger one -- the forward is deterministic (ncde_amd/solver.py doubles it)
    return (int)std::min(by_max, 2.0 * p->n_knots + 2.0 * ts->n_t + 256.0);
}
// the largest capacity a record of `bytes` holds
int tape_cap_of(const NcdeProblem* p, int n_t, size_t bytes) {
    int lo = 0, hi = 1 << 24;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tape_layout(p, n_t, mid).total <= bytes) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

struct TapeWs {
    size_t off_knots, off_kf, off_kb, off_dz0, off_f0b, off_scb, off_gp, off_pn2, off_wp, total;
};
TapeWs tape_ws(const NcdeProblem* p, const Layout& y) {
    TapeWs w{};
    size_t o = 0;
    const size_t BH = (size_t)p->batch * p->hidden;
    auto take = [&](size_t bytes) { const size_t at = o; o = al256(o + bytes); return at; };
    w.off_knots = take(sizeof(float) * p->n_knots);
    w.off_kf = take(4 * (size_t)y.n_wg * 7 * y.HS);
    w.off_kb = take(4 * (size_t)y.n_wg * 7 * y.HS);
    w.off_dz0 = take(4 * BH);
    w.off_f0b = take(4 * BH);
    w.off_scb = take(4 * BH);
    w.off_gp = take(4 * (size_t)y.n_wg * (y.theta_size + 1));
    w.off_pn2 = take(sizeof(double) * 4 * y.n_wg);
    w.off_wp = take(4 * ncde_dpf_pack_floats(p, 1));      // fused sweep (ncde_dpf_tape): weight image (0 bytes where it does not apply)
    w.total = o;
    return w;
}
}  // namespace

int64_t ncde_dp_workspace_bytes(const NcdeProblem* p, int n_t, int adj) {
    const Layout y = make_layout(p);
    if (adj == 2) return (int64_t)tape_ws(p, y).total;      // reverse sweep of a taped solve
    return (int64_t)dp_plan(p, y, n_t, adj != 0).total;
}

int64_t ncde_dp_record_bytes(const NcdeProblem* p, const NcdeTimeSpec* ts, const NcdeAdaptiveOptions* op) {
    return (int64_t)tape_layout(p, ts->n_t, tape_default_cap(p, ts, op)).total;
}

bool ncde_dp_supported(const NcdeProblem* p, int adj, char* why, size_t n) {
    const Layout y = make_layout(p);
    if (y.variant) { snprintf(why, n, "dopri5 runs the original field with the matmul input only"); return false; }
    if (!adj && y.lds_fwd > (size_t)kLdsLimit) { snprintf(why, n, "dopri5 forward needs %zu B of LDS", y.lds_fwd); return false; }
    if (adj) {
        const size_t base = sizeof(float) * (size_t)(4 * y.HS + ((y.L > 0 ? y.L : 1) + 4) * y.DS + 2 * y.Cp * 16 + 4 * 16 * 17) + 4 * 256 * sizeof(double);
        if (base > (size_t)kLdsLimit) { snprintf(why, n, "dopri5 adjoint needs %zu B of LDS", base); return false; }
        if (y.dlast > 128) { snprintf(why, n, "dopri5 adjoint supports a last hidden width <= 128 (got %d)", y.dlast); return false; }
    }
    return true;
}

// The reverse sweep of a taped solve (ncde_dp_tape_backward) keeps its cotangents in per-thread arrays of DP_NE elements
// (element e = tid + 256 q < 16 H, i.e. H <= 128) and needs 7 [H][16] arrays of LDS where the stage kernels need 4.  Checked when
// the record is SIZED and when the recording forward is launched, so that a model outside these bounds fails before any work is
// done instead of producing silently wrong gradients (rows h >= 128 never initialised) or failing only in backward().
bool ncde_dp_tape_supported(const NcdeProblem* p, char* why, size_t n) {
    const Layout y = make_layout(p);
    if (y.Hp > 16 * DP_NE) { snprintf(why, n, "dopri5 with adjoint=False supports hidden <= %d (got %d)", 16 * DP_NE, p->hidden); return false; }
    if (y.dlast > 128) { snprintf(why, n, "dopri5 with adjoint=False supports a last hidden width <= 128 (got %d)", y.dlast); return false; }
    const size_t lds = sizeof(float) * (size_t)(7 * y.HS + ((y.L > 0 ? y.L : 1) + 4) * y.DS + 2 * y.Cp * 16 + 4 * 16 * 17) + 4 * 256 * sizeof(double);
    if (lds > (size_t)kLdsLimit) { snprintf(why, n, "dopri5 with adjoint=False needs %zu B of LDS for its reverse sweep (> %d)", lds, kLdsLimit); return false; }
    return true;
}

// Runs the solve to completion (this call synchronises: the number of attempts is data dependent).
int ncde_dp_solve(const NcdeProblem* p, const NcdeTimeSpec* ts, const NcdeAdaptiveOptions* op, int adj, float* out, const float* z_out,
                  const float* grad_out, const NcdeGrads* g, void* ws, size_t ws_bytes, hipStream_t st, NcdeAdaptiveStats* stats,
                  char* err, size_t errn, void* record, size_t record_bytes, const double* replay, int replay_n) {
    const Layout y = make_layout(p);
    const int n_t = ts->n_t;
    const DpPlan w = dp_plan(p, y, n_t, adj != 0);
    if (ws_bytes < w.total) { snprintf(err, errn, "workspace %zu B < %zu B", ws_bytes, w.total); return NCDE_ERR_WORKSPACE; }
    for (int i = 1; i < n_t; ++i)
        if (!(ts->t[i] > ts->t[i - 1])) { snprintf(err, errn, "t must be strictly increasing (decreasing output times are outside the fused path)"); return NCDE_ERR_INVALID; }
    char* base = (char*)ws;
    DpArgs d;
    memset(&d, 0, sizeof(d));
    fill_kargs(p, y, &d.a);
    d.a.n_out = n_t;
    d.ctrl = (DpCtrl*)(base + w.off_ctrl);
    d.adj = adj; d.n_wg = w.n_wg; d.n_t = n_t; d.n_knots = p->n_knots; d.theta1 = w.theta1;
    d.t_out = (const double*)(base + w.off_t);
    d.knots = ts->knots ? (const float*)(base + w.off_knots) : nullptr;
    d.XOUT = (float*)(base + w.off_xout);
    d.Y0 = (float*)(base + w.off_y0); d.YC = (float*)(base + w.off_yc); d.KY = (float*)(base + w.off_ky);
    d.PN = (double*)(base + w.off_pn);
    d.TR = (double*)(base + w.off_tr);
    d.WP = (float*)(base + w.off_wp);
    if (adj) { d.SEGP = (double*)(base + w.off_segp); d.n_rblk = ncde_dpf_reduce_blocks(p, w.theta1); }
    d.trace_cap = op->trace && op->trace_capacity > 0 ? std::min(op->trace_capacity, kTraceCap) : 0;
    if (adj) {
        d.A0 = (float*)(base + w.off_a0); d.AC = (float*)(base + w.off_ac); d.KA = (float*)(base + w.off_ka);
        d.GP = (float*)(base + w.off_gp); d.KT = (float*)(base + w.off_kt);
        d.G0T = (float*)(base + w.off_g0t); d.GCT = (float*)(base + w.off_gct);
    }
    d.out = out; d.z_out = z_out; d.grad_out = grad_out;
    if (record) {      // taped forward (adjoint=False): keep what the reverse sweep needs of every accepted step
        if (adj) { snprintf(err, errn, "a record is kept by the forward solve only"); return NCDE_ERR_INVALID; }
        const int cap = tape_cap_of(p, n_t, record_bytes);
        if (cap < 1) { snprintf(err, errn, "record of %zu B holds no step (ncde_dopri5_record_bytes)", record_bytes); return NCDE_ERR_WORKSPACE; }
        const TapeLayout tl = tape_layout(p, n_t, cap);
        char* rb = (char*)record;
        d.tape = (DpTapeHeader*)rb;
        d.tape_steps = (DpStepRec*)(rb + tl.off_steps);
        d.tape_x = (float*)(rb + tl.off_x);
        d.tape_y = (float*)(rb + tl.off_y);
        d.tape_cap = cap;
        DpTapeHeader hh;
        memset(&hh, 0, sizeof(hh));
        hh.magic = DP_TAPE_MAGIC; hh.cap = cap; hh.n_t = n_t;
        hipError_t e0 = hipMemcpyAsync(rb, &hh, sizeof(hh), hipMemcpyHostToDevice, st);
        if (e0 == hipSuccess) e0 = hipStreamSynchronize(st);      // hh is a local
        if (e0 != hipSuccess) { snprintf(err, errn, "record header upload failed: %s", hipGetErrorString(e0)); return NCDE_ERR_HIP; }
    }
    d.rtol = op->rtol; d.atol = op->atol; d.min_step = op->min_step; d.max_step = op->max_step > 0.0 ? op->max_step : INFINITY;
    d.first_step = op->first_step;
    d.safety = op->safety > 0.0 ? op->safety : 0.9;
    d.ifactor = op->ifactor > 0.0 ? op->ifactor : 10.0;
    d.dfactor = op->dfactor > 0.0 ? op->dfactor : 0.2;
    d.max_num_steps = op->max_num_steps > 0 ? op->max_num_steps : 2147483647;
    // parameter tensors = segments of the flat gradient layout (one rms each in the mixed norm)
    {
        int n = 0;
        for (int l = 0; l < p->n_layers; ++l) {
            bool firstW = true, firstB = true;
            for (int q = 0; q < l; ++q) {
                if (p->layer_W[q] == p->layer_W[l]) firstW = false;
                if (p->layer_b[q] == p->layer_b[l]) firstB = false;
            }
            // a NULL destination = the tensor is not among the adjoint's parameters (requires_grad False, or left out of
            // adjoint_params): the reference's augmented state does not contain it (adjoint.py:176-189), so it has no say in
            // the mixed error norm either
            if (firstW && (!adj || g->grad_layer_W[l])) { d.seg_off[n] = y.gW_off[l]; d.seg_len[n] = p->layer_out[l] * p->layer_in[l]; ++n; }
            if (firstB && (!adj || g->grad_layer_b[l])) { d.seg_off[n] = y.gb_off[l]; d.seg_len[n] = p->layer_out[l]; ++n; }
        }
        if (!adj || g->grad_Wo) { d.seg_off[n] = y.gWo_off; d.seg_len[n] = y.rows * y.dlast; ++n; }
        if (!adj || g->grad_bo) { d.seg_off[n] = y.gbo_off; d.seg_len[n] = y.rows; ++n; }
        d.nseg = n;
    }
    // LDS plans
    const size_t lds_fwd = sizeof(float) * (size_t)(2 * y.HS + 2 * y.DS + y.Cp * 16);
    size_t lds_adj = sizeof(float) * (size_t)(4 * y.HS + ((y.L > 0 ? y.L : 1) + 4) * y.DS + 2 * y.Cp * 16 + 4 * 16 * 17);
    const size_t red_bytes = 4 * 256 * sizeof(double);
    d.gacc_in_lds = adj && lds_adj + sizeof(float) * (size_t)w.theta1 + red_bytes <= (size_t)kLdsLimit;
    if (d.gacc_in_lds) lds_adj += sizeof(float) * (size_t)w.theta1;
    const size_t lds = adj ? lds_adj : lds_fwd;

#define DP_TRY(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) { snprintf(err, errn, "%s failed: %s", #expr, hipGetErrorString(e_)); return NCDE_ERR_HIP; } \
    } while (0)

    DP_TRY(hipMemcpyAsync(base + w.off_t, ts->t, sizeof(double) * n_t, hipMemcpyHostToDevice, st));
    if (replay && replay_n > 0) {
        if (replay_n > kTraceCap) { snprintf(err, errn, "replay of %d attempts: at most %d", replay_n, kTraceCap); return NCDE_ERR_INVALID; }
        double* rdev = (double*)(base + w.off_tr) + 4 * (size_t)kTraceCap;
        DP_TRY(hipMemcpyAsync(rdev, replay, sizeof(double) * 2 * replay_n, hipMemcpyHostToDevice, st));
        DP_TRY(hipStreamSynchronize(st));
        d.replay = rdev;
        d.replay_n = replay_n;
    }
    std::vector<float> kf;
    if (ts->knots) {
        kf.resize(p->n_knots);
        for (int i = 0; i < p->n_knots; ++i) kf[i] = (float)ts->knots[i];
        DP_TRY(hipMemcpyAsync(base + w.off_knots, kf.data(), sizeof(float) * p->n_knots, hipMemcpyHostToDevice, st));
        DP_TRY(hipStreamSynchronize(st));      // kf is a local: the copy must have left it before we go on
    }
    const long long BH = (long long)p->batch * p->hidden;
    const int egrid = (int)((BH + 255) / 256);
    const int tgrid = (w.theta1 + 255) / 256;
    DP_TRY(ncde_lds_optin((const void*)ncde_dp_stage, lds));
    hipLaunchKernelGGL(ncde_dp_init, dim3(egrid), dim3(256), 0, st, d);
    DP_TRY(hipGetLastError());
    DpCtrl hc;
    const int rounds_per_sync = 16;
    // Fused attempt kernels (ncde_adaptive_fast.hip): ONE launch per attempt.  The host still only counts launches; how many it enqueues
    // before it looks at the controller again follows the progress so far (attempts per unit of solver time), so a solve is polled a
    // handful of times, not once per 16 attempts.
    d.fused = ncde_dpf_supported(p, adj) ? 1 : 0;
    if (d.fused) {
        const int rcp = ncde_dpf_prepare(p, &d, sizeof(d), adj, st);
        if (rcp != NCDE_OK) { snprintf(err, errn, "fused dopri5: weight image failed (%d)", rcp); return rcp; }
        const double t_lo = ts->t[0], t_hi = ts->t[n_t - 1];
        int rounds = 48;
        for (long long guard = 0; guard < (1LL << 40); ++guard) {
            const int rc = ncde_dpf_launch(p, &d, sizeof(d), adj, rounds, st);
            if (rc != NCDE_OK) { snprintf(err, errn, "fused dopri5 launch failed (%d)", rc); return rc; }
            DP_TRY(hipMemcpyAsync(&hc, d.ctrl, sizeof(DpCtrl), hipMemcpyDeviceToHost, st));
            DP_TRY(hipStreamSynchronize(st));
            if (hc.error != 0 || hc.phase == DP_DONE) break;
            const double done = adj ? (hc.t0 + t_hi) / (t_hi - t_lo) : (hc.t0 - t_lo) / (t_hi - t_lo);
            const double launched = (double)hc.n_attempts + 2.0;
            double remaining = done > 1e-3 ? launched * (1.0 - done) / done : 4.0 * launched;
            remaining = remaining * 1.05 + 6.0 + (adj ? 3.0 * (hc.interval - 1) : 0.0);
            rounds = (int)std::min(std::min(1024.0, 2.0 * rounds), std::max(8.0, remaining));      // (early estimates overshoot: first steps are short)
        }
    }
    for (long long guard = 0; !d.fused && guard < (1LL << 40); ++guard) {
        for (int r = 0; r < rounds_per_sync; ++r) {
            for (int slot = 1; slot <= 6; ++slot) {
                hipLaunchKernelGGL(ncde_dp_stage, dim3(w.n_wg), dim3(GEN_THREADS), lds, st, d, slot);
                if (adj) hipLaunchKernelGGL(ncde_dp_reduce_theta, dim3(tgrid), dim3(256), 0, st, d, slot);
            }
            hipLaunchKernelGGL(ncde_dp_control, dim3(1), dim3(256), 0, st, d);
            hipLaunchKernelGGL(ncde_dp_commit, dim3(egrid), dim3(256), 0, st, d);
        }
        DP_TRY(hipGetLastError());
        DP_TRY(hipMemcpyAsync(&hc, d.ctrl, sizeof(DpCtrl), hipMemcpyDeviceToHost, st));
        DP_TRY(hipStreamSynchronize(st));
        if (hc.error != 0 || hc.phase == DP_DONE) break;
    }
    if (d.trace_cap > 0) {
        const int rows = std::min(d.trace_cap, hc.n_attempts);
        if (rows > 0) DP_TRY(hipMemcpy(op->trace, d.TR, sizeof(double) * 4 * rows, hipMemcpyDeviceToHost));
    }
    if (stats) {
        stats->nfe = hc.nfe; stats->n_accepted = hc.n_accept; stats->n_rejected = hc.n_reject;
    }
    if (hc.error == 1) { snprintf(err, errn, "underflow in dt %g", hc.dt); return NCDE_ERR_INVALID; }
    if (hc.error == 2) { snprintf(err, errn, "non-finite values in state `y`"); return NCDE_ERR_INVALID; }
    if (hc.error == 3) { snprintf(err, errn, "max_num_steps exceeded (%d)", d.max_num_steps); return NCDE_ERR_INVALID; }
    if (hc.error == 4) { snprintf(err, errn, "the record holds %d accepted steps and the solve needs more: pass a larger record", d.tape_cap); return NCDE_ERR_WORKSPACE; }
    if (adj) {
        // dL/dz0 = a at the start time; dL/dtheta = the parameter part, scattered into the caller's buffers
        const float* a_end = (d.fused && hc.cur) ? d.AC : d.A0;        // fused attempt kernels: the state is double-buffered
        const float* g_end = (d.fused && hc.cur) ? d.GCT : d.G0T;
        DP_TRY(hipMemcpyAsync(g->grad_z0, a_end, sizeof(float) * BH, hipMemcpyDeviceToDevice, st));
        // one "partial" of theta_size floats: a pure scatter into the destinations that exist
        ReduceSegs segs{};
        int n = 0;
        for (int l = 0; l < p->n_layers; ++l) {
            bool firstW = true, firstB = true;
            for (int q = 0; q < l; ++q) {
                if (p->layer_W[q] == p->layer_W[l]) firstW = false;
                if (p->layer_b[q] == p->layer_b[l]) firstB = false;
            }
            if (firstW && g->grad_layer_W[l]) { segs.off[n] = y.gW_off[l]; segs.len[n] = p->layer_out[l] * p->layer_in[l]; segs.dst[n] = g->grad_layer_W[l]; ++n; }
            if (firstB && g->grad_layer_b[l]) { segs.off[n] = y.gb_off[l]; segs.len[n] = p->layer_out[l]; segs.dst[n] = g->grad_layer_b[l]; ++n; }
        }
        if (g->grad_Wo) { segs.off[n] = y.gWo_off; segs.len[n] = y.rows * y.dlast; segs.dst[n] = g->grad_Wo; ++n; }
        if (g->grad_bo) { segs.off[n] = y.gbo_off; segs.len[n] = y.rows; segs.dst[n] = g->grad_bo; ++n; }
        segs.n = n;
        if (n > 0) {
            hipLaunchKernelGGL(ncde_reduce_partials, dim3((y.theta_size + 255) / 256), dim3(256), 0, st, g_end, 1, y.theta_size, segs);
            DP_TRY(hipGetLastError());
        }
    }
    return NCDE_OK;
#undef DP_TRY
}

// Reverse sweep of a taped solve: dL/dz0 and dL/dtheta from the record of ncde_dp_solve(..., record).  Synchronises the stream
// once to read the record header back (step count / overflow: a record of an unfinished solve must be refused before any launch), and
// once more for the (tiny) knot upload when the control has a user knot grid; everything after that is stream-ordered.
int ncde_dp_tape_backward_run(const NcdeProblem* p, const NcdeTimeSpec* ts, const NcdeAdaptiveOptions* op, const void* record, size_t record_bytes,
                              const float* grad_out, const NcdeGrads* g, void* ws, size_t ws_bytes, hipStream_t st, char* err, size_t errn) {
    const Layout y = make_layout(p);
    const int n_t = ts->n_t;
    const TapeWs w = tape_ws(p, y);
    if (ws_bytes < w.total) { snprintf(err, errn, "workspace %zu B < %zu B", ws_bytes, w.total); return NCDE_ERR_WORKSPACE; }
    const int cap = tape_cap_of(p, n_t, record_bytes);
    if (cap < 1) { snprintf(err, errn, "record of %zu B holds no step", record_bytes); return NCDE_ERR_WORKSPACE; }
    const TapeLayout tl = tape_layout(p, n_t, cap);
    char* base = (char*)ws;
    char* rb = (char*)const_cast<void*>(record);
    DpArgs d;
    memset(&d, 0, sizeof(d));
    fill_kargs(p, y, &d.a);
    d.a.n_out = n_t;
    d.n_wg = y.n_wg; d.n_t = n_t; d.n_knots = p->n_knots; d.theta1 = y.theta_size + 1;
    d.tape = (DpTapeHeader*)rb;
    d.tape_steps = (DpStepRec*)(rb + tl.off_steps);
    d.tape_x = (float*)(rb + tl.off_x);
    d.tape_y = (float*)(rb + tl.off_y);
    d.tape_cap = cap;
    d.knots = ts->knots ? (const float*)(base + w.off_knots) : nullptr;
    d.KF = (float*)(base + w.off_kf); d.KBR = (float*)(base + w.off_kb);
    d.DZ0 = (float*)(base + w.off_dz0); d.F0B = (float*)(base + w.off_f0b); d.SCB = (float*)(base + w.off_scb);
    d.GP = (float*)(base + w.off_gp); d.PN2 = (double*)(base + w.off_pn2);
    d.grad_out = grad_out; d.gz0 = g->grad_z0;
    d.rtol = op->rtol; d.atol = op->atol;
#define DP_TRY(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) { snprintf(err, errn, "%s failed: %s", #expr, hipGetErrorString(e_)); return NCDE_ERR_HIP; } \
    } while (0)
    std::vector<float> kf;
    if (ts->knots) {
        kf.resize(p->n_knots);
        for (int i = 0; i < p->n_knots; ++i) kf[i] = (float)ts->knots[i];
        DP_TRY(hipMemcpyAsync(base + w.off_knots, kf.data(), sizeof(float) * p->n_knots, hipMemcpyHostToDevice, st));
        DP_TRY(hipStreamSynchronize(st));
    }
    DpTapeHeader hh;
    DP_TRY(hipMemcpyAsync(&hh, rb, sizeof(hh), hipMemcpyDeviceToHost, st));
    DP_TRY(hipStreamSynchronize(st));
    if (hh.magic != DP_TAPE_MAGIC || hh.n_t != n_t || hh.n_steps < 1 || hh.n_steps > cap || hh.overflow) {
        snprintf(err, errn, "not the record of a finished taped solve of this problem (magic %x, n_t %d, steps %d / %d)", hh.magic, hh.n_t, hh.n_steps, cap);
        return NCDE_ERR_INVALID;
    }
    size_t lds = sizeof(float) * (size_t)(7 * y.HS + ((y.L > 0 ? y.L : 1) + 4) * y.DS + 2 * y.Cp * 16 + 4 * 16 * 17);
    const size_t red_bytes = 4 * 256 * sizeof(double);
    d.gacc_in_lds = lds + sizeof(float) * (size_t)d.theta1 + red_bytes <= (size_t)kLdsLimit;
    if (d.gacc_in_lds) lds += sizeof(float) * (size_t)d.theta1;
    DP_TRY(ncde_lds_optin((const void*)ncde_dp_tape_backward, lds));
    DP_TRY(ncde_lds_optin((const void*)ncde_dp_tape_finish, lds));
    if (ncde_dpf_tape_supported(p)) {      // the sweep on the fused stage machinery (ncde_adaptive_fast.hip); same hand-over to the finish launches
        d.WP = (float*)(base + w.off_wp);
        const int rc = ncde_dpf_tape_launch(p, &d, sizeof(d), st);
        if (rc != NCDE_OK) { snprintf(err, errn, "fused taped sweep failed (%d)", rc); return rc; }
    } else {
        hipLaunchKernelGGL(ncde_dp_tape_backward, dim3(y.n_wg), dim3(GEN_THREADS), lds, st, d);
    }
    if (hh.delta_active) hipLaunchKernelGGL(ncde_dp_tape_finish, dim3(y.n_wg), dim3(GEN_THREADS), lds, st, d, 1);
    hipLaunchKernelGGL(ncde_dp_tape_finish, dim3(y.n_wg), dim3(GEN_THREADS), lds, st, d, 2);
    DP_TRY(hipGetLastError());
    // per-workgroup partials have stride theta_size + 1 (the time slot): compact sum via the common reduction on a strided view
    {
        ReduceSegs segs{};
        int n = 0;
        for (int l = 0; l < p->n_layers; ++l) {
            bool firstW = true, firstB = true;
            for (int q = 0; q < l; ++q) {
                if (p->layer_W[q] == p->layer_W[l]) firstW = false;
                if (p->layer_b[q] == p->layer_b[l]) firstB = false;
            }
            if (firstW) { segs.off[n] = y.gW_off[l]; segs.len[n] = p->layer_out[l] * p->layer_in[l]; segs.dst[n] = g->grad_layer_W[l]; ++n; }
            if (firstB) { segs.off[n] = y.gb_off[l]; segs.len[n] = p->layer_out[l]; segs.dst[n] = g->grad_layer_b[l]; ++n; }
        }
        segs.off[n] = y.gWo_off; segs.len[n] = y.rows * y.dlast; segs.dst[n] = g->grad_Wo; ++n;
        segs.off[n] = y.gbo_off; segs.len[n] = y.rows; segs.dst[n] = g->grad_bo; ++n;
        segs.n = n;
        for (int i = 0; i < n; ++i)
            if (!segs.dst[i]) { snprintf(err, errn, "NcdeGrads: NULL destination for a parameter gradient"); return NCDE_ERR_INVALID; }
        hipLaunchKernelGGL(ncde_reduce_partials, dim3((d.theta1 + 255) / 256), dim3(256), 0, st, d.GP, y.n_wg, d.theta1, segs);
        DP_TRY(hipGetLastError());
    }
    return NCDE_OK;
#undef DP_TRY
}
