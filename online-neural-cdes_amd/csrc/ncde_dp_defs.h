// Shared definitions of the adaptive dopri5 kernels: the per-launch kernels of ncde_adaptive.hip and the fused attempt kernels of
// ncde_adaptive_fast.hip -- controller state, argument block, tableau, stage descriptors, dense output, and the CONTROLLER itself
// (dp_control_body: norms over the whole batch, accept / reject, next step, phase transitions) as a device function, so that it
// runs either as its own one-workgroup launch (ncde_dp_control) or in the last workgroup of a fused attempt launch to arrive.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>

#include "ncde_common.h"

namespace {

enum { DP_INIT0 = 0, DP_INIT1 = 1, DP_STEP = 2, DP_DONE = 3, DP_FIN = 4 };      // DP_FIN: fused adjoint only (dense output at an interval end)
constexpr int DP_MAXSEG = 2 * NCDE_MAX_LAYERS + 8;

struct DpCtrl {
    double t0, dt, t1;      // solver time (forward: t; adjoint: s = -t): current time, attempted step, t0 + dt
    double t_goal;          // adjoint: end of the current output interval in solver time
    double h0d;
    int phase, accepted_now, finish_now;
    int j_out, j_begin, j_end;       // forward: next output row; rows emitted by the step just accepted
    int interval;                    // adjoint: solving t[interval] -> t[interval - 1]
    int row_now;                     // adjoint commit: output row the finished interval ends at
    int n_attempts, n_accept, n_reject, nfe, steps_this_solve, error;
    float h0, dtf, dtf_commit, x_end;
    StageDesc st[7];                 // st[0]: the f0 / probe evaluation; st[1..6]: stages 2..7 of the attempt
    // fused attempt kernels (ncde_adaptive_fast.hip)
    int cur;                         // which of the two state buffers holds (y0, k1) [forward] / (y0, a0) [adjoint]
    unsigned ticket;                 // workgroups of the current launch that have finished (the last one runs the controller)
    StageDesc st_k1;                 // adjoint: where stage 1 of the next attempt is (re-)evaluated -- st[0] after the initial-step
                                     // phases, st[6] of the attempt just accepted afterwards (FSAL, rk_common.py:76-80)
};

// Record of a taped forward solve (adjoint=False): the accepted steps, what the reverse sweep of ncde_dp_tape_* needs.
struct DpTapeHeader {
    int magic, cap, n_steps, delta_active;   // delta_active: dt of step 1 came from _select_initial_step and attempt 1 was accepted
    int h0_const, h1_const, n_t, overflow;
    float h0, h1, d0, d1, d2, pad_[3];
    double t_start;
    double pad2_[23];
};
static_assert(sizeof(DpTapeHeader) == 256, "tape header is 256 bytes");
struct DpStepRec {
    double t0, dt;
    int j_begin, j_end, attempt, pad_;
};
constexpr int DP_TAPE_MAGIC = 0x44503554;

struct DpArgs {
    KArgs a;
    DpCtrl* ctrl;
    // taped solve: record of the accepted steps (forward writes, the reverse sweep reads)
    DpTapeHeader* tape;
    DpStepRec* tape_steps;
    float* tape_x;           // [n_t] dense-output abscissae
    float* tape_y;           // [cap][B][H] state at the start of every accepted step
    int tape_cap;
    // reverse sweep of the taped solve
    float* KF;  float* KBR;  // [n_wg][7][HS] stage derivatives / their cotangents, per workgroup
    float* DZ0; float* F0B; float* SCB;   // [B][H]
    double* PN2;             // [n_wg][4]
    float* gz0;
    int adj, n_wg, n_t, n_knots, theta1;   // theta1 = theta_size + 1 (the vjp_t slot)
    const double* t_out;     // [n_t] device
    const float* knots;      // [n_knots] device or NULL
    float* XOUT;             // [n_t] dense-output abscissae of the accepted step
    float* Y0;  float* YC;  float* KY;      // [B][H], [B][H], [7][B][H]
    float* A0;  float* AC;  float* KA;      // adjoint: the a part
    float* GP;               // [n_wg][theta1] per-workgroup stage partials
    float* KT;               // [7][theta1] reduced parameter-part stage derivatives (incl. vjp_t)
    float* G0T; float* GCT;  // [theta1] current / candidate parameter part
    double* PN;              // [n_wg][4] per-workgroup partial sums of squares
    double* TR;              // [trace_cap][4] diagnostics: t0, dt, accepted, error ratio per attempt
    int trace_cap;
    const double* replay;    // [replay_n][2] forced (dt, accepted) per attempt (verification), device
    int replay_n;
    float* out;              // forward: [B][n_t][H]
    const float* z_out;      // adjoint
    const float* grad_out;
    double rtol, atol, min_step, max_step, first_step, safety, ifactor, dfactor;
    int max_num_steps;
    int nseg, seg_off[DP_MAXSEG], seg_len[DP_MAXSEG];
    int lds_words_fwd, lds_words_adj, gacc_in_lds;
    float* WP;               // fused attempt kernels: per-lane weight image (ncde_dpf_pack / ncde_dpa_pack)
    double* SEGP;            // fused adjoint: [reduce block][2][DP_MAXSEG + 1] per-segment sums of squares of the parameter part
    int n_rblk;              // number of reduce blocks
    int fused;               // 1: the fused attempt kernels drive the solve (state ping-pong by ctrl->cur, no commit launch)
};

__device__ const float kAlpha[6] = {(float)(1.0 / 5), (float)(3.0 / 10), (float)(4.0 / 5), (float)(8.0 / 9), 1.0f, 1.0f};
__device__ const float kBeta[6][6] = {
    {(float)(1.0 / 5), 0, 0, 0, 0, 0},
    {(float)(3.0 / 40), (float)(9.0 / 40), 0, 0, 0, 0},
    {(float)(44.0 / 45), (float)(-56.0 / 15), (float)(32.0 / 9), 0, 0, 0},
    {(float)(19372.0 / 6561), (float)(-25360.0 / 2187), (float)(64448.0 / 6561), (float)(-212.0 / 729), 0, 0},
    {(float)(9017.0 / 3168), (float)(-355.0 / 33), (float)(46732.0 / 5247), (float)(49.0 / 176), (float)(-5103.0 / 18656), 0},
    {(float)(35.0 / 384), 0.0f, (float)(500.0 / 1113), (float)(125.0 / 192), (float)(-2187.0 / 6784), (float)(11.0 / 84)}};
__device__ const float kCErr[7] = {(float)(35.0 / 384 - 1951.0 / 21600), 0.0f, (float)(500.0 / 1113 - 22642.0 / 50085),
                                   (float)(125.0 / 192 - 451.0 / 720), (float)(-2187.0 / 6784 - -12231.0 / 42400),
                                   (float)(11.0 / 84 - 649.0 / 6300), (float)(-1.0 / 60.0)};
__device__ const float kMid[7] = {(float)(6025192743.0 / 30085553152.0 / 2), 0.0f, (float)(51252292925.0 / 65400821598.0 / 2),
                                  (float)(-2691868925.0 / 45128329728.0 / 2), (float)(187940372067.0 / 1594534317056.0 / 2),
                                  (float)(-1776094331.0 / 19743644256.0 / 2), (float)(11237099.0 / 235043384.0 / 2)};

// bucketize(t, knots, right=False) - 1 clamped, fraction and knot spacing (interpolation_linear.py:212-219)
__device__ StageDesc dp_stage_desc(float t, const float* knots, int n_knots) {
    StageDesc d;
    if (!knots) {
        d.idx = piece_index(t, n_knots - 1);
        d.frac = t - (float)d.idx;
        d.kdt = 1.0f;
        return d;
    }
    int lo = 0, hi = n_knots;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (knots[mid] < t) lo = mid + 1;
        else hi = mid;
    }
    int idx = lo - 1;
    idx = idx < 0 ? 0 : (idx > n_knots - 2 ? n_knots - 2 : idx);
    d.idx = idx;
    d.frac = t - knots[idx];
    d.kdt = knots[idx + 1] - knots[idx];
    return d;
}

// stage descriptors of one attempt [t0, t0 + dt] (rk_common.py:61-73): stage times in fp32; the two alpha = 1 stages sit
// one ulp before t1 (misc.py:183-188).  neg: solver time is the negated real time (adjoint).
__device__ void dp_plan_attempt(DpCtrl* c, const DpArgs& d) {
    c->t1 = c->t0 + c->dt;
    const float t0f = (float)c->t0, dtf = (float)c->dt, t1f = (float)c->t1;
    c->dtf = dtf;
    for (int i = 0; i < 6; ++i) {
        float ti;
        if (kAlpha[i] == 1.0f) ti = nextafterf(t1f, -INFINITY);
        else ti = t0f + kAlpha[i] * dtf;
        c->st[i + 1] = dp_stage_desc(d.adj ? -ti : ti, d.knots, d.n_knots);
    }
}

__device__ __forceinline__ float dp_poly(float y0, float y1, float ym, float f0, float f1, float dt, float x) {
    // interp.py:4-61: fit on (y0, y1, y_mid, f0, f1), evaluate at x in [0, 1]
    const float a = 2.0f * dt * (f1 - f0) - 8.0f * (y1 + y0) + 16.0f * ym;
    const float b = dt * (5.0f * f0 - 3.0f * f1) + 18.0f * y0 + 14.0f * y1 - 32.0f * ym;
    const float c = dt * (f1 - 4.0f * f0) - 11.0f * y0 - 5.0f * y1 + 16.0f * ym;
    const float dd = dt * f0;
    float total = y0 + x * dd;
    float xp = x;
    xp = xp * x; total = total + xp * c;
    xp = xp * x; total = total + xp * b;
    xp = xp * x; total = total + xp * a;
    return total;
}


__device__ __forceinline__ double dp_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// block-wide sum / max of a double (256 threads = 4 waves); every thread gets the total.  Fixed order: butterfly inside a wave, then
// (w0 + w1) + (w2 + w3).
__device__ double dp_block_sum(double v, double* sh) {
    const int tid = threadIdx.x;
    const double w = dp_wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) sh[tid >> 6] = w;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__device__ double dp_block_max(double v, double* sh) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    __syncthreads();
    if ((tid & 63) == 0) sh[tid >> 6] = v;
    __syncthreads();
    return fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}

// mixed norm of the parameter part (adjoint.py:239-242): max over parameter tensors of rms(v / scale), and |vjp_t| / scale.
// which: 0 = state / scale(state), 1 = f0 / scale, 2 = (f1 - f0) / scale, 3 = err / tol (also writes the candidate GCT)
__device__ double dp_theta_norm(const DpArgs& d, int which, float dtf, double* sh) {
    const int tid = threadIdx.x;
    const float rtolf = (float)d.rtol, atolf = (float)d.atol;
    double best = 0.0;
    for (int sg = 0; sg <= d.nseg; ++sg) {      // sg == nseg: the vjp_t slot (|.|, not rms)
        const int off = sg < d.nseg ? d.seg_off[sg] : d.theta1 - 1;
        const int len = sg < d.nseg ? d.seg_len[sg] : 1;
        double acc = 0.0;
        for (int e = tid; e < len; e += 256) {
            const int k = off + e;
            const float g0 = d.G0T[k];
            float q;
            if (which == 3) {
                float inc = 0.0f, err = 0.0f;
                for (int j = 0; j < 6; ++j) inc += d.KT[(long long)j * d.theta1 + k] * (kBeta[5][j] * dtf);
                for (int j = 0; j < 7; ++j) err += d.KT[(long long)j * d.theta1 + k] * (dtf * kCErr[j]);
                const float g1 = g0 + inc;
                d.GCT[k] = g1;
                q = err / (atolf + rtolf * fmaxf(fabsf(g0), fabsf(g1)));
            } else {
                const float scale = atolf + fabsf(g0) * rtolf;
                const float k0 = d.KT[k];
                q = which == 0 ? g0 / scale : (which == 1 ? k0 / scale : (d.KT[(long long)d.theta1 + k] - k0) / scale);
            }
            acc += (double)q * q;
        }
        const double tot = dp_block_sum(acc, sh);
        const double nrm = sg < d.nseg ? sqrt(tot / (double)len) : sqrt(tot);
        best = fmax(best, nrm);
    }
    return best;
}


// the same from the per-block, per-segment sums ncde_dpf_reduce left (slot 0 / 1: the first / second norm of the phase)
// ... in two halves, so that the controller can have these loads in flight together with those of the y / a partial sums: the per-thread
// sums over the reduce blocks (dp_theta_norm_load, up to eight segments), then the exchange (dp_theta_norm_combine)
__device__ __forceinline__ void dp_theta_norm_load(const DpArgs& d, int slot, int s0, double (&acc)[8]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.0;
    for (int b = tid; b < d.n_rblk; b += 256) {
        const double* row = d.SEGP + ((long long)b * 2 + slot) * (DP_MAXSEG + 1) + s0;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (s0 + i <= d.nseg) acc[i] += __hip_atomic_load(&row[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ double dp_theta_norm_combine(const DpArgs& d, int s0, const double (&acc)[8], double* sh) {      // (256 threads; sh: >= 32 doubles)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double best = 0.0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double w = dp_wave_sum(acc[i]);
        if (lane == 0) sh[wave * 8 + i] = w;
    }
    __syncthreads();
    for (int i = 0; i < 8 && s0 + i <= d.nseg; ++i) {
        const int sg = s0 + i;
        const double tot = (sh[i] + sh[8 + i]) + (sh[16 + i] + sh[24 + i]);
        best = fmax(best, sg < d.nseg ? sqrt(tot / (double)d.seg_len[sg]) : sqrt(tot));
    }
    __syncthreads();
    return best;
}
__device__ double dp_theta_norm_fused(const DpArgs& d, int slot, double* sh) {
    double best = 0.0;
    for (int s0 = 0; s0 <= d.nseg; s0 += 8) {      // eight segments per pass
        double acc[8];
        dp_theta_norm_load(d, slot, s0, acc);
        best = fmax(best, dp_theta_norm_combine(d, s0, acc, sh));
    }
    return best;
}

__device__ void dp_control_body(const DpArgs& d, double* sh, int* sh_flags, float* sh_xp) {
    int& sh_accept = sh_flags[0];
    int& sh_finish = sh_flags[1];
    float& sh_x = *sh_xp;
    DpCtrl* c = d.ctrl;
    const int tid = threadIdx.x;
    const int phase = c->phase;
    if (phase == DP_DONE || c->error != 0) {
        if (tid == 0) c->accepted_now = 0;
        return;
    }
    if (phase == DP_FIN) {      // fused adjoint: the candidate buffers now hold (y, a, g_theta) at the interval end
        if (tid == 0) {
            c->cur ^= 1;
            c->accepted_now = 0;
            if (c->interval == 1) {
                c->phase = DP_DONE;
            } else {       // next output interval: a fresh solve (new f0, new initial step), adjoint.py:116-130
                c->interval -= 1;
                c->t0 = -d.t_out[c->interval];
                c->t_goal = -d.t_out[c->interval - 1];
                c->t1 = c->t0;
                c->steps_this_solve = 0;
                c->st[0] = dp_stage_desc(-(float)c->t0, d.knots, d.n_knots);
                c->phase = DP_INIT0;
            }
        }
        return;
    }
    // batch-wide sums of squares from the stage kernels
    double p[4] = {0, 0, 0, 0};
    for (int w = tid; w < d.n_wg; w += 256)      // (agent-scope loads: in a fused launch other workgroups of THIS launch wrote them)
        for (int q = 0; q < 4; ++q) p[q] += __hip_atomic_load(&d.PN[(long long)w * 4 + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // fused adjoint: the parameter part's per-block sums, requested now so that both sets of write-through data travel together
    const bool seg_pre = d.fused && d.adj && d.nseg < 8;
    double seg0[8], seg1[8];
    if (seg_pre) {
        dp_theta_norm_load(d, 0, 0, seg0);
        if (phase == DP_INIT0) dp_theta_norm_load(d, 1, 0, seg1);
    }
    double tot[4];
    for (int q = 0; q < 4; ++q) tot[q] = dp_block_sum(p[q], sh);
    const double nel = (double)d.a.B * (double)d.a.H;
    const float dtf = c->dtf;

    if (phase == DP_INIT0) {
        double d0 = sqrt(tot[0] / nel), d1 = sqrt(tot[d.adj ? 2 : 1] / nel);
        if (d.adj) {
            d0 = fmax(d0, sqrt(tot[1] / nel));
            d1 = fmax(d1, sqrt(tot[3] / nel));
            d0 = fmax(d0, seg_pre ? dp_theta_norm_combine(d, 0, seg0, sh) : (d.fused ? dp_theta_norm_fused(d, 0, sh) : dp_theta_norm(d, 0, dtf, sh)));
            d1 = fmax(d1, seg_pre ? dp_theta_norm_combine(d, 0, seg1, sh) : (d.fused ? dp_theta_norm_fused(d, 1, sh) : dp_theta_norm(d, 1, dtf, sh)));
        }
        if (tid == 0) {
            const float d0f = (float)d0, d1f = (float)d1;
            c->nfe += 1;
            c->st_k1 = c->st[0];      // fused adjoint: stage 1 of the first attempt is re-evaluated where f0 was
            c->h0d = d1;      // keep d1 for the second half of the rule
            float h0;
            if (d0f < 1e-5f || d1f < 1e-5f) h0 = 1e-6f;
            else h0 = 0.01f * d0f / d1f;
            c->h0 = h0;
            if (d.tape) {
                d.tape->h0 = h0; d.tape->d0 = d0f; d.tape->d1 = d1f;
                d.tape->h0_const = (d0f < 1e-5f || d1f < 1e-5f) ? 1 : 0;
                d.tape->t_start = c->t0;
            }
            if (d.first_step > 0.0) {      // options['first_step']: no probe evaluation (rk_common.py:160-164)
                c->dt = d.first_step;
                if (c->n_attempts < d.replay_n) c->dt = d.replay[2 * c->n_attempts];
                dp_plan_attempt(c, d);
                c->phase = DP_STEP;
            } else {
                const float tp = (float)c->t0 + h0;
                c->st[0] = dp_stage_desc(d.adj ? -tp : tp, d.knots, d.n_knots);
                c->phase = DP_INIT1;
            }
            c->accepted_now = 0;
        }
        return;
    }
    if (phase == DP_INIT1) {
        double s2 = sqrt(tot[0] / nel);
        if (d.adj) {
            s2 = fmax(s2, sqrt(tot[1] / nel));
            s2 = fmax(s2, seg_pre ? dp_theta_norm_combine(d, 0, seg0, sh) : (d.fused ? dp_theta_norm_fused(d, 0, sh) : dp_theta_norm(d, 2, dtf, sh)));
        }
        if (tid == 0) {
            const float h0 = c->h0, d1f = (float)c->h0d;
            const float d2f = (float)s2 / h0;
            float h1;
            if (d1f <= 1e-15f && d2f <= 1e-15f) h1 = fmaxf(1e-6f, h0 * 1e-3f);
            else h1 = powf(0.01f / fmaxf(d1f, d2f), 1.0f / 5.0f);
            c->nfe += 1;
            c->dt = (double)fminf(100.0f * h0, h1);
            if (c->n_attempts < d.replay_n) c->dt = d.replay[2 * c->n_attempts];
            if (d.tape) {
                d.tape->h1 = h1; d.tape->d2 = d2f;
                d.tape->h1_const = (d1f <= 1e-15f && d2f <= 1e-15f) ? 1 : 0;
            }
            dp_plan_attempt(c, d);
            c->phase = DP_STEP;
            c->accepted_now = 0;
        }
        return;
    }
    // ---- DP_STEP: error ratio of the attempt ---------------------------------------------------------------------
    double ratio = sqrt(tot[0] / nel);
    if (d.adj) {
        ratio = fmax(ratio, sqrt(tot[1] / nel));
        ratio = fmax(ratio, seg_pre ? dp_theta_norm_combine(d, 0, seg0, sh) : (d.fused ? dp_theta_norm_fused(d, 0, sh) : dp_theta_norm(d, 3, dtf, sh)));
    }
    if (tid == 0) {
        const float ratiof = (float)ratio;
        const double dt = c->dt;
        bool accept = ratiof <= 1.0f;
        if (dt > d.max_step) accept = false;
        if (dt <= d.min_step) accept = true;
        if (c->n_attempts < d.replay_n) accept = d.replay[2 * c->n_attempts + 1] != 0.0;           // replay of a recorded sequence
        if (!(ratio == ratio) || !(fabs(ratio) <= 1.79e308)) { c->error = 2; accept = false; }      // non-finite state
        c->nfe += 6;
        if (c->n_attempts < d.trace_cap) {
            double* tr = d.TR + 4LL * c->n_attempts;
            tr[0] = c->t0; tr[1] = dt; tr[2] = accept ? 1.0 : 0.0; tr[3] = (double)ratiof;
        }
        c->n_attempts += 1;
        c->steps_this_solve += 1;
        if (c->steps_this_solve > d.max_num_steps) c->error = 3;
        // next step size (misc.py:84-97)
        double dt_next;
        if (ratiof == 0.0f) dt_next = dt * d.ifactor;
        else {
            const double dfac = ratiof < 1.0f ? 1.0 : d.dfactor;
            const double factor = fmin(d.ifactor, fmax(d.safety / pow((double)ratiof, 0.2), dfac));
            dt_next = dt * factor;
        }
        dt_next = fmin(fmax(dt_next, d.min_step), d.max_step);
        if (c->n_attempts < d.replay_n) dt_next = d.replay[2 * c->n_attempts];      // (n_attempts already counts this attempt)
        int finish = 0;
        c->dtf_commit = c->dtf;
        if (accept) {
            c->n_accept += 1;
            const double t0 = c->t0, t1 = c->t1;
            if (!d.adj) {
                int j = c->j_out;
                c->j_begin = j;
                while (j < d.n_t && !(d.t_out[j] > t1)) {
                    d.XOUT[j] = (float)((d.t_out[j] - t0) / (t1 - t0));
                    if (d.tape) d.tape_x[j] = d.XOUT[j];
                    ++j;
                }
                c->j_end = c->j_out = j;
                if (j > c->j_begin) c->steps_this_solve = 0;      // the reference counts attempts per _advance(next_t) (rk_common.py:196-203)
                if (j >= d.n_t) finish = 1;
                if (d.tape) {          // tape of the accepted steps
                    const int m = c->n_accept - 1;
                    if (m < d.tape_cap) {
                        DpStepRec* r = d.tape_steps + m;
                        r->t0 = t0; r->dt = dt; r->j_begin = c->j_begin; r->j_end = c->j_end; r->attempt = c->n_attempts - 1;
                        d.tape->n_steps = m + 1;
                        if (m == 0) d.tape->delta_active = (c->n_attempts == 1 && !(d.first_step > 0.0)) ? 1 : 0;
                    } else {
                        d.tape->overflow = 1;
                        c->error = 4;
                    }
                }
            } else if (!(c->t_goal > t1)) {
                finish = 1;
                sh_x = (float)((c->t_goal - t0) / (t1 - t0));
                c->x_end = sh_x;
                c->row_now = c->interval - 1;
            }
            c->t0 = t1;
        } else {
            c->n_reject += 1;
        }
        c->dt = dt_next;
        if (!(c->t0 + c->dt > c->t0) && !finish && c->error == 0) c->error = 1;      // 'underflow in dt' (rk_common.py:232)
        c->accepted_now = accept ? 1 : 0;
        c->finish_now = finish;
        if (d.fused && accept && !(d.adj && finish)) c->cur ^= 1;      // fused attempt kernels: the candidate buffers become the state
        if (d.fused && d.adj && accept && !finish) c->st_k1 = c->st[6];   // ... and stage 1 of the next attempt is this attempt's stage 7 (FSAL)
        sh_accept = accept ? 1 : 0;
        sh_finish = finish;
        if (finish && d.fused && d.adj) {
            c->phase = DP_FIN;      // one more launch: the dense output at the interval end (same stages, other weights); then DP_FIN below
        } else if (finish) {
            if (!d.adj || c->interval == 1) {
                c->phase = DP_DONE;
            } else {       // next output interval: a fresh solve (new f0, new initial step), adjoint.py:116-130
                c->interval -= 1;
                c->t0 = -d.t_out[c->interval];
                c->t_goal = -d.t_out[c->interval - 1];
                c->t1 = c->t0;
                c->steps_this_solve = 0;
                c->st[0] = dp_stage_desc(-(float)c->t0, d.knots, d.n_knots);
                c->phase = DP_INIT0;
            }
        } else {
            dp_plan_attempt(c, d);
        }
    }
    __syncthreads();
    if (d.adj && sh_accept && !d.fused) {      // roll the parameter part (every thread): candidate, or the dense output at the interval end
        const float x = sh_x, dtc = c->dtf_commit;
        for (int k = tid; k < d.theta1; k += 256) {
            float g1 = d.GCT[k];
            if (sh_finish) {
                const float g0 = d.G0T[k];
                float gm = 0.0f;
                for (int j = 0; j < 7; ++j) gm += d.KT[(long long)j * d.theta1 + k] * (dtc * kMid[j]);
                gm = g0 + gm;
                g1 = dp_poly(g0, g1, gm, d.KT[k], d.KT[6LL * d.theta1 + k], dtc, x);
            }
            d.G0T[k] = g1;
            d.KT[k] = d.KT[6LL * d.theta1 + k];      // FSAL (unused after a finished interval: INIT0 recomputes plane 0)
        }
    }
}

}  // namespace
