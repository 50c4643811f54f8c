// Generic (runtime-dimension) Neural-CDE kernels for gfx950.
//
// One workgroup = one tile of 16 samples, 4 waves.  All activations live in LDS as [unit][sample]
// (sample fastest), which is simultaneously the B-operand image (k = unit, n = sample) and the
// A-operand image (m = unit, k = sample) of v_mfma_f32_16x16x4_f32, so every GEMM of the forward
// stage, of the VJP and of the weight gradient is an MFMA chain fed from LDS (activations) and
// global/L2 (weights).  The whole time loop runs inside the kernel: the state never leaves the CU.
//
// This family covers ANY (H, layer widths, C, T, B); shape-specialised register-resident kernels
// live in ncde_fast.hip.  Reference semantics restated (relative to /root/reference):
//   stage loop          modules/torchdiffeq/torchdiffeq/_impl/solvers.py:103-117, fixed_grid.py:6-29,
//                       rk_common.py:106-114
//   f_theta(z).dX/dt    src/ncde/vector_fields/base.py:83-104, modules/torchcde/torchcde/solver.py:112-137
//   dX/dt               modules/torchcde/torchcde/interpolation_linear.py:212-234, interpolation_cubic.py:315-336
//   adjoint sweep       modules/torchdiffeq/torchdiffeq/_impl/adjoint.py:37-145, misc.py:152-159
#include "ncde_common.h"

#include "ncde_generic_stage.h"

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(GEN_THREADS) void ncde_fwd_generic(KArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * NCDE_TILE;
    const int H = a.H, C = a.C, Hp = ru16(H), Cp = ru4(C);
    int Dp = Hp;
    for (int l = 0; l < a.n_layers; ++l) Dp = max(Dp, ru16(a.dout[l]));
    const int HS = Hp * 16, DS = Dp * 16;
    float* YS = lds;
    float* ACT0 = YS + HS;
    float* ACT1 = ACT0 + DS;
    float* Y0 = ACT1 + DS;
    float* K1 = Y0 + HS;
    float* K2 = K1 + HS;
    float* KO = K2 + HS;
    float* DX = KO + HS;
    const int total = 5 * HS + 2 * DS + Cp * 16;
    for (int e = tid; e < total; e += GEN_THREADS) lds[e] = 0.0f;
    __syncthreads();
    for (int e = tid; e < HS; e += GEN_THREADS) {
        const int h = e >> 4, s = e & 15, b = b0 + s;
        if (h < H && b < a.B) {
            const float v = a.z0[(long long)b * H + h];
            Y0[e] = v;
            YS[e] = v;
            a.out[((long long)b * a.n_out) * H + h] = v;
        }
    }
    const int S = n_stages(a.method);
    int cur_idx = -1;
    const bool planned = a.plan != nullptr;
    if (planned && !plan_header_ok(a, S)) return;      // (uniform: before the first barrier)
    const int n_steps = planned ? a.n_steps_fwd : a.T - 1;
    const int* pout = planned ? a.plan + plan_off_out(S, a.n_steps_fwd) : nullptr;
    for (int n = 0; n < n_steps; ++n) {
        const int* pstep = planned ? a.plan + plan_off_fwd() + n * plan_step_words(S) : nullptr;
        const float dt = planned ? __int_as_float(pstep[0]) : 1.0f;
        for (int j = 0; j < S; ++j) {
            const StageDesc sd = planned ? plan_stage(pstep, j) : default_stage(a.method, (float)n + stage_offset(a.method, j), a.n_pieces);
            if (a.interp != NCDE_INTERP_LINEAR || sd.idx != cur_idx) {
                load_dx(a, b0, sd, DX, Cp, tid);
                cur_idx = sd.idx;
            }
            __syncthreads();
            if (a.stages) {  // record the stage input for the exact discrete backward
                float* rec = a.stages + ((long long)(n * S + j) * a.B + b0) * H;
                for (int e = tid; e < 16 * H; e += GEN_THREADS) {
                    const int s = e / H, h = e - s * H;
                    if (b0 + s < a.B) rec[e] = YS[h * 16 + s];
                }
            }
            gen_stage_forward(a, YS, ACT0, ACT1, DX, KO, Hp, Cp, tid);
            for (int e = tid; e < HS; e += GEN_THREADS) {
                float y0 = Y0[e], k1 = K1[e], k2 = K2[e];
                const float yprev = y0;
                bool last;
                const float ys = StageCombine::apply(a.method, j, KO[e], dt, y0, k1, k2, last);
                YS[e] = ys;
                K1[e] = k1;
                K2[e] = k2;
                if (last) {
                    Y0[e] = y0;
                    const int h = e >> 4, s = e & 15, b = b0 + s;
                    if (h < H && b < a.B) {
                        if (planned) {   // outputs this step brackets: y1 itself at a grid hit, else linear interpolation (solvers.py:166-172)
                            const int q0 = pstep[1], q1 = q0 + pstep[2];
                            for (int q = q0; q < q1; ++q) {
                                const int kind = pout[2 * q];
                                const float slope = __int_as_float(pout[2 * q + 1]);
                                const float v = kind == 1 ? y0 : (kind == 0 ? yprev : yprev + slope * (y0 - yprev));
                                a.out[((long long)b * a.n_out + q) * H + h] = v;
                            }
                        } else if (a.output == NCDE_OUT_KNOTS) a.out[((long long)b * a.n_out + (n + 1)) * H + h] = y0;
                        else if (n == a.T - 2) a.out[((long long)b * a.n_out + 1) * H + h] = y0;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// adjoint (reverse sweep of y, a, g_theta)
// ------------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(GEN_THREADS) void ncde_adj_generic(KArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * NCDE_TILE;
    const int H = a.H, C = a.C, Hp = ru16(H), Cp = ru4(C), L = a.n_layers;
    int Dp = Hp;
    for (int l = 0; l < L; ++l) Dp = max(Dp, ru16(a.dout[l]));
    const int HS = Hp * 16, DS = Dp * 16;
    // state arrays [Hp][16]
    float* YS = lds;           // y stage input (= x_0)
    float* AS = YS + HS;       // a stage input
    float* Y0 = AS + HS;
    float* A0 = Y0 + HS;
    float* KY1 = A0 + HS;
    float* KY2 = KY1 + HS;
    float* KA1 = KY2 + HS;
    float* KA2 = KA1 + HS;
    float* KOY = KA2 + HS;     // f(y).dX of the stage
    float* KOA = KOY + HS;     // a^T df/dy of the stage
    float* X = KOA + HS;       // x_1..x_L, [L][Dp][16]
    float* G0 = X + (L > 0 ? L : 1) * DS;  // dL/dx ping
    float* G1 = G0 + DS;       // dL/dpre pong
    float* PW2 = G1 + DS;      // partial buffers of waves 2,3 (waves 0,1 use G0,G1)
    float* DX = PW2 + 2 * DS;
    float* SC = DX + Cp * 16;  // per-wave 16x17 transpose scratch
    float* GL = SC + GEN_NW * 16 * 17;  // parameter-gradient partial when it fits in LDS
    const int total = 10 * HS + ((L > 0 ? L : 1) + 4) * DS + Cp * 16 + GEN_NW * 16 * 17 + (a.gacc_in_lds ? a.theta_size : 0);
    for (int e = tid; e < total; e += GEN_THREADS) lds[e] = 0.0f;
    float* gacc = a.gacc_in_lds ? GL : a.gpart + (long long)blockIdx.x * a.theta_size;
    if (!a.gacc_in_lds)
        for (int e = tid; e < a.theta_size; e += GEN_THREADS) gacc[e] = 0.0f;
    __syncthreads();
    const int last_row = a.n_out - 1;
    const int S = n_stages(a.method);
    const bool disc = a.discrete != 0;
    const bool planned = a.plan != nullptr;
    if (planned && !plan_header_ok(a, S)) return;      // (uniform: before the first barrier)
    const int pw_ = plan_step_words(S);
    const int* pfwd = planned ? a.plan + plan_off_fwd() : nullptr;
    const int* pout = planned ? a.plan + plan_off_out(S, a.n_steps_fwd) : nullptr;
    const int* padj = planned ? a.plan + plan_off_adj(S, a.n_steps_fwd, a.n_out) : nullptr;
    const int n_rsteps = planned ? (disc ? a.n_steps_fwd : a.n_steps_adj) : a.T - 1;
    for (int e = tid; e < HS; e += GEN_THREADS) {
        const int h = e >> 4, s = e & 15, b = b0 + s;
        if (h < H && b < a.B) {
            const long long o = ((long long)b * a.n_out + last_row) * H + h;
            float g = a.grad_out[o];
            if (planned && disc) g = plan_out_cotangent(a, pfwd + (a.n_steps_fwd - 1) * pw_, pout, 1, (long long)b * a.n_out, h);
            A0[e] = g;
            if (disc) {   // cotangent of the last stage's k: rk4 (a dt)/8, midpoint / euler dt a
                const float dtl = planned ? __int_as_float(pfwd[(a.n_steps_fwd - 1) * pw_]) : 1.0f;
                AS[e] = a.method == NCDE_RK4_38 ? (g * dtl) * 0.125f : dtl * g;
            } else {
                const float y = a.z_out[o];
                Y0[e] = y; YS[e] = y; AS[e] = g;
            }
        }
    }
    for (int rstep = 0; rstep < n_rsteps; ++rstep) {
        // default grid: reverse step knot n -> n-1 (negated time s: -n -> -(n-1)); m = the forward step being transposed
        const int n = a.T - 1 - rstep, m = n_rsteps - 1 - rstep;
        const int* pstep = planned ? (disc ? pfwd + m * pw_ : padj + rstep * pw_) : nullptr;
        const float dt = planned ? __int_as_float(pstep[0]) : 1.0f;
        for (int j = 0; j < S; ++j) {
            StageDesc sd;
            if (planned) {
                sd = plan_stage(pstep, disc ? S - 1 - j : j);
            } else {
                const float s0 = -(float)n;
                // discrete mode walks the stages of forward step n-1 -> n backwards, at their forward times
                sd = default_stage(a.method, disc ? (float)(n - 1) + stage_offset(a.method, S - 1 - j) : -(s0 + stage_offset(a.method, j)), a.n_pieces);
            }
            const float w = disc ? 1.0f : stage_weight(a.method, j) * dt;
            load_dx(a, b0, sd, DX, Cp, tid);
            if (disc) {
                const float* rec = a.stages + ((long long)(m * S + (S - 1 - j)) * a.B + b0) * H;
                for (int e = tid; e < 16 * H; e += GEN_THREADS) {
                    const int s = e / H, h = e - s * H;
                    YS[h * 16 + s] = b0 + s < a.B ? rec[e] : 0.0f;
                }
            }
            __syncthreads();
            gen_stage_vjp(a, YS, AS, DX, nullptr, X, G0, G1, PW2, SC, KOY, KOA, gacc, w, Hp, Cp, DS, tid);
            if (disc) {
                // ---- transpose of the Butcher step: KOA = dL/dY_stage; next cotangent / new a ----------------
                // RK4 (3/8): c4 = a dt/8; c3 = 3 c4 + dt d4; c2 = 3 c4 - dt d4 + dt d3; c1 = c4 + dt d4 - dt/3 d3 + dt/3 d2;
                // a += d4+d3+d2+d1
                for (int e = tid; e < HS; e += GEN_THREADS) {
                    const float d = KOA[e];
                    float a0 = A0[e];
                    bool last = false;
                    float next = 0.0f;
                    if (a.method == NCDE_RK4_38) {
                        const float c4 = (a0 * dt) * 0.125f;
                        const float dt3 = dt * 0.333333343267440796f;
                        if (j == 0) { KA1[e] = d; next = 3.0f * c4 + dt * d; }
                        else if (j == 1) { KA2[e] = d; next = (3.0f * c4 - dt * KA1[e]) + dt * d; }
                        else if (j == 2) { KY1[e] = d; next = ((c4 + dt * KA1[e]) - dt3 * KA2[e]) + dt3 * d; }
                        else { a0 = (((a0 + KA1[e]) + KA2[e]) + KY1[e]) + d; last = true; }
                    } else if (a.method == NCDE_MIDPOINT) {
                        if (j == 0) { KA1[e] = d; next = (0.5f * dt) * d; }
                        else { a0 = (a0 + KA1[e]) + d; last = true; }
                    } else {
                        a0 = a0 + d; last = true;
                    }
                    if (last) {
                        const int h = e >> 4, s = e & 15, b = b0 + s;
                        const bool valid = h < H && b < a.B;
                        float dtp = 1.0f;   // dt of the forward step transposed next
                        if (planned) {
                            if (valid) {
                                const long long brow = (long long)b * a.n_out;
                                a0 = a0 + plan_out_cotangent(a, pstep, pout, 0, brow, h);
                                if (m > 0) a0 = a0 + plan_out_cotangent(a, pstep - pw_, pout, 1, brow, h);
                                else a0 = a0 + a.grad_out[brow * H + h];      // row 0 of the solution is z0 itself
                            }
                            if (m > 0) dtp = __int_as_float(pstep[-pw_]);
                        } else if (a.output == NCDE_OUT_KNOTS || n == 1) {
                            a0 = a0 + (valid ? a.grad_out[((long long)b * a.n_out + (a.output == NCDE_OUT_KNOTS ? n - 1 : 0)) * H + h] : 0.0f);
                        }
                        A0[e] = a0;
                        next = a.method == NCDE_RK4_38 ? (a0 * dtp) * 0.125f : dtp * a0;
                        if (m == 0 && valid) a.grad_z0[(long long)b * H + h] = a0;
                    }
                    AS[e] = next;
                }
                __syncthreads();
                continue;
            }
            // ---- Butcher bookkeeping in negated time: dy/ds = -f, da/ds = +a^T df/dy ----------------
            for (int e = tid; e < HS; e += GEN_THREADS) {
                float y0 = Y0[e], k1 = KY1[e], k2 = KY2[e];
                bool last;
                const float ys = StageCombine::apply(a.method, j, -KOY[e], dt, y0, k1, k2, last);
                YS[e] = ys; KY1[e] = k1; KY2[e] = k2;
                float a0 = A0[e], q1 = KA1[e], q2 = KA2[e];
                const float as = StageCombine::apply(a.method, j, KOA[e], dt, a0, q1, q2, last);
                KA1[e] = q1; KA2[e] = q2;
                if (!last) {
                    AS[e] = as;
                } else {
                    const int h = e >> 4, s = e & 15, b = b0 + s;
                    const bool valid = h < H && b < a.B;
                    if (planned) {   // end of one reverse solve t[i] -> t[i-1]: reset y to the stored z(t[i-1]), a += dL/dz(t[i-1])
                        const int row = pstep[1];
                        if (row >= 0) {
                            const long long o = ((long long)b * a.n_out + row) * H + h;
                            y0 = valid ? a.z_out[o] : 0.0f;
                            a0 = a0 + (valid ? a.grad_out[o] : 0.0f);
                        }
                    } else if (a.output == NCDE_OUT_KNOTS) {  // reset y to the stored value, add dL/dz at this knot
                        const long long o = ((long long)b * a.n_out + (n - 1)) * H + h;
                        y0 = valid ? a.z_out[o] : 0.0f;
                        a0 = a0 + (valid ? a.grad_out[o] : 0.0f);
                    } else if (n == 1) {
                        a0 = a0 + (valid ? a.grad_out[((long long)b * a.n_out) * H + h] : 0.0f);
                    }
                    Y0[e] = y0; YS[e] = y0; A0[e] = a0; AS[e] = a0;
                    if (rstep == n_rsteps - 1 && valid) a.grad_z0[(long long)b * H + h] = a0;
                }
            }
            __syncthreads();
        }
    }
    if (a.gacc_in_lds) {
        float* dst = a.gpart + (long long)blockIdx.x * a.theta_size;
        for (int e = tid; e < a.theta_size; e += GEN_THREADS) dst[e] = GL[e];
    }
}

// ------------------------------------------------------------------------------------------------
// K4: deterministic reduction of per-workgroup partials + scatter into the caller's gradient buffers
// ------------------------------------------------------------------------------------------------
struct ReduceSegs {
    int n;
    int off[2 * NCDE_MAX_LAYERS + 6];
    int len[2 * NCDE_MAX_LAYERS + 6];
    float* dst[2 * NCDE_MAX_LAYERS + 6];
};

extern "C" __global__ __launch_bounds__(256) void ncde_reduce_partials(const float* __restrict__ gpart, int n_part,
                                                                        int theta_size, ReduceSegs segs) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= theta_size) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int p = 0;
    for (; p + 3 < n_part; p += 4) {
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
