// Shared device-side definitions for the Neural-CDE kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ncde_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NCDE_TILE 16  // samples per workgroup tile = N of v_mfma_f32_16x16x4_f32

// Kernel argument block (passed by value).
struct KArgs {
    int B, T, C, H, interp, method, output, n_layers, n_pieces, n_out;
    int din[NCDE_MAX_LAYERS], dout[NCDE_MAX_LAYERS];
    const float* W[NCDE_MAX_LAYERS];
    const float* b[NCDE_MAX_LAYERS];
    const float* Wo;
    const float* bo;
    const float* coeffs;
    long long cs_b, cs_t;
    const float* z0;
    float* out;  // forward: [B, n_out, H]
    // adjoint
    const float* z_out;
    const float* grad_out;
    float* grad_z0;
    float* gpart;  // [gridDim.x][theta_size] per-workgroup parameter-gradient partials
    int gW_off[NCDE_MAX_LAYERS], gb_off[NCDE_MAX_LAYERS];
    int gWo_off, gbo_off, theta_size;
    int gacc_in_lds;  // generic adjoint: keep the partial in LDS and flush at the end
    // exact discrete backward (adjoint=False): record of every stage input, [(n*S + j)][B][H]
    float* stages;    // forward: written when non-null; backward: read
    int discrete;     // adjoint kernels: 1 = transpose the discretised solve instead of the continuous adjoint
    // batch-tiled adjoint (ncde_tiled.hip): per-stage records consumed by the output-layer gradient pass
    float* recA;      // x_L            [stage][sample tile][dlast/4][16][4]
    float* recB;      // x_L            [stage][sample tile][dlast][16]
    float* recC;      // w * cotangent  [stage][sample tile][H][16]
    float* recD;      // dX/dt          [stage][sample tile][C/4][16][4]
    int gstride;      // floats per workgroup partial in gpart (hidden-layer parameters only)
    // time-windowed backward of the batch-tiled family: one launch sweeps the reverse steps n = win_hi .. win_lo + 1; the
    // carried state (y, a per element) and the hidden-layer partial in gpart link consecutive windows
    int win_hi, win_lo, resume;
    float* carry;     // [2][n_workgroups][H * 16]
    // vector-field variants (generic family only): gated heads, reset net, evaluate / derivative input modes
    int field_kind, field_input;
    int d0;           // width of the field input u: H (matmul) or H + C
    int rows;         // rows of the heads: H*C (matmul) or H
    const float* Wg;
    const float* bg;
    const float* Wr;
    const float* br;
    int gWg_off, gbg_off, gWr_off, gbr_off;
    // general time axis (generic / variant families): host-built time plan (ncde_plan.h), NULL = default integer grid, step 1
    const int* plan;
    int n_steps_fwd, n_steps_adj;
    // batch-tiled family: fragment-ordered copies of the output layer and of the gate head (ncde_pack_panels)
    const float* Wo_pk;
    // split-fp16 kernels (ncde_fast.hip): range-fault word per workgroup; only_faulted = re-execute the faulted ones, skip the rest
    int* fault;
    int only_faulted;
    const float* Wg_pk;
    const unsigned* Wo_bf;   // split-bf16 copy (backward sweep: beside the fp32 copy, which feeds the transposed products)
    // variant family: LDS-resident copies of weight matrices (float offset into the dynamic LDS, -1 = streamed from L2);
    // a resident matrix [N][K] is stored with row stride K + 1, column K holding the bias
    int wres[NCDE_MAX_LAYERS], wres_o, wres_g, wres_r;
    // batch-tiled direct modes: LDS-resident row-major copies of the small matrices, [N][K] with row stride K + 4, the bias behind
    // (float offset into the dynamic LDS, 0 = streamed from L2)
    int tres[NCDE_MAX_LAYERS], tres_o, tres_g;
    // zero-padded problems (ncde_abi.hip pads H / layer widths to multiples of 16 and C to a multiple of 4 so that the batch-tiled
    // family covers any shape): the caller's tensors keep their REAL extents -- Hr = row width of z0 / out / z_out / grad_out /
    // grad_z0 / the stage record (units >= Hr are not read or written; they stay exactly 0), Cc = channels of the coefficient
    // tensor (channels >= Cc have dX/dt = X = 0).  Unpadded: Hr = H, Cc = C.
    int Hr, Cc;
    // XCD-cooperative output phase of the batch-tiled sweep (ncde_coop.h): packed register images of Wo per group member, the
    // inter-workgroup exchange area, the group counters + abort word, {sw, 1/sw}, members per group, number of groups
    const unsigned* coop_img;
    float* coop_x;
    float* coop_state;      // [n_workgroups][6][state slice]: the Butcher registers (y0, k.. / a0, k..) between bookkeeping phases
    unsigned* coop_sync;
    const float* coop_scale;
    int coop_M, coop_G;
    // round 6: the launch sequence's STATUS word (workspace; zeroed once per C-ABI call, never between its time windows): a cooperative
    // kernel that gives up (spin time-out) sets it, every later cooperative launch of the call returns at once, and the per-workgroup
    // kernels enqueued behind them with run_if = this word re-execute the whole pass (they return at once while it is 0)
    // round 6: records of the cooperative sweep for ncde_dwo_h2 (2-piece fp16, see ncde_tiled.hip): per (stage, sample tile) scales
    // {1/(sx sw) per sample [16], u_T, 1/u_T}, and the window's largest weighted cotangent bound / u_T as float bits (atomicMax)
    float* recS;
    unsigned* win_max;
    int dw2_parts;           // ncde_dwo_h2: part-groups (its grid is 1-D: row groups x parts)
    int dw2_accum;           // ncde_dwo_h2: add to the partial in gpartB (a later time window, or a later batch chunk) instead of overwriting it
    int Brec;                // batch stride of the stage record when this launch covers a CHUNK of the batch (0: = B)
    unsigned* coop_status;
    const unsigned* run_if;
    unsigned coop_spin;      // polls of a group counter before giving up
    int coop_inject;         // NCDE_FLAG_COOP_FAULT_INJECT: workgroup 1 skips its first arrival (tests the time-out path)
};

// ---- time plan (built on the host by ncde_time_plan_build, csrc/ncde_timeplan.hip; layout in 4-byte words) -------------
//   header  [8]                     : magic, S, n_fwd, n_adj, n_out, off_fwd, off_out, off_adj
//   forward step  [3 + 3S] x n_fwd  : dt (f32), first output row emitted after the step, number of such rows,
//                                     S x { piece index (i32), t - knot[idx] (f32), knot[idx+1] - knot[idx] (f32) }
//   output  [2] x n_out             : kind (0 = the step's y0, 1 = its y1, 2 = y0 + slope (y1 - y0)), slope (f32)
//   adjoint step [3 + 3S] x n_adj   : dt in negated time (f32), output row to reset to after the step (or -1), pad,
//                                     S x stage descriptors at the REAL time of the stage
#define NCDE_PLAN_MAGIC 0x4e43504c
#define NCDE_PLAN_HEADER 8
struct StageDesc {
    int idx;
    float frac;   // t - knot[idx]
    float kdt;    // knot[idx + 1] - knot[idx]  (1 on the default grid)
};
__host__ __device__ __forceinline__ int plan_step_words(int S) { return 3 + 3 * S; }
__host__ __device__ __forceinline__ int plan_off_fwd() { return NCDE_PLAN_HEADER; }
__host__ __device__ __forceinline__ int plan_off_out(int S, int n_fwd) { return NCDE_PLAN_HEADER + n_fwd * plan_step_words(S); }
__host__ __device__ __forceinline__ int plan_off_adj(int S, int n_fwd, int n_out) { return plan_off_out(S, n_fwd) + 2 * n_out; }
// A planned kernel walks the table with offsets computed from the CALLER's counts: refuse (leave the outputs untouched) a table
// whose own header says something else -- a stale plan, or one built for another method, would otherwise index out of bounds.
__device__ __forceinline__ bool plan_header_ok(const KArgs& a, int S) {
    const int* h = a.plan;
    return h[0] == NCDE_PLAN_MAGIC && h[1] == S && h[2] == a.n_steps_fwd && h[3] == a.n_steps_adj && h[4] == a.n_out;
}
__device__ __forceinline__ StageDesc plan_stage(const int* step, int j) {
    StageDesc d;
    d.idx = step[3 + 3 * j];
    d.frac = __int_as_float(step[4 + 3 * j]);
    d.kdt = __int_as_float(step[5 + 3 * j]);
    return d;
}

__device__ __forceinline__ int ru4(int x) { return (x + 3) & ~3; }
__device__ __forceinline__ int ru16(int x) { return (x + 15) & ~15; }

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    // D[16x16] += A[16x4] * B[4x16]; lane l supplies A[l&15][l>>4], B[l>>4][l&15];
    // D: col = l&15, row = 4*(l>>4) + reg.
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// tanh(x) = 1 - 2/(exp(2x)+1): mul, v_exp, add, v_rcp, fma (two transcendental issues).  Saturates correctly
// (exp -> inf gives 1, exp -> 0 gives -1); absolute error < 2e-7 over the whole range.
__device__ __forceinline__ float tanh_dev(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);  // exp(2x)
    const float r = __builtin_amdgcn_rcpf(e + 1.0f);
    return __builtin_fmaf(-2.0f, r, 1.0f);
}

// tanh of a pre-activation that was ALREADY scaled by 2*log2(e) (the specialised kernels fold that factor
// into Wo and bo when they load them): v_exp, add, v_rcp, fma.
#define NCDE_TANH_PRESCALE 2.8853900817779268f
__device__ __forceinline__ float tanh_prescaled(float x2l) {
    const float e = __builtin_amdgcn_exp2f(x2l);
    const float r = __builtin_amdgcn_rcpf(e + 1.0f);
    return __builtin_fmaf(-2.0f, r, 1.0f);
}

// relu as ONE instruction: v_med3_f32(x, 0, +inf).  (fmaxf() costs two -- hipcc canonicalises the MFMA result
// first.  Do NOT use inline asm here: an asm statement that reads an MFMA result register directly is invisible
// to the hazard recognizer, which then omits the MFMA->VALU wait states -- stale accumulators on some schedules.)
__device__ __forceinline__ float relu_dev(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, __builtin_inff()); }

// Number of stages and the fp32 stage-time offsets torchdiffeq produces with dt = 1
// (fixed_grid.py:6-29, rk_common.py:111-113: dt*(1/3), dt*(2/3) rounded to fp32).
__device__ __forceinline__ int n_stages(int method) { return method == NCDE_RK4_38 ? 4 : (method == NCDE_MIDPOINT ? 2 : 1); }
__device__ __forceinline__ float stage_offset(int method, int j) {
    if (method == NCDE_RK4_38) return j == 0 ? 0.0f : (j == 1 ? 0.333333343267440796f : (j == 2 ? 0.666666686534881592f : 1.0f));
    if (method == NCDE_MIDPOINT) return j == 0 ? 0.0f : 0.5f;
    return 0.0f;
}
// RK quadrature weight of stage j (what multiplies k_j in y1 = y0 + sum_j w_j k_j).
__device__ __forceinline__ float stage_weight(int method, int j) {
    if (method == NCDE_RK4_38) return (j == 0 || j == 3) ? 0.125f : 0.375f;
    if (method == NCDE_MIDPOINT) return j == 0 ? 0.0f : 1.0f;
    return 1.0f;
}

// bucketize(t, knots, right=False) - 1 clamped to [0, n_pieces-1]: the LEFT piece at an exact knot
// (interpolation_linear.py:212-219).
__device__ __forceinline__ int piece_index(float t, int n_pieces) {
    int idx = (int)__builtin_ceilf(t) - 1;
    idx = idx < 0 ? 0 : idx;
    return idx > n_pieces - 1 ? n_pieces - 1 : idx;
}

// sum over the 16 lanes of a row (lanes sharing lane>>4); every lane gets the total.
__device__ __forceinline__ float row16_sum(float v) {
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    return v;
}

struct StageCombine {
    // Butcher bookkeeping on one state array set (Y0, K1, K2) given the fresh stage derivative k and the step dt, in the
    // operation order of fixed_grid.py:6-29 / rk_common.py:106-114 (dt = 1 on the default grid: the products are exact).
    // Returns the next stage input (or the new state after the last stage); `last` tells which.
    __device__ static __forceinline__ float apply(int method, int j, float k, float dt, float& y0, float& k1, float& k2, bool& last) {
        last = false;
        if (method == NCDE_RK4_38) {
            if (j == 0) { k1 = k; return y0 + (dt * k) * 0.333333343267440796f; }
            if (j == 1) { k2 = k; return y0 + dt * (k - k1 * 0.333333343267440796f); }
            if (j == 2) { const float ys = y0 + dt * ((k1 - k2) + k); k2 = k2 + k; return ys; }
            last = true;
            y0 = y0 + (((k1 + 3.0f * k2) + k) * dt) * 0.125f;
            return y0;
        }
        if (method == NCDE_MIDPOINT) {
            if (j == 0) return y0 + k * (0.5f * dt);
            last = true;
            y0 = y0 + dt * k;
            return y0;
        }
        last = true;
        y0 = y0 + dt * k;
        return y0;
    }
};

// stage descriptor of stage j of forward step n on the default integer grid with step 1
__device__ __forceinline__ StageDesc default_stage(int method, float t, int n_pieces) {
    StageDesc d;
    d.idx = piece_index(t, n_pieces);
    d.frac = t - (float)d.idx;
    d.kdt = 1.0f;
    return d;
}

// Exact discrete backward on a planned time axis: cotangent that the outputs emitted after forward step `pstep` send to
// that step's y1 (part = 1) or y0 (part = 0) -- the transpose of the output pick / interpolation of solvers.py:108-116, 166-172.
__device__ __forceinline__ float plan_out_cotangent(const KArgs& a, const int* pstep, const int* pout, int part, long long brow, int h) {
    float acc = 0.0f;
    const int q0 = pstep[1], q1 = q0 + pstep[2];
    for (int q = q0; q < q1; ++q) {
        const int kind = pout[2 * q];
        const float slope = __int_as_float(pout[2 * q + 1]);
        const float g = a.grad_out[(brow + q) * a.Hr + h];
        if (part == 1) acc += kind == 1 ? g : (kind == 2 ? slope * g : 0.0f);
        else acc += kind == 0 ? g : (kind == 2 ? g - slope * g : 0.0f);
    }
    return acc;
}

