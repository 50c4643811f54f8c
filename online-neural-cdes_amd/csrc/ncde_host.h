// Host-side helpers shared by ncde_abi.hip and ncde_fast.hip: parameter-gradient layout, K4 launch.
#pragma once
#include <mutex>
#include <map>
#include <utility>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include "ncde_common.h"

struct ReduceSegs {
    int n;
    int off[2 * NCDE_MAX_LAYERS + 6];
    int len[2 * NCDE_MAX_LAYERS + 6];
    float* dst[2 * NCDE_MAX_LAYERS + 6];
};
extern "C" __global__ void ncde_reduce_partials(const float* gpart, int n_part, int theta_size, ReduceSegs segs);

constexpr int kLdsLimit = 160 * 1024;

// Development switches are environment variables ONLY in builds with -DNCDE_DEV_KNOBS; the shipped library is stateless and
// reads nothing but its arguments.
inline const char* ncde_dev_env(const char* name) {
#ifdef NCDE_DEV_KNOBS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// Raise a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) only when a launch needs more than what was
// already granted for this (device, kernel) -- not a hipFuncSetAttribute call in front of every launch.  (Granting a flat 160 KB
// would fail for kernels that also hold static __shared__ arrays.)
inline hipError_t ncde_lds_optin(const void* fn, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> granted;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lk(mu);
    size_t& have = granted[{dev, fn}];
    if (bytes <= have) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) have = bytes;
    return e;
}
inline int hru4(int x) { return (x + 3) & ~3; }
inline int hru16(int x) { return (x + 15) & ~15; }

struct Layout {
    int Hp, Cp, Dp, HS, DS, L;
    int theta_size;
    int gW_off[NCDE_MAX_LAYERS], gb_off[NCDE_MAX_LAYERS], gWo_off, gbo_off;
    int gWg_off, gbg_off, gWr_off, gbr_off;
    int d0, rows;          // field input width, head rows
    bool variant;          // gated field or evaluate / derivative input (generic family only)
    int dlast;
    int n_wg;
    size_t lds_fwd, lds_adj;
    int gacc_in_lds;
};

inline Layout make_layout(const NcdeProblem* p) {
    Layout y{};
    y.L = p->n_layers;
    y.Hp = hru16(p->hidden);
    y.Cp = hru4(p->channels);
    y.variant = p->field_kind != NCDE_FIELD_ORIGINAL || p->field_input != NCDE_INPUT_MATMUL;
    y.d0 = p->field_input == NCDE_INPUT_MATMUL ? p->hidden : p->hidden + p->channels;
    y.rows = p->field_input == NCDE_INPUT_MATMUL ? p->hidden * p->channels : p->hidden;
    y.Dp = std::max(y.Hp, hru16(y.d0));
    for (int l = 0; l < y.L; ++l) y.Dp = std::max(y.Dp, hru16(p->layer_out[l]));
    y.HS = y.Hp * 16;
    y.DS = y.Dp * 16;
    y.dlast = y.L ? p->layer_out[y.L - 1] : p->hidden;
    int off = 0;
    for (int l = 0; l < y.L; ++l) {
        int prevW = -1, prevB = -1;
        for (int q = 0; q < l; ++q) {
            if (p->layer_W[q] == p->layer_W[l]) prevW = q;
            if (p->layer_b[q] == p->layer_b[l]) prevB = q;
        }
        if (prevW >= 0) y.gW_off[l] = y.gW_off[prevW];
        else { y.gW_off[l] = off; off += p->layer_out[l] * p->layer_in[l]; }
        if (prevB >= 0) y.gb_off[l] = y.gb_off[prevB];
        else { y.gb_off[l] = off; off += p->layer_out[l]; }
    }
    y.gWo_off = off; off += y.rows * y.dlast;
    y.gbo_off = off; off += y.rows;
    y.gWg_off = y.gbg_off = y.gWr_off = y.gbr_off = 0;
    if (p->field_kind != NCDE_FIELD_ORIGINAL) {
        y.gWg_off = off; off += y.rows * y.dlast;
        y.gbg_off = off; off += y.rows;
    }
    if (p->field_kind == NCDE_FIELD_GRU) {
        y.gWr_off = off; off += y.d0 * y.d0;
        y.gbr_off = off; off += y.d0;
    }
    y.theta_size = off;
    y.n_wg = (p->batch + NCDE_TILE - 1) / NCDE_TILE;
    y.lds_fwd = sizeof(float) * (size_t)(5 * y.HS + 2 * y.DS + y.Cp * 16);
    size_t adj = sizeof(float) * (size_t)(10 * y.HS + ((y.L > 0 ? y.L : 1) + 4) * y.DS + y.Cp * 16 + 4 * 16 * 17);
    y.gacc_in_lds = adj + sizeof(float) * (size_t)y.theta_size <= (size_t)kLdsLimit;
    y.lds_adj = adj + (y.gacc_in_lds ? sizeof(float) * (size_t)y.theta_size : 0);
    return y;
}


// fills the dims / pointers / gradient offsets common to every kernel family
inline void fill_kargs(const NcdeProblem* p, const Layout& y, KArgs* a) {
    memset(a, 0, sizeof(*a));
    a->B = p->batch; a->T = p->n_knots; a->C = p->channels; a->H = p->hidden;
    a->interp = p->interp; a->method = p->method; a->output = p->output; a->n_layers = p->n_layers;
    a->n_pieces = p->n_knots - 1;
    a->n_out = p->output == NCDE_OUT_TIMES ? p->n_t_out : (p->output == NCDE_OUT_KNOTS ? p->n_knots : 2);
    if (p->output == NCDE_OUT_TIMES) {
        a->plan = (const int*)p->time_plan;
        a->n_steps_fwd = p->n_steps_fwd;
        a->n_steps_adj = p->n_steps_adj;
    }
    for (int l = 0; l < p->n_layers; ++l) {
        a->din[l] = p->layer_in[l]; a->dout[l] = p->layer_out[l];
        a->W[l] = p->layer_W[l]; a->b[l] = p->layer_b[l];
        a->gW_off[l] = y.gW_off[l]; a->gb_off[l] = y.gb_off[l];
    }
    a->Wo = p->Wo; a->bo = p->bo; a->coeffs = p->coeffs;
    a->cs_b = p->coeffs_stride_b; a->cs_t = p->coeffs_stride_t;
    a->z0 = p->z0;
    a->gWo_off = y.gWo_off; a->gbo_off = y.gbo_off; a->theta_size = y.theta_size;
    a->field_kind = p->field_kind; a->field_input = p->field_input; a->d0 = y.d0; a->rows = y.rows;
    a->Wg = p->Wg; a->bg = p->bg; a->Wr = p->Wr; a->br = p->br;
    a->gWg_off = y.gWg_off; a->gbg_off = y.gbg_off; a->gWr_off = y.gWr_off; a->gbr_off = y.gbr_off;
    a->gacc_in_lds = y.gacc_in_lds;
    // internal (zero-padded problems, see KArgs): reserved_ = (real hidden size << 12) | real channel count, 0 = not padded
    a->Hr = p->reserved_ ? (p->reserved_ >> 12) : p->hidden;
    a->Cc = p->reserved_ ? (p->reserved_ & 0xFFF) : p->channels;
}

// K4: sum the per-workgroup partials and scatter into the caller's buffers. 0 = ok, else NcdeStatus.
inline int launch_reduce_partials(const NcdeProblem* p, const Layout& y, const NcdeGrads* g, const float* gpart, int n_part,
                                  hipStream_t st) {
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
    if (p->field_kind != NCDE_FIELD_ORIGINAL) {
        segs.off[n] = y.gWg_off; segs.len[n] = y.rows * y.dlast; segs.dst[n] = g->grad_Wg; ++n;
        segs.off[n] = y.gbg_off; segs.len[n] = y.rows; segs.dst[n] = g->grad_bg; ++n;
    }
    if (p->field_kind == NCDE_FIELD_GRU) {
        segs.off[n] = y.gWr_off; segs.len[n] = y.d0 * y.d0; segs.dst[n] = g->grad_Wr; ++n;
        segs.off[n] = y.gbr_off; segs.len[n] = y.d0; segs.dst[n] = g->grad_br; ++n;
    }
    segs.n = n;
    for (int i = 0; i < n; ++i)
        if (!segs.dst[i]) return NCDE_ERR_INVALID;
    hipLaunchKernelGGL(ncde_reduce_partials, dim3((y.theta_size + 255) / 256), dim3(256), 0, st, gpart, n_part, y.theta_size, segs);
    return hipGetLastError() == hipSuccess ? NCDE_OK : NCDE_ERR_HIP;
}
