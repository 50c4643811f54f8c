// Host side of the C-ABI (include/ncde_hip.h): validation, kernel selection, launches.
// No torch types, no allocation, no synchronisation on the hot path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "ncde_common.h"
#include "ncde_fast.h"

extern "C" __global__ void ncde_fwd_generic(KArgs a);
extern "C" __global__ void ncde_adj_generic(KArgs a);

struct ReduceSegs {
    int n;
    int off[2 * NCDE_MAX_LAYERS + 2];
    int len[2 * NCDE_MAX_LAYERS + 2];
    float* dst[2 * NCDE_MAX_LAYERS + 2];
};
extern "C" __global__ void ncde_reduce_partials(const float* gpart, int n_part, int theta_size, ReduceSegs segs);

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(NCDE_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

constexpr int kLdsLimit = 160 * 1024;

inline int hru4(int x) { return (x + 3) & ~3; }
inline int hru16(int x) { return (x + 15) & ~15; }

int validate(const NcdeProblem* p) {
    if (!p) return fail(NCDE_ERR_INVALID, "problem is NULL");
    if (p->abi_version != NCDE_ABI_VERSION) return fail(NCDE_ERR_INVALID, "abi_version %d != %d", p->abi_version, NCDE_ABI_VERSION);
    if (p->batch < 1 || p->channels < 1 || p->hidden < 1) return fail(NCDE_ERR_INVALID, "batch/channels/hidden must be >= 1");
    if (p->n_knots < 2) return fail(NCDE_ERR_INVALID, "Must have a time dimension of size at least 2 (n_knots=%d)", p->n_knots);
    if (p->interp != NCDE_INTERP_LINEAR && p->interp != NCDE_INTERP_CUBIC) return fail(NCDE_ERR_INVALID, "unknown interp %d", p->interp);
    if (p->method != NCDE_EULER && p->method != NCDE_MIDPOINT && p->method != NCDE_RK4_38)
        return fail(NCDE_ERR_INVALID, "Invalid method %d. Must be one of {euler, midpoint, rk4}", p->method);
    if (p->output != NCDE_OUT_INTERVAL && p->output != NCDE_OUT_KNOTS) return fail(NCDE_ERR_INVALID, "unknown output mode %d", p->output);
    if (p->n_layers < 0 || p->n_layers > NCDE_MAX_LAYERS) return fail(NCDE_ERR_INVALID, "n_layers %d outside [0, %d]", p->n_layers, NCDE_MAX_LAYERS);
    int d = p->hidden;
    for (int l = 0; l < p->n_layers; ++l) {
        if (p->layer_in[l] != d) return fail(NCDE_ERR_INVALID, "layer %d: in=%d does not chain from %d", l, p->layer_in[l], d);
        if (p->layer_out[l] < 1) return fail(NCDE_ERR_INVALID, "layer %d: out=%d", l, p->layer_out[l]);
        if (!p->layer_W[l] || !p->layer_b[l]) return fail(NCDE_ERR_INVALID, "layer %d: NULL weight/bias", l);
        d = p->layer_out[l];
    }
    if (!p->Wo || !p->bo || !p->coeffs || !p->z0) return fail(NCDE_ERR_INVALID, "NULL Wo/bo/coeffs/z0");
    if (p->coeffs_stride_t < (p->interp == NCDE_INTERP_CUBIC ? 4 : 1) * (int64_t)p->channels)
        return fail(NCDE_ERR_INVALID, "coeffs_stride_t %lld too small", (long long)p->coeffs_stride_t);
    return NCDE_OK;
}

struct Layout {
    int Hp, Cp, Dp, HS, DS, L;
    int theta_size;
    int gW_off[NCDE_MAX_LAYERS], gb_off[NCDE_MAX_LAYERS], gWo_off, gbo_off;
    int dlast;
    int n_wg;
    size_t lds_fwd, lds_adj;
    int gacc_in_lds;
};

Layout make_layout(const NcdeProblem* p) {
    Layout y{};
    y.L = p->n_layers;
    y.Hp = hru16(p->hidden);
    y.Cp = hru4(p->channels);
    y.Dp = y.Hp;
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
    y.gWo_off = off; off += p->hidden * p->channels * y.dlast;
    y.gbo_off = off; off += p->hidden * p->channels;
    y.theta_size = off;
    y.n_wg = (p->batch + NCDE_TILE - 1) / NCDE_TILE;
    y.lds_fwd = sizeof(float) * (size_t)(5 * y.HS + 2 * y.DS + y.Cp * 16);
    size_t adj = sizeof(float) * (size_t)(10 * y.HS + ((y.L > 0 ? y.L : 1) + 4) * y.DS + y.Cp * 16 + 4 * 16 * 17);
    y.gacc_in_lds = adj + sizeof(float) * (size_t)y.theta_size <= (size_t)kLdsLimit;
    y.lds_adj = adj + (y.gacc_in_lds ? sizeof(float) * (size_t)y.theta_size : 0);
    return y;
}

void fill_kargs(const NcdeProblem* p, const Layout& y, KArgs* a) {
    memset(a, 0, sizeof(*a));
    a->B = p->batch; a->T = p->n_knots; a->C = p->channels; a->H = p->hidden;
    a->interp = p->interp; a->method = p->method; a->output = p->output; a->n_layers = p->n_layers;
    a->n_pieces = p->n_knots - 1;
    a->n_out = p->output == NCDE_OUT_KNOTS ? p->n_knots : 2;
    for (int l = 0; l < p->n_layers; ++l) {
        a->din[l] = p->layer_in[l]; a->dout[l] = p->layer_out[l];
        a->W[l] = p->layer_W[l]; a->b[l] = p->layer_b[l];
        a->gW_off[l] = y.gW_off[l]; a->gb_off[l] = y.gb_off[l];
    }
    a->Wo = p->Wo; a->bo = p->bo; a->coeffs = p->coeffs;
    a->cs_b = p->coeffs_stride_b; a->cs_t = p->coeffs_stride_t;
    a->z0 = p->z0;
    a->gWo_off = y.gWo_off; a->gbo_off = y.gbo_off; a->theta_size = y.theta_size;
    a->gacc_in_lds = y.gacc_in_lds;
}

int generic_supported(const NcdeProblem* p, const Layout& y, int pass) {
    if (pass == 0 && y.lds_fwd > (size_t)kLdsLimit)
        return fail(NCDE_ERR_UNSUPPORTED, "generic forward needs %zu B of LDS (> %d)", y.lds_fwd, kLdsLimit);
    if (pass == 1) {
        if (y.lds_adj > (size_t)kLdsLimit) return fail(NCDE_ERR_UNSUPPORTED, "generic adjoint needs %zu B of LDS (> %d)", y.lds_adj, kLdsLimit);
        if (y.dlast > 128) return fail(NCDE_ERR_UNSUPPORTED, "generic adjoint supports a last hidden width <= 128 (got %d)", y.dlast);
    }
    (void)p;
    return NCDE_OK;
}

// pick the kernel family: 1 = fast, 0 = generic, <0 = error
int select_family(const NcdeProblem* p, const Layout& y, int pass) {
    const bool fast_ok = ncde_fast_supported(p, pass);
    if (p->flags & NCDE_FLAG_FORCE_FAST) {
        if (!fast_ok) return fail(NCDE_ERR_UNSUPPORTED, "no shape-specialised kernel for this problem (pass %d)", pass);
        return 1;
    }
    if (fast_ok && !(p->flags & NCDE_FLAG_FORCE_GENERIC)) return 1;
    const int rc = generic_supported(p, y, pass);
    return rc == NCDE_OK ? 0 : rc;
}

int launch_reduce(const NcdeProblem* p, const Layout& y, const NcdeGrads* g, const float* gpart, int n_part, hipStream_t st) {
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
    segs.off[n] = y.gWo_off; segs.len[n] = p->hidden * p->channels * y.dlast; segs.dst[n] = g->grad_Wo; ++n;
    segs.off[n] = y.gbo_off; segs.len[n] = p->hidden * p->channels; segs.dst[n] = g->grad_bo; ++n;
    segs.n = n;
    for (int i = 0; i < n; ++i)
        if (!segs.dst[i]) return fail(NCDE_ERR_INVALID, "NcdeGrads: NULL destination for parameter segment %d", i);
    hipLaunchKernelGGL(ncde_reduce_partials, dim3((y.theta_size + 255) / 256), dim3(256), 0, st, gpart, n_part, y.theta_size, segs);
    HIP_TRY(hipGetLastError());
    return NCDE_OK;
}

int launch_forward(const NcdeProblem* p, const Layout& y, int family, float* out, void* ws, size_t ws_bytes, hipStream_t st) {
    if (family == 1) {
        const int rc = ncde_fast_forward(p, out, ws, ws_bytes, st);
        if (rc != NCDE_OK) return fail(rc, "fast forward launch failed");
        return NCDE_OK;
    }
    KArgs a;
    fill_kargs(p, y, &a);
    a.out = out;
    HIP_TRY(hipFuncSetAttribute((const void*)ncde_fwd_generic, hipFuncAttributeMaxDynamicSharedMemorySize, (int)y.lds_fwd));
    hipLaunchKernelGGL(ncde_fwd_generic, dim3(y.n_wg), dim3(256), y.lds_fwd, st, a);
    HIP_TRY(hipGetLastError());
    return NCDE_OK;
}

int launch_adjoint(const NcdeProblem* p, const Layout& y, int family, const float* z_out, const float* grad_out,
                   const NcdeGrads* g, void* ws, size_t ws_bytes, hipStream_t st, bool main_kernel_only) {
    if (family == 1) {
        const int rc = ncde_fast_adjoint(p, z_out, grad_out, g, ws, ws_bytes, st, main_kernel_only);
        if (rc != NCDE_OK) return fail(rc, "fast adjoint launch failed");
        return NCDE_OK;
    }
    KArgs a;
    fill_kargs(p, y, &a);
    a.z_out = z_out; a.grad_out = grad_out; a.grad_z0 = g->grad_z0;
    a.gpart = (float*)ws;
    HIP_TRY(hipFuncSetAttribute((const void*)ncde_adj_generic, hipFuncAttributeMaxDynamicSharedMemorySize, (int)y.lds_adj));
    hipLaunchKernelGGL(ncde_adj_generic, dim3(y.n_wg), dim3(256), y.lds_adj, st, a);
    HIP_TRY(hipGetLastError());
    if (main_kernel_only) return NCDE_OK;
    return launch_reduce(p, y, g, (const float*)ws, y.n_wg, st);
}

}  // namespace

extern "C" {

int ncde_version(void) { return NCDE_ABI_VERSION; }
const char* ncde_last_error_string(void) { return g_err; }

int ncde_num_outputs(const NcdeProblem* p) {
    const int rc = validate(p);
    if (rc != NCDE_OK) return rc;
    return p->output == NCDE_OUT_KNOTS ? p->n_knots : 2;
}

int64_t ncde_workspace_bytes(const NcdeProblem* p, int pass) {
    const int rc = validate(p);
    if (rc != NCDE_OK) return rc;
    const Layout y = make_layout(p);
    const int fam = select_family(p, y, pass);
    if (fam < 0) return fam;
    if (fam == 1) return ncde_fast_workspace_bytes(p, pass);
    if (pass == 0) return 256;
    return (int64_t)sizeof(float) * (int64_t)y.n_wg * (int64_t)y.theta_size + 256;
}

const char* ncde_kernel_name(const NcdeProblem* p, int pass) {
    if (validate(p) != NCDE_OK) return nullptr;
    const Layout y = make_layout(p);
    const int fam = select_family(p, y, pass);
    if (fam < 0) return nullptr;
    return fam == 1 ? ncde_fast_kernel_name(p, pass) : (pass == 0 ? "ncde_fwd_generic" : "ncde_adj_generic");
}

int ncde_forward(const NcdeProblem* p, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = validate(p);
    if (rc != NCDE_OK) return rc;
    if (!out) return fail(NCDE_ERR_INVALID, "out is NULL");
    const Layout y = make_layout(p);
    const int fam = select_family(p, y, 0);
    if (fam < 0) return fam;
    const int64_t need = ncde_workspace_bytes(p, 0);
    if (need > 0 && (!workspace || (int64_t)workspace_bytes < need)) return fail(NCDE_ERR_WORKSPACE, "workspace %zu B < %lld B", workspace_bytes, (long long)need);
    return launch_forward(p, y, fam, out, workspace, workspace_bytes, (hipStream_t)stream);
}

int ncde_adjoint(const NcdeProblem* p, const float* z_out, const float* grad_out, const NcdeGrads* grads, void* workspace,
                 size_t workspace_bytes, void* stream) {
    int rc = validate(p);
    if (rc != NCDE_OK) return rc;
    if (!z_out || !grad_out || !grads || !grads->grad_z0) return fail(NCDE_ERR_INVALID, "NULL z_out/grad_out/grads");
    const Layout y = make_layout(p);
    const int fam = select_family(p, y, 1);
    if (fam < 0) return fam;
    const int64_t need = ncde_workspace_bytes(p, 1);
    if (!workspace || (int64_t)workspace_bytes < need) return fail(NCDE_ERR_WORKSPACE, "workspace %zu B < %lld B", workspace_bytes, (long long)need);
    return launch_adjoint(p, y, fam, z_out, grad_out, grads, workspace, workspace_bytes, (hipStream_t)stream, false);
}

int ncde_time_kernel(const NcdeProblem* p, int pass, float* out, const float* grad_out, const NcdeGrads* grads, void* workspace,
                     size_t workspace_bytes, void* stream, int iters, float* ms_per_launch) {
    int rc = validate(p);
    if (rc != NCDE_OK) return rc;
    if (iters < 1 || !ms_per_launch) return fail(NCDE_ERR_INVALID, "iters < 1 or NULL result");
    const Layout y = make_layout(p);
    const int fam = select_family(p, y, pass);
    if (fam < 0) return fam;
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    // one untimed launch (also sets function attributes), then `iters` timed ones
    for (int it = -1; it < iters; ++it) {
        if (it == 0) HIP_TRY(hipEventRecord(e0, st));
        if (pass == 0) rc = launch_forward(p, y, fam, out, workspace, workspace_bytes, st);
        else rc = launch_adjoint(p, y, fam, out, grad_out, grads, workspace, workspace_bytes, st, true);
        if (rc != NCDE_OK) return rc;
    }
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    HIP_TRY(hipEventDestroy(e0));
    HIP_TRY(hipEventDestroy(e1));
    *ms_per_launch = ms / (float)iters;
    return NCDE_OK;
}

}  // extern "C"
