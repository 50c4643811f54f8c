// Host side of the C-ABI (include/ncde_hip.h): validation, kernel selection, launches.
// No torch types, no allocation, no synchronisation on the hot path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstring>

#include "ncde_adaptive.h"
#include "ncde_common.h"
#include "ncde_fast.h"
#include "ncde_host.h"
#include "ncde_tiled.h"
#include "ncde_timeplan.h"
#include "ncde_variant.h"

extern "C" __global__ void ncde_fwd_generic(KArgs a);
extern "C" __global__ void ncde_adj_generic(KArgs a);

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

// Copy the caller's struct into a full version-2 struct: a version-1 caller passes a shorter struct (no trailing
// field_kind .. br members), which reads as the original field with the matmul input.
int normalize(const NcdeProblem* in, NcdeProblem* out) {
    if (!in) return fail(NCDE_ERR_INVALID, "problem is NULL");
    if (in->abi_version < 1 || in->abi_version > NCDE_ABI_VERSION)
        return fail(NCDE_ERR_INVALID, "abi_version %d not in [1, %d]", in->abi_version, NCDE_ABI_VERSION);
    memset(out, 0, sizeof(*out));
    // version 1 ends before field_kind, version 2 before time_plan: the missing tail reads as zeros (original field, matmul
    // input, default time axis)
    memcpy(out, in, in->abi_version >= 3 ? sizeof(NcdeProblem) : (in->abi_version == 2 ? offsetof(NcdeProblem, time_plan) : offsetof(NcdeProblem, field_kind)));
    out->abi_version = NCDE_ABI_VERSION;
    return NCDE_OK;
}

int validate(const NcdeProblem* p) {
    if (!p) return fail(NCDE_ERR_INVALID, "problem is NULL");
    if (p->batch < 1 || p->channels < 1 || p->hidden < 1) return fail(NCDE_ERR_INVALID, "batch/channels/hidden must be >= 1");
    if (p->n_knots < 2) return fail(NCDE_ERR_INVALID, "Must have a time dimension of size at least 2 (n_knots=%d)", p->n_knots);
    if (p->interp != NCDE_INTERP_LINEAR && p->interp != NCDE_INTERP_CUBIC) return fail(NCDE_ERR_INVALID, "unknown interp %d", p->interp);
    if (p->method != NCDE_EULER && p->method != NCDE_MIDPOINT && p->method != NCDE_RK4_38)
        return fail(NCDE_ERR_INVALID, "Invalid method %d. Must be one of {euler, midpoint, rk4}", p->method);
    if (p->output != NCDE_OUT_INTERVAL && p->output != NCDE_OUT_KNOTS && p->output != NCDE_OUT_TIMES) return fail(NCDE_ERR_INVALID, "unknown output mode %d", p->output);
    if (p->output == NCDE_OUT_TIMES) {
        if (!p->time_plan) return fail(NCDE_ERR_INVALID, "output = NCDE_OUT_TIMES needs a time plan (ncde_time_plan_build)");
        if (p->n_t_out < 2 || p->n_steps_fwd < 1 || p->n_steps_adj < 1) return fail(NCDE_ERR_INVALID, "time plan counts: n_t_out %d, n_steps_fwd %d, n_steps_adj %d", p->n_t_out, p->n_steps_fwd, p->n_steps_adj);
    } else if (p->time_plan) {
        return fail(NCDE_ERR_INVALID, "a time plan is only read with output = NCDE_OUT_TIMES");
    }
    if (p->n_layers < 0 || p->n_layers > NCDE_MAX_LAYERS) return fail(NCDE_ERR_INVALID, "n_layers %d outside [0, %d]", p->n_layers, NCDE_MAX_LAYERS);
    if (p->field_kind < NCDE_FIELD_ORIGINAL || p->field_kind > NCDE_FIELD_GRU) return fail(NCDE_ERR_INVALID, "unknown field_kind %d", p->field_kind);
    if (p->field_input < NCDE_INPUT_MATMUL || p->field_input > NCDE_INPUT_DERIVATIVE)
        return fail(NCDE_ERR_INVALID, "vector_field_type string not recognised (field_input %d)", p->field_input);
    if (p->field_kind != NCDE_FIELD_ORIGINAL && (!p->Wg || !p->bg)) return fail(NCDE_ERR_INVALID, "gated field: NULL Wg/bg");
    if (p->field_kind == NCDE_FIELD_GRU && (!p->Wr || !p->br)) return fail(NCDE_ERR_INVALID, "GRU field: NULL Wr/br");
    int d = p->field_input == NCDE_INPUT_MATMUL ? p->hidden : p->hidden + p->channels;
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

int generic_supported(const NcdeProblem* p, const Layout& y, int pass) {
    if (pass == 0 && y.lds_fwd > (size_t)kLdsLimit)
        return fail(NCDE_ERR_UNSUPPORTED, "generic forward needs %zu B of LDS (> %d)", y.lds_fwd, kLdsLimit);
    if (pass >= 1) {
        if (y.lds_adj > (size_t)kLdsLimit) return fail(NCDE_ERR_UNSUPPORTED, "generic adjoint needs %zu B of LDS (> %d)", y.lds_adj, kLdsLimit);
        if (y.dlast > 128) return fail(NCDE_ERR_UNSUPPORTED, "generic adjoint supports a last hidden width <= 128 (got %d)", y.dlast);
    }
    (void)p;
    return NCDE_OK;
}

// pick the kernel family: 1 = fast (shape-specialised), 2 = tiled (batch-tiled, large hidden), 0 = generic, <0 = error
int select_family(const NcdeProblem* p, const Layout& y, int pass) {
    if (p->output == NCDE_OUT_TIMES) {   // general time axis: the plan-driven kernels -- batch-tiled where the shape allows
                                         // (multiples of 16 / 4; 2.8x the generic family at cfg2 widths, 10x at cfg5's), else generic / variant
        if (p->flags & NCDE_FLAG_FORCE_FAST) return fail(NCDE_ERR_UNSUPPORTED, "the shape-specialised kernels run the default time axis only");
        const bool tiled_ok = !(p->flags & NCDE_FLAG_FORCE_GENERIC) && ncde_tiled_supported(p, pass);
        if (p->flags & NCDE_FLAG_FORCE_TILED) {
            if (!tiled_ok) return fail(NCDE_ERR_UNSUPPORTED, "the batch-tiled family does not cover this problem (pass %d)", pass);
            return 2;
        }
        if (tiled_ok && ncde_tiled_preferred(p, pass)) return 2;
        if (y.variant) {
            if (!ncde_variant_supported(p, pass)) return fail(NCDE_ERR_UNSUPPORTED, "vector-field variant outside what ncde_variant.hip covers (pass %d)", pass);
            return 3;
        }
        const int rc = generic_supported(p, y, pass);
        return rc == NCDE_OK ? 0 : rc;
    }
    if (y.variant) {   // gated fields / evaluate / derivative inputs: the batch-tiled family knows the minimal-gated field;
                       // everything else runs on their own kernels on the generic structure
        if (!(p->flags & (NCDE_FLAG_FORCE_GENERIC | NCDE_FLAG_FORCE_FAST)) && ncde_tiled_supported(p, pass) && ncde_tiled_preferred(p, pass)) return 2;
        if ((p->flags & NCDE_FLAG_FORCE_FAST) || !ncde_variant_supported(p, pass))
            return fail(NCDE_ERR_UNSUPPORTED, "vector-field variant outside what ncde_variant.hip covers (pass %d)", pass);
        return 3;
    }
    if ((p->flags & NCDE_FLAG_FORCE_TILED) && ncde_tiled_supported(p, pass)) return 2;   // also ahead of a specialised kernel
    const bool fast_ok = ncde_fast_supported(p, pass);
    if (p->flags & NCDE_FLAG_FORCE_FAST) {
        if (!fast_ok) return fail(NCDE_ERR_UNSUPPORTED, "no shape-specialised kernel for this problem (pass %d)", pass);
        return 1;
    }
    if (fast_ok && !(p->flags & NCDE_FLAG_FORCE_GENERIC)) return 1;
    if (!(p->flags & NCDE_FLAG_FORCE_GENERIC) && ncde_tiled_supported(p, pass) && ncde_tiled_preferred(p, pass)) return 2;
    const int rc = generic_supported(p, y, pass);
    return rc == NCDE_OK ? 0 : rc;
}

int launch_reduce(const NcdeProblem* p, const Layout& y, const NcdeGrads* g, const float* gpart, int n_part, hipStream_t st) {
    const int rc = launch_reduce_partials(p, y, g, gpart, n_part, st);
    if (rc == NCDE_ERR_INVALID) return fail(rc, "NcdeGrads: NULL destination for a parameter gradient");
    if (rc != NCDE_OK) return fail(rc, "ncde_reduce_partials launch failed");
    return NCDE_OK;
}

int launch_forward(const NcdeProblem* p, const Layout& y, int family, float* out, float* stages, void* ws, size_t ws_bytes,
                   hipStream_t st) {
    if (family == 1) {
        const int rc = ncde_fast_forward(p, out, stages, ws, ws_bytes, st);
        if (rc != NCDE_OK) return fail(rc, "fast forward launch failed");
        return NCDE_OK;
    }
    if (family == 3) {
        const int rc = ncde_variant_forward(p, out, stages, st);
        if (rc != NCDE_OK) return fail(rc, "variant forward launch failed");
        return NCDE_OK;
    }
    if (family == 2) {
        const int rc = ncde_tiled_forward(p, out, stages, ws, ws_bytes, st);
        if (rc != NCDE_OK) return fail(rc, "tiled forward launch failed");
        return NCDE_OK;
    }
    KArgs a;
    fill_kargs(p, y, &a);
    a.out = out;
    a.stages = stages;
    HIP_TRY(ncde_lds_optin((const void*)ncde_fwd_generic, y.lds_fwd));
    hipLaunchKernelGGL(ncde_fwd_generic, dim3(y.n_wg), dim3(256), y.lds_fwd, st, a);
    HIP_TRY(hipGetLastError());
    return NCDE_OK;
}

// discrete = false: continuous adjoint, `src` = z_out; discrete = true: exact backward, `src` = the stage record
int launch_adjoint(const NcdeProblem* p, const Layout& y, int family, const float* src, const float* grad_out,
                   const NcdeGrads* g, void* ws, size_t ws_bytes, hipStream_t st, bool main_kernel_only, bool discrete) {
    if (family == 1) {
        const int rc = ncde_fast_adjoint(p, src, grad_out, g, ws, ws_bytes, st, main_kernel_only, discrete);
        if (rc != NCDE_OK) return fail(rc, "fast adjoint launch failed");
        return NCDE_OK;
    }
    if (family == 3) {
        const int rc = ncde_variant_adjoint(p, src, grad_out, g, ws, st, main_kernel_only, discrete);
        if (rc == NCDE_ERR_INVALID) return fail(rc, "NcdeGrads: NULL destination for a parameter gradient");
        if (rc != NCDE_OK) return fail(rc, "variant adjoint launch failed");
        return NCDE_OK;
    }
    if (family == 2) {
        const int rc = ncde_tiled_adjoint(p, src, grad_out, g, ws, ws_bytes, st, main_kernel_only, discrete);
        if (rc == NCDE_ERR_INVALID) return fail(rc, "NcdeGrads: NULL destination for a parameter gradient");
        if (rc != NCDE_OK) return fail(rc, "tiled adjoint launch failed");
        return NCDE_OK;
    }
    KArgs a;
    fill_kargs(p, y, &a);
    a.grad_out = grad_out; a.grad_z0 = g->grad_z0;
    if (discrete) { a.stages = const_cast<float*>(src); a.discrete = 1; }
    else a.z_out = src;
    a.gpart = (float*)ws;
    HIP_TRY(ncde_lds_optin((const void*)ncde_adj_generic, y.lds_adj));
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
    NcdeProblem q_;
    int rc = normalize(p, &q_);
    if (rc != NCDE_OK) return rc;
    p = &q_;
    rc = validate(p);
    if (rc != NCDE_OK) return rc;
    return p->output == NCDE_OUT_TIMES ? p->n_t_out : (p->output == NCDE_OUT_KNOTS ? p->n_knots : 2);
}

namespace {
// common front end of the dopri5 calls: the problem is validated as a default-axis one (method / output are not used)
int dopri5_prepare(const NcdeProblem* p, const NcdeTimeSpec* ts, const NcdeAdaptiveOptions* opt, int adj, NcdeProblem* q) {
    int rc = normalize(p, q);
    if (rc != NCDE_OK) return rc;
    q->method = NCDE_RK4_38;
    q->output = NCDE_OUT_INTERVAL;
    q->time_plan = nullptr;
    rc = validate(q);
    if (rc != NCDE_OK) return rc;
    if (!ts || !ts->t || ts->n_t < 2) return fail(NCDE_ERR_INVALID, "time spec: need >= 2 output times");
    if (opt && (!(opt->rtol > 0.0) || !(opt->atol >= 0.0))) return fail(NCDE_ERR_INVALID, "rtol must be > 0 and atol >= 0");
    char why[200] = "";
    if (!ncde_dp_supported(q, adj, why, sizeof(why))) return fail(NCDE_ERR_UNSUPPORTED, "%s", why);
    return NCDE_OK;
}
}  // namespace

int64_t ncde_dopri5_workspace_bytes(const NcdeProblem* p, const NcdeTimeSpec* ts, int pass) {
    NcdeProblem q_;
    NcdeAdaptiveOptions o{};
    o.rtol = 1e-4;
    const int rc = dopri5_prepare(p, ts, &o, pass != 0, &q_);
    if (rc != NCDE_OK) return rc;
    return ncde_dp_workspace_bytes(&q_, ts->n_t, pass);
}

int64_t ncde_dopri5_record_bytes(const NcdeProblem* p, const NcdeTimeSpec* ts, const NcdeAdaptiveOptions* opt) {
    NcdeProblem q_;
    if (!opt) return fail(NCDE_ERR_INVALID, "options are NULL");
    const int rc = dopri5_prepare(p, ts, opt, 0, &q_);
    if (rc != NCDE_OK) return rc;
    return ncde_dp_record_bytes(&q_, ts, opt);
}

int ncde_dopri5_forward_record(const NcdeProblem* p, const NcdeTimeSpec* ts, const NcdeAdaptiveOptions* opt, float* out, void* record,
                               size_t record_bytes, void* workspace, size_t workspace_bytes, void* stream, NcdeAdaptiveStats* stats) {
    NcdeProblem q_;
    if (!opt) return fail(NCDE_ERR_INVALID, "options are NULL");
    int rc = dopri5_prepare(p, ts, opt, 0, &q_);
    if (rc != NCDE_OK) return rc;
    if (!out || !workspace || !record) return fail(NCDE_ERR_INVALID, "out / workspace / record is NULL");
    char msg[256] = "";
    const bool v4 = p->abi_version >= 4;
    rc = ncde_dp_solve(&q_, ts, opt, 0, out, nullptr, nullptr, nullptr, workspace, workspace_bytes, (hipStream_t)stream, stats, msg, sizeof(msg),
                       record, record_bytes, v4 ? opt->replay : nullptr, v4 ? opt->replay_count : 0);
    if (rc != NCDE_OK) return fail(rc, "%s", msg);
    return NCDE_OK;
}

int ncde_dopri5_backward(const NcdeProblem* p, const NcdeTimeSpec* ts, const NcdeAdaptiveOptions* opt, const void* record,
                         size_t record_bytes, const float* grad_out, const NcdeGrads* grads, void* workspace, size_t workspace_bytes,
                         void* stream) {
    NcdeProblem q_;
    if (!opt) return fail(NCDE_ERR_INVALID, "options are NULL");
    int rc = dopri5_prepare(p, ts, opt, 1, &q_);
    if (rc != NCDE_OK) return rc;
    if (!record || !grad_out || !grads || !grads->grad_z0 || !workspace) return fail(NCDE_ERR_INVALID, "NULL record/grad_out/grads/workspace");
    char msg[256] = "";
    rc = ncde_dp_tape_backward_run(&q_, ts, opt, record, record_bytes, grad_out, grads, workspace, workspace_bytes, (hipStream_t)stream, msg, sizeof(msg));
    if (rc != NCDE_OK) return fail(rc, "%s", msg);
    return NCDE_OK;
}

int ncde_dopri5_forward(const NcdeProblem* p, const NcdeTimeSpec* ts, const NcdeAdaptiveOptions* opt, float* out, void* workspace,
                        size_t workspace_bytes, void* stream, NcdeAdaptiveStats* stats) {
    NcdeProblem q_;
    if (!opt) return fail(NCDE_ERR_INVALID, "options are NULL");
    int rc = dopri5_prepare(p, ts, opt, 0, &q_);
    if (rc != NCDE_OK) return rc;
    if (!out || !workspace) return fail(NCDE_ERR_INVALID, "out / workspace is NULL");
    char msg[256] = "";
    const bool v4 = p->abi_version >= 4;      // the replay members of the options exist from ABI version 4 on
    rc = ncde_dp_solve(&q_, ts, opt, 0, out, nullptr, nullptr, nullptr, workspace, workspace_bytes, (hipStream_t)stream, stats, msg, sizeof(msg),
                       nullptr, 0, v4 ? opt->replay : nullptr, v4 ? opt->replay_count : 0);
    if (rc != NCDE_OK) return fail(rc, "%s", msg);
    return NCDE_OK;
}

int ncde_dopri5_adjoint(const NcdeProblem* p, const NcdeTimeSpec* ts, const NcdeAdaptiveOptions* opt, const float* z_out,
                        const float* grad_out, const NcdeGrads* grads, void* workspace, size_t workspace_bytes, void* stream,
                        NcdeAdaptiveStats* stats) {
    NcdeProblem q_;
    if (!opt) return fail(NCDE_ERR_INVALID, "options are NULL");
    int rc = dopri5_prepare(p, ts, opt, 1, &q_);
    if (rc != NCDE_OK) return rc;
    if (!z_out || !grad_out || !grads || !grads->grad_z0 || !workspace) return fail(NCDE_ERR_INVALID, "NULL z_out/grad_out/grads/workspace");
    char msg[256] = "";
    const bool v4 = p->abi_version >= 4;
    rc = ncde_dp_solve(&q_, ts, opt, 1, nullptr, z_out, grad_out, grads, workspace, workspace_bytes, (hipStream_t)stream, stats, msg, sizeof(msg),
                       nullptr, 0, v4 ? opt->replay : nullptr, v4 ? opt->replay_count : 0);
    if (rc != NCDE_OK) return fail(rc, "%s", msg);
    return NCDE_OK;
}

int ncde_time_plan_build(const NcdeProblem* p, const NcdeTimeSpec* ts, void* host_buffer, size_t bytes, NcdeTimePlanInfo* info) {
    NcdeProblem q_;
    int rc = normalize(p, &q_);
    if (rc != NCDE_OK) return rc;
    if (q_.n_knots < 2) return fail(NCDE_ERR_INVALID, "Must have a time dimension of size at least 2 (n_knots=%d)", q_.n_knots);
    if (q_.method != NCDE_EULER && q_.method != NCDE_MIDPOINT && q_.method != NCDE_RK4_38) return fail(NCDE_ERR_INVALID, "Invalid method %d", q_.method);
    char msg[256] = "";
    rc = ncde_time_plan_build_impl(&q_, ts, host_buffer, bytes, info, msg, sizeof(msg));
    if (rc != NCDE_OK) return fail(rc, "%s", msg);
    return NCDE_OK;
}

int64_t ncde_workspace_bytes(const NcdeProblem* p, int pass) {
    NcdeProblem q_;
    int rc = normalize(p, &q_);
    if (rc != NCDE_OK) return rc;
    p = &q_;
    rc = validate(p);
    if (rc != NCDE_OK) return rc;
    const Layout y = make_layout(p);
    const int fam = select_family(p, y, pass);
    if (fam < 0) return fam;
    if (fam == 1) return ncde_fast_workspace_bytes(p, pass);
    if (fam == 2) return ncde_tiled_workspace_bytes(p, pass);
    if (fam == 3) return ncde_variant_workspace_bytes(p, pass);
    if (pass == 0) return 256;
    return (int64_t)sizeof(float) * (int64_t)y.n_wg * (int64_t)y.theta_size + 256;
}

int64_t ncde_stage_record_bytes(const NcdeProblem* p) {
    NcdeProblem q_;
    int rc = normalize(p, &q_);
    if (rc != NCDE_OK) return rc;
    p = &q_;
    rc = validate(p);
    if (rc != NCDE_OK) return rc;
    const int S = p->method == NCDE_RK4_38 ? 4 : (p->method == NCDE_MIDPOINT ? 2 : 1);
    const int n_steps = p->output == NCDE_OUT_TIMES ? p->n_steps_fwd : p->n_knots - 1;
    return (int64_t)sizeof(float) * (int64_t)n_steps * S * (int64_t)p->batch * p->hidden;
}

const char* ncde_kernel_name(const NcdeProblem* p, int pass) {
    NcdeProblem q_;
    if (normalize(p, &q_) != NCDE_OK) return nullptr;
    p = &q_;
    if (validate(p) != NCDE_OK) return nullptr;
    const Layout y = make_layout(p);
    const int fam = select_family(p, y, pass);
    if (fam < 0) return nullptr;
    if (fam == 2) return ncde_tiled_kernel_name(p, pass);
    if (fam == 3) return pass == 0 ? "ncde_fwd_variant" : (pass == 1 ? "ncde_adj_variant" : "ncde_adj_variant<discrete>");
    return fam == 1 ? ncde_fast_kernel_name(p, pass) : (pass == 0 ? "ncde_fwd_generic" : (pass == 1 ? "ncde_adj_generic" : "ncde_adj_generic<discrete>"));
}

int ncde_forward(const NcdeProblem* p, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    NcdeProblem q_;
    int rc = normalize(p, &q_);
    if (rc != NCDE_OK) return rc;
    p = &q_;
    rc = validate(p);
    if (rc != NCDE_OK) return rc;
    if (!out) return fail(NCDE_ERR_INVALID, "out is NULL");
    const Layout y = make_layout(p);
    const int fam = select_family(p, y, 0);
    if (fam < 0) return fam;
    const int64_t need = ncde_workspace_bytes(p, 0);
    if (need > 0 && (!workspace || (int64_t)workspace_bytes < need)) return fail(NCDE_ERR_WORKSPACE, "workspace %zu B < %lld B", workspace_bytes, (long long)need);
    return launch_forward(p, y, fam, out, nullptr, workspace, workspace_bytes, (hipStream_t)stream);
}

int ncde_forward_record(const NcdeProblem* p, float* out, float* stages, void* workspace, size_t workspace_bytes, void* stream) {
    NcdeProblem q_;
    int rc = normalize(p, &q_);
    if (rc != NCDE_OK) return rc;
    p = &q_;
    rc = validate(p);
    if (rc != NCDE_OK) return rc;
    if (!out || !stages) return fail(NCDE_ERR_INVALID, "out/stages is NULL");
    if (p->flags & NCDE_FLAG_DEBUG_PROFILE) return fail(NCDE_ERR_UNSUPPORTED, "no instrumented recording forward");
    const Layout y = make_layout(p);
    const int fam = select_family(p, y, 0);
    if (fam < 0) return fam;
    const int64_t need = ncde_workspace_bytes(p, 0);
    if (need > 0 && (!workspace || (int64_t)workspace_bytes < need)) return fail(NCDE_ERR_WORKSPACE, "workspace %zu B < %lld B", workspace_bytes, (long long)need);
    return launch_forward(p, y, fam, out, stages, workspace, workspace_bytes, (hipStream_t)stream);
}

int ncde_backward(const NcdeProblem* p, const float* stages, const float* grad_out, const NcdeGrads* grads, void* workspace,
                  size_t workspace_bytes, void* stream) {
    NcdeProblem q_;
    int rc = normalize(p, &q_);
    if (rc != NCDE_OK) return rc;
    p = &q_;
    rc = validate(p);
    if (rc != NCDE_OK) return rc;
    if (!stages || !grad_out || !grads || !grads->grad_z0) return fail(NCDE_ERR_INVALID, "NULL stages/grad_out/grads");
    const Layout y = make_layout(p);
    const int fam = select_family(p, y, 2);
    if (fam < 0) return fam;
    const int64_t need = ncde_workspace_bytes(p, 2);
    if (!workspace || (int64_t)workspace_bytes < need) return fail(NCDE_ERR_WORKSPACE, "workspace %zu B < %lld B", workspace_bytes, (long long)need);
    return launch_adjoint(p, y, fam, stages, grad_out, grads, workspace, workspace_bytes, (hipStream_t)stream, false, true);
}

int ncde_adjoint(const NcdeProblem* p, const float* z_out, const float* grad_out, const NcdeGrads* grads, void* workspace,
                 size_t workspace_bytes, void* stream) {
    NcdeProblem q_;
    int rc = normalize(p, &q_);
    if (rc != NCDE_OK) return rc;
    p = &q_;
    rc = validate(p);
    if (rc != NCDE_OK) return rc;
    if (!z_out || !grad_out || !grads || !grads->grad_z0) return fail(NCDE_ERR_INVALID, "NULL z_out/grad_out/grads");
    const Layout y = make_layout(p);
    const int fam = select_family(p, y, 1);
    if (fam < 0) return fam;
    const int64_t need = ncde_workspace_bytes(p, 1);
    if (!workspace || (int64_t)workspace_bytes < need) return fail(NCDE_ERR_WORKSPACE, "workspace %zu B < %lld B", workspace_bytes, (long long)need);
    return launch_adjoint(p, y, fam, z_out, grad_out, grads, workspace, workspace_bytes, (hipStream_t)stream, false, false);
}

int ncde_time_kernel(const NcdeProblem* p, int pass, float* out, const float* grad_out, const NcdeGrads* grads, void* workspace,
                     size_t workspace_bytes, void* stream, int iters, float* ms_per_launch) {
    NcdeProblem q_;
    int rc = normalize(p, &q_);
    if (rc != NCDE_OK) return rc;
    p = &q_;
    rc = validate(p);
    if (rc != NCDE_OK) return rc;
    if (iters < 1 || !ms_per_launch) return fail(NCDE_ERR_INVALID, "iters < 1 or NULL result");
    if (pass < 0 || pass > 2) return fail(NCDE_ERR_INVALID, "pass %d outside {0, 1, 2}", pass);
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
        if (pass == 0) rc = launch_forward(p, y, fam, out, nullptr, workspace, workspace_bytes, st);
        else rc = launch_adjoint(p, y, fam, out, grad_out, grads, workspace, workspace_bytes, st, true, pass == 2);
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
