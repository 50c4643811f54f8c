// Host side of the C-ABI (include/ncde_hip.h): validation, kernel selection, launches.
// No torch types, no allocation, no synchronisation on the hot path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstring>

#include "ncde_adaptive.h"
#include "ncde_adaptive_fast.h"
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
    // reserved_ is NOT an input: whatever the caller left there (an uninitialised stack struct, say) is dropped.  Internally the
    // field carries the real extents of a zero-padded problem, and only make_pad_plan() -- after this point -- ever sets it.
    out->reserved_ = 0;
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


// ---- zero-padding into the batch-tiled family ----------------------------------------------------------------------------------
// The batch-tiled kernels want H and the layer widths as multiples of 16 (the last one 16 / 32 / 64 / 128 for the backward) and C as a
// multiple of 4.  Any other shape -- the reference's default hidden_hidden_dim = 15 (src/ncde/ncde.py:47), odd channel counts,
// its hyper-parameter ranges (experiments/configurations/configurations.json5:34-36) -- is padded HERE: the parameters are copied
// once per call into zero-padded buffers at the head of the workspace (a few hundred KB), the kernels run on the padded problem, and
// the parameter gradients are sliced back.  The caller's activations are NOT copied: the kernels take the real row width of
// z0 / out / z_out / grad_out / grad_z0 / the stage record and the real channel count of the coefficient tensor (KArgs.Hr, Cc).
// Exactness: a padded hidden unit has zero weights and bias, so relu(0) = 0 feeds zeros forward and relu'(0) = 0 masks its
// cotangent; a padded state row has zero rows in Wo / bo, so tanh(0) * dX = 0 leaves it at 0; a padded channel has dX = 0.
struct PadSegs {
    int n;
    const float* src[12];
    float* dst[12];
    int d[12][6];      // real extents n0, n1, n2 and padded extents p0, p1, p2 (row-major, last index fastest)
};
// dir = 0: dst (padded) <- src (real), zeros elsewhere;  dir = 1: dst (real) <- src (padded)
__global__ __launch_bounds__(256) void ncde_pad_params(PadSegs sg, int dir) {
    const int k = blockIdx.y;
    const int n0 = sg.d[k][0], n1 = sg.d[k][1], n2 = sg.d[k][2], p1 = sg.d[k][4], p2 = sg.d[k][5];
    const long long total = dir == 0 ? (long long)sg.d[k][3] * p1 * p2 : (long long)n0 * n1 * n2;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        if (dir == 0) {
            const int i2 = (int)(e % p2), i1 = (int)((e / p2) % p1), i0 = (int)(e / ((long long)p1 * p2));
            sg.dst[k][e] = (i0 < n0 && i1 < n1 && i2 < n2) ? sg.src[k][((long long)i0 * n1 + i1) * n2 + i2] : 0.0f;
        } else {
            const int i2 = (int)(e % n2), i1 = (int)((e / n2) % n1), i0 = (int)(e / ((long long)n1 * n2));
            sg.dst[k][e] = sg.src[k][((long long)i0 * p1 + i1) * p2 + i2];
        }
    }
}

// (the backward's output phase is instantiated for last widths 16 .. 256 in powers of two; the forward takes any multiple of 16 above 128)
inline int pad_width(int w, int pass) {
    return w <= 16 ? 16 : (w <= 32 ? 32 : (w <= 64 ? 64 : (w <= 128 ? 128 : ((pass != 0 && w <= 256) ? 256 : hru16(w)))));
}

struct PadPlan {
    bool ok;
    int inner;                           // family the padded problem runs on: 2 = batch-tiled, 1 = shape-specialised (register-resident)
    NcdeProblem q;                       // the padded problem; its parameter pointers are OFFSETS (floats, +1) until bound to a workspace
    int n_seg;
    int dims[12][6];
    const float* real[12];               // the caller's parameter of segment k
    long long off[12];                   // float offset of its padded copy
    int slot_W[NCDE_MAX_LAYERS], slot_b[NCDE_MAX_LAYERS], slot_Wo, slot_bo, slot_Wg, slot_bg;
    long long param_floats;              // padded parameters (and, for the backward, the same again for their gradients)
};

// Would padding bring `p` into the batch-tiled family (target 0: original / minimal-gated field, matmul input), or onto one of the
// shape-specialised kernel sets (target 1: (H, HH, C) = (32, 32, 20), target 2: (64, 64, 4), target 3: (32, 32, 40) forward only; original field, default time axis, ONE
// shared inner layer as the reference's fields have)?  Measured at B = 4096, T = 99 (profiles/r04_shape_sweep_perf.txt): the
// specialised kernels at their full padded size take no longer than the SMALLEST batch-tiled shapes (forward 0.6 - 0.9 ms vs 0.7 ms,
// adjoint 1.8 - 2.0 ms vs 2.1 ms at (4, 32, 16)), so a shape they can hold is always sent there.
PadPlan make_pad_plan(const NcdeProblem* p, int pass, int target = 0) {
    PadPlan P{};
    P.ok = false;
    P.inner = target == 0 ? 2 : 1;
    if (p->n_layers < 1 || p->field_input != NCDE_INPUT_MATMUL || p->field_kind == NCDE_FIELD_GRU) return P;
    if (p->hidden > 2048 || p->channels > 4095) return P;
    // (target 3, round 5: (32, 32, 40) -- a forward-only kernel set: pass 0 on the default time axis; targets 4 - 6, round 6: (32, 32, 4 / 8 /
    // 12) -- few channels, every pass, default time axis)
    const int tH = target == 2 ? 64 : 32, tHH = tH, tC = target == 1 ? 20 : (target == 2 ? 4 : (target == 3 ? 40 : 4 * (target - 3)));
    if (target == 3 && (pass != 0 || p->output == NCDE_OUT_TIMES)) return P;
    if (target >= 4 && p->output == NCDE_OUT_TIMES) return P;
    if (target != 0) {
        if (p->field_kind != NCDE_FIELD_ORIGINAL || p->hidden > tH || p->channels > tC) return P;
        // (any batch-tiled knob is a request for that family; the non-default adjoint variants exist for the exact shapes only)
        if (p->flags & (NCDE_FLAG_ADJOINT_V1 | NCDE_FLAG_ADJOINT_V2 | NCDE_FLAG_ADJOINT_V4 | NCDE_FLAG_DEBUG_PROFILE | NCDE_FLAG_FORCE_TILED |
                        NCDE_FLAG_TILED_NS1 | NCDE_FLAG_TILED_NS2 | NCDE_FLAG_TILED_NS4 | 0x00FF0000u | 0x200u)) return P;
        for (int l = 0; l < p->n_layers; ++l) {
            if (p->layer_out[l] > tHH) return P;
            if (l >= 1 && (p->layer_W[l] != p->layer_W[1] || p->layer_b[l] != p->layer_b[1])) return P;
        }
        if (p->n_layers > 1 && (p->layer_W[1] == p->layer_W[0] || p->layer_b[1] == p->layer_b[0])) return P;
    }
    NcdeProblem& q = P.q;
    q = *p;
    q.hidden = target == 0 ? hru16(p->hidden) : tH;
    q.channels = target == 0 ? hru4(p->channels) : tC;
    q.reserved_ = (p->hidden << 12) | p->channels;
    int n = 0;
    long long off = 0;
    auto add = [&](const float* src, int n0, int n1, int n2, int p0, int p1, int p2) {
        for (int k = 0; k < n; ++k)
            if (P.real[k] == src) return k;      // a shared layer: one padded copy
        const int k = n++;
        P.real[k] = src;
        P.dims[k][0] = n0; P.dims[k][1] = n1; P.dims[k][2] = n2; P.dims[k][3] = p0; P.dims[k][4] = p1; P.dims[k][5] = p2;
        P.off[k] = off;
        off += ((long long)p0 * p1 * p2 + 3) & ~3LL;
        return k;
    };
    int din = q.hidden;
    for (int l = 0; l < p->n_layers; ++l) {
        const int dout = target == 0 ? pad_width(p->layer_out[l], pass) : tHH;
        q.layer_in[l] = din; q.layer_out[l] = dout;
        P.slot_W[l] = add(p->layer_W[l], 1, p->layer_out[l], p->layer_in[l], 1, dout, din);
        P.slot_b[l] = add(p->layer_b[l], 1, 1, p->layer_out[l], 1, 1, dout);
        // the same real matrix must get the same padded shape in every slot it is used in
        if (P.dims[P.slot_W[l]][4] != dout || P.dims[P.slot_W[l]][5] != din || P.dims[P.slot_b[l]][5] != dout) return P;
        din = dout;
    }
    const int dl = p->layer_out[p->n_layers - 1];
    P.slot_Wo = add(p->Wo, p->hidden, p->channels, dl, q.hidden, q.channels, din);
    P.slot_bo = add(p->bo, 1, p->hidden, p->channels, 1, q.hidden, q.channels);
    if (p->field_kind == NCDE_FIELD_MINIMAL) {
        P.slot_Wg = add(p->Wg, p->hidden, p->channels, dl, q.hidden, q.channels, din);
        P.slot_bg = add(p->bg, 1, p->hidden, p->channels, 1, q.hidden, q.channels);
    }
    if (n > 12) return P;
    P.n_seg = n;
    P.param_floats = (off + 63) & ~63LL;
    // bind to a fake base so that the support checks see aligned, distinct pointers (offset + 64 floats: never NULL)
    auto fake = [&](int k) { return reinterpret_cast<const float*>((uintptr_t)(P.off[k] + 64) * sizeof(float)); };
    for (int l = 0; l < p->n_layers; ++l) { q.layer_W[l] = fake(P.slot_W[l]); q.layer_b[l] = fake(P.slot_b[l]); }
    q.Wo = fake(P.slot_Wo); q.bo = fake(P.slot_bo);
    if (p->field_kind == NCDE_FIELD_MINIMAL) { q.Wg = fake(P.slot_Wg); q.bg = fake(P.slot_bg); }
    P.ok = target == 0 ? ncde_tiled_supported(&q, pass) : ncde_fast_supported(&q, pass);
    return P;
}
// the padded plan a problem takes, if any: a shape-specialised kernel set first, then the batch-tiled family
PadPlan pick_pad_plan(const NcdeProblem* p, int pass, bool allow_tiled) {
    for (int target : {4, 5, 6, 1, 2, 3}) {      // (the smallest channel set that holds the problem first)
        PadPlan P = make_pad_plan(p, pass, target);
        if (P.ok) return P;
    }
    if (allow_tiled && !ncde_tiled_supported(p, pass)) return make_pad_plan(p, pass, 0);
    PadPlan none{};
    return none;
}
// head of the workspace of a padded call: [padded parameters | (backward) padded parameter gradients | the tiled family's own workspace]
long long pad_head_floats(const PadPlan& P, int pass) { return (pass == 0 ? 1 : 2) * P.param_floats + 64; }

void pad_bind(PadPlan& P, const NcdeProblem* p, float* base) {
    NcdeProblem& q = P.q;
    for (int l = 0; l < p->n_layers; ++l) { q.layer_W[l] = base + P.off[P.slot_W[l]]; q.layer_b[l] = base + P.off[P.slot_b[l]]; }
    q.Wo = base + P.off[P.slot_Wo]; q.bo = base + P.off[P.slot_bo];
    if (p->field_kind == NCDE_FIELD_MINIMAL) { q.Wg = base + P.off[P.slot_Wg]; q.bg = base + P.off[P.slot_bg]; }
}
int pad_launch(const PadPlan& P, float* base, int dir, float* const* real_dst, hipStream_t st) {
    PadSegs sg{};
    sg.n = P.n_seg;
    long long biggest = 1;
    for (int k = 0; k < P.n_seg; ++k) {
        for (int i = 0; i < 6; ++i) sg.d[k][i] = P.dims[k][i];
        if (dir == 0) { sg.src[k] = P.real[k]; sg.dst[k] = base + P.off[k]; }
        else { sg.src[k] = base + P.off[k]; sg.dst[k] = real_dst[k]; if (!real_dst[k]) return NCDE_ERR_INVALID; }
        biggest = std::max(biggest, (long long)P.dims[k][3] * P.dims[k][4] * P.dims[k][5]);
    }
    const int gx = (int)std::min<long long>((biggest + 255) / 256, 512);
    hipLaunchKernelGGL(ncde_pad_params, dim3(gx, P.n_seg), dim3(256), 0, st, sg, dir);
    return hipGetLastError() == hipSuccess ? NCDE_OK : NCDE_ERR_HIP;
}

// pick the kernel family: 1 = fast (shape-specialised), 2 = tiled (batch-tiled, large hidden), 4 = tiled on the zero-padded problem,
// 3 = variant, 0 = generic, <0 = error
int select_family_unpadded(const NcdeProblem* p, const Layout& y, int pass);
int select_family(const NcdeProblem* p, const Layout& y, int pass) {
    const int fam = select_family_unpadded(p, y, pass);
    if (p->flags & (NCDE_FLAG_FORCE_GENERIC | NCDE_FLAG_FORCE_FAST)) return fam;
    // no specialised kernel of its own: a zero-padded run on a specialised kernel set (also instead of the batch-tiled family, which
    // the smallest shapes do not use well), else -- instead of the generic / variant kernels -- on the batch-tiled family
    if (fam == 0 || fam == 2 || fam == 3 || fam == NCDE_ERR_UNSUPPORTED) {
        const bool fallback = fam != 2;
        if (pick_pad_plan(p, pass, fallback).ok) return 4;
    }
    return fam;
}
int select_family_unpadded(const NcdeProblem* p, const Layout& y, int pass) {
    if (p->output == NCDE_OUT_TIMES) {   // general time axis: the plan-driven kernels -- batch-tiled where the shape allows
                                         // (multiples of 16 / 4; 2.8x the generic family at cfg2 widths, 10x at cfg5's), else generic / variant
        // (round 4: the shape-specialised kernels walk the plan too -- forward of both shapes, continuous adjoint of (32, 32, 20) nl = 3
        // and of H = 64 / C <= 4; the planned discrete backward stays on the batch-tiled family)
        const bool fast_ok = !y.variant && !(p->flags & (NCDE_FLAG_FORCE_GENERIC | NCDE_FLAG_FORCE_TILED)) && ncde_fast_supported(p, pass);
        if (p->flags & NCDE_FLAG_FORCE_FAST) {
            if (!fast_ok) return fail(NCDE_ERR_UNSUPPORTED, "no shape-specialised kernel for this problem on a general time axis (pass %d)", pass);
            return 1;
        }
        if (fast_ok) return 1;
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
    if (family == 4) {
        PadPlan P = pick_pad_plan(p, 0, true);
        if (!P.ok) return fail(NCDE_ERR_UNSUPPORTED, "padded problem outside the aligned kernel families");
        float* base = (float*)ws;
        const long long head = pad_head_floats(P, 0);
        pad_bind(P, p, base);
        if (pad_launch(P, base, 0, nullptr, st) != NCDE_OK) return fail(NCDE_ERR_HIP, "parameter padding launch failed");
        const size_t inner_bytes = ws_bytes - sizeof(float) * (size_t)head;
        const int rc = P.inner == 1 ? ncde_fast_forward(&P.q, out, stages, base + head, inner_bytes, st)
                                    : ncde_tiled_forward(&P.q, out, stages, base + head, inner_bytes, st);
        if (rc != NCDE_OK) return fail(rc, "forward (zero-padded problem) launch failed");
        return NCDE_OK;
    }
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
    if (family == 4) {
        PadPlan P = pick_pad_plan(p, discrete ? 2 : 1, true);
        if (!P.ok) return fail(NCDE_ERR_UNSUPPORTED, "padded problem outside the aligned kernel families");
        float* base = (float*)ws;
        const long long head = pad_head_floats(P, 1);
        pad_bind(P, p, base);
        if (pad_launch(P, base, 0, nullptr, st) != NCDE_OK) return fail(NCDE_ERR_HIP, "parameter padding launch failed");
        float* gbase = base + P.param_floats;      // padded gradients, same layout as the padded parameters
        NcdeGrads gq{};
        gq.grad_z0 = g->grad_z0;                   // real row width: written in place
        float* real_dst[12] = {nullptr};
        for (int l = 0; l < p->n_layers; ++l) {
            gq.grad_layer_W[l] = gbase + P.off[P.slot_W[l]]; gq.grad_layer_b[l] = gbase + P.off[P.slot_b[l]];
            real_dst[P.slot_W[l]] = g->grad_layer_W[l]; real_dst[P.slot_b[l]] = g->grad_layer_b[l];
        }
        gq.grad_Wo = gbase + P.off[P.slot_Wo]; gq.grad_bo = gbase + P.off[P.slot_bo];
        real_dst[P.slot_Wo] = g->grad_Wo; real_dst[P.slot_bo] = g->grad_bo;
        if (p->field_kind == NCDE_FIELD_MINIMAL) {
            gq.grad_Wg = gbase + P.off[P.slot_Wg]; gq.grad_bg = gbase + P.off[P.slot_bg];
            real_dst[P.slot_Wg] = g->grad_Wg; real_dst[P.slot_bg] = g->grad_bg;
        }
        const size_t inner_bytes = ws_bytes - sizeof(float) * (size_t)head;
        const int rc = P.inner == 1 ? ncde_fast_adjoint(&P.q, src, grad_out, &gq, base + head, inner_bytes, st, main_kernel_only, discrete)
                                    : ncde_tiled_adjoint(&P.q, src, grad_out, &gq, base + head, inner_bytes, st, main_kernel_only, discrete);
        if (rc != NCDE_OK) return fail(rc, "adjoint (zero-padded problem) launch failed");
        if (main_kernel_only) return NCDE_OK;
        const int rc2 = pad_launch(P, gbase, 1, real_dst, st);
        if (rc2 == NCDE_ERR_INVALID) return fail(rc2, "NcdeGrads: NULL destination for a parameter gradient");
        if (rc2 != NCDE_OK) return fail(rc2, "gradient un-padding launch failed");
        return NCDE_OK;
    }
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
    if (!ncde_dp_supported(q, adj == 2 ? 0 : adj, why, sizeof(why))) return fail(NCDE_ERR_UNSUPPORTED, "%s", why);
    if (adj == 2 && (!ncde_dp_supported(q, 1, why, sizeof(why)) || !ncde_dp_tape_supported(q, why, sizeof(why)))) return fail(NCDE_ERR_UNSUPPORTED, "%s", why);
    return NCDE_OK;
}
}  // namespace

int64_t ncde_dopri5_workspace_bytes(const NcdeProblem* p, const NcdeTimeSpec* ts, int pass) {
    NcdeProblem q_;
    NcdeAdaptiveOptions o{};
    o.rtol = 1e-4;
    const int rc = dopri5_prepare(p, ts, &o, pass == 2 ? 2 : (pass != 0), &q_);
    if (rc != NCDE_OK) return rc;
    return ncde_dp_workspace_bytes(&q_, ts->n_t, pass);
}

const char* ncde_dopri5_kernel_name(const NcdeProblem* p, int pass) {
    NcdeProblem q_;
    NcdeAdaptiveOptions o{};
    o.rtol = 1e-4;
    const double t2[2] = {0.0, 1.0};
    NcdeTimeSpec ts{};
    ts.n_t = 2;
    ts.t = t2;
    if (dopri5_prepare(p, &ts, &o, pass == 2 ? 2 : (pass != 0), &q_) != NCDE_OK) return nullptr;
    if (pass == 2) return ncde_dpf_tape_supported(&q_) ? ncde_dpf_tape_kernel_name(&q_) : "ncde_dp_tape_backward";
    const char* f = ncde_dpf_kernel_name(&q_, pass != 0);
    if (f) return f;
    return pass ? "ncde_dp_stage x 6 + ncde_dp_reduce_theta x 6 + ncde_dp_control + ncde_dp_commit" : "ncde_dp_stage x 6 + ncde_dp_control + ncde_dp_commit";
}

int64_t ncde_dopri5_record_bytes(const NcdeProblem* p, const NcdeTimeSpec* ts, const NcdeAdaptiveOptions* opt) {
    NcdeProblem q_;
    if (!opt) return fail(NCDE_ERR_INVALID, "options are NULL");
    const int rc = dopri5_prepare(p, ts, opt, 2, &q_);      // 2: a taped solve -- also the limits of its reverse sweep
    if (rc != NCDE_OK) return rc;
    return ncde_dp_record_bytes(&q_, ts, opt);
}

int ncde_dopri5_forward_record(const NcdeProblem* p, const NcdeTimeSpec* ts, const NcdeAdaptiveOptions* opt, float* out, void* record,
                               size_t record_bytes, void* workspace, size_t workspace_bytes, void* stream, NcdeAdaptiveStats* stats) {
    NcdeProblem q_;
    if (!opt) return fail(NCDE_ERR_INVALID, "options are NULL");
    int rc = dopri5_prepare(p, ts, opt, 2, &q_);
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
    int rc = dopri5_prepare(p, ts, opt, 2, &q_);
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
    if (fam == 4) {
        const PadPlan P = pick_pad_plan(p, pass, true);
        const int64_t inner = P.inner == 1 ? ncde_fast_workspace_bytes(&P.q, pass) : ncde_tiled_workspace_bytes(&P.q, pass);
        return inner < 0 ? inner : inner + (int64_t)sizeof(float) * pad_head_floats(P, pass);
    }
    if (fam == 2) return ncde_tiled_workspace_bytes(p, pass);
    if (fam == 3) return ncde_variant_workspace_bytes(p, pass);
    if (pass == 0) return 256;
    return (int64_t)sizeof(float) * (int64_t)y.n_wg * (int64_t)y.theta_size + 256;
}

int64_t ncde_coop_status_offset(const NcdeProblem* p, int pass) {
    NcdeProblem q_;
    int rc = normalize(p, &q_);
    if (rc != NCDE_OK) return rc;
    p = &q_;
    rc = validate(p);
    if (rc != NCDE_OK) return rc;
    if (pass < 0 || pass > 2) return fail(NCDE_ERR_INVALID, "pass %d outside {0, 1, 2}", pass);
    const Layout y = make_layout(p);
    const int fam = select_family(p, y, pass);
    if (fam < 0) return fam;
    int64_t off = -1;
    if (fam == 2) off = ncde_tiled_status_offset(p, pass);
    else if (fam == 4) {
        const PadPlan P = pick_pad_plan(p, pass, true);
        if (P.inner == 2) {
            off = ncde_tiled_status_offset(&P.q, pass);
            if (off >= 0) off += (int64_t)sizeof(float) * pad_head_floats(P, pass);
        }
    }
    if (off < 0) return fail(NCDE_ERR_UNSUPPORTED, "this problem / pass launches no cooperative kernel");
    return off;
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
    if (fam == 4) {      // (on the zero-padded problem)
        const PadPlan P = pick_pad_plan(p, pass, true);
        return P.inner == 1 ? ncde_fast_kernel_name(&P.q, pass) : ncde_tiled_kernel_name(&P.q, pass);
    }
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
