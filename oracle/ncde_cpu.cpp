// ORACLE (test infrastructure, NOT product code): scalar C++ restatement of the hot path behind the SAME C-ABI as the HIP
// library (include/ncde_hip.h), with HOST pointers -- SURVEY.md §8(b): "the same ABI is implemented by the CPU restatement so
// tests and the CPU baseline run through identical plumbing".  Built into oracle/_build/libncde_cpu.so by oracle/Makefile;
// loaded only by tests/ and by bench.py's cpu_baseline leg.  The product package never loads it (there is no CPU fallback).
//
// What it covers: the original vector field with the matmul input on the default integer knot grid with step 1 -- forward
// (ncde_forward), continuous adjoint (ncde_adjoint), recording forward + exact discrete backward (ncde_forward_record,
// ncde_backward) -- for euler / midpoint / rk4 (3/8 rule), linear and cubic control paths, final-time or every-knot outputs.
// One OpenMP thread per sample (samples never interact), plain fp32 loops in the reference's operation order for the
// time-stepping arithmetic; dot products are SIMD reductions (torch's addmm order is BLAS-dependent anyway, so this restatement
// is pinned to the reference through the golden fixtures at the same tolerances as the oracle's hand VJPs, not bit for bit).
// Reference lines restated (relative to /root/reference):
//   knot index / dX/dt          modules/torchcde/torchcde/interpolation_linear.py:212-234, interpolation_cubic.py:315-336
//   f_theta, contraction        src/ncde/vector_fields/base.py:64-104, modules/torchcde/torchcde/solver.py:112-137
//   fixed-grid loop, tableaux   modules/torchdiffeq/torchdiffeq/_impl/solvers.py:94-119, fixed_grid.py:6-29, rk_common.py:106-114
//   adjoint sweep               modules/torchdiffeq/torchdiffeq/_impl/adjoint.py:37-145, misc.py:152-159
#include <omp.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "ncde_hip.h"

namespace {

thread_local char g_err[256] = "";
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

struct Dims {
    int B, T, C, H, L, S, n_out, n_pieces;
    int din[NCDE_MAX_LAYERS], dout[NCDE_MAX_LAYERS];
    int slot[NCDE_MAX_LAYERS];       // unique-parameter slot of each layer (shared layers share a slot)
    int n_slots, slot_layer[NCDE_MAX_LAYERS];
    int dlast;
};

int setup(const NcdeProblem* p, Dims* d) {
    if (!p) return fail(NCDE_ERR_INVALID, "problem is NULL");
    if (p->abi_version < 1 || p->abi_version > NCDE_ABI_VERSION) return fail(NCDE_ERR_INVALID, "abi_version %d", p->abi_version);
    if (p->abi_version >= 2 && (p->field_kind != NCDE_FIELD_ORIGINAL || p->field_input != NCDE_INPUT_MATMUL))
        return fail(NCDE_ERR_UNSUPPORTED, "CPU restatement: original field, matmul input only");
    if (p->abi_version >= 3 && p->output == NCDE_OUT_TIMES) return fail(NCDE_ERR_UNSUPPORTED, "CPU restatement: default time axis only");
    if (p->batch < 1 || p->channels < 1 || p->hidden < 1 || p->n_knots < 2 || p->n_layers < 0 || p->n_layers > NCDE_MAX_LAYERS)
        return fail(NCDE_ERR_INVALID, "bad dimensions");
    if (p->method != NCDE_EULER && p->method != NCDE_MIDPOINT && p->method != NCDE_RK4_38) return fail(NCDE_ERR_INVALID, "Invalid method %d", p->method);
    d->B = p->batch; d->T = p->n_knots; d->C = p->channels; d->H = p->hidden; d->L = p->n_layers;
    d->S = p->method == NCDE_RK4_38 ? 4 : (p->method == NCDE_MIDPOINT ? 2 : 1);
    d->n_out = p->output == NCDE_OUT_KNOTS ? p->n_knots : 2;
    d->n_pieces = p->n_knots - 1;
    d->n_slots = 0;
    int in = p->hidden;
    for (int l = 0; l < p->n_layers; ++l) {
        if (p->layer_in[l] != in) return fail(NCDE_ERR_INVALID, "layer %d: in=%d does not chain from %d", l, p->layer_in[l], in);
        d->din[l] = p->layer_in[l]; d->dout[l] = p->layer_out[l];
        in = p->layer_out[l];
        d->slot[l] = -1;
        for (int q = 0; q < l; ++q)
            if (p->layer_W[q] == p->layer_W[l]) d->slot[l] = d->slot[q];
        if (d->slot[l] < 0) { d->slot[l] = d->n_slots; d->slot_layer[d->n_slots++] = l; }
    }
    d->dlast = in;
    return NCDE_OK;
}

float stage_offset(int method, int j) {
    if (method == NCDE_RK4_38) return j == 0 ? 0.0f : (j == 1 ? 0.333333343267440796f : (j == 2 ? 0.666666686534881592f : 1.0f));
    if (method == NCDE_MIDPOINT) return j == 0 ? 0.0f : 0.5f;
    return 0.0f;
}
float stage_weight(int method, int j) {
    if (method == NCDE_RK4_38) return (j == 0 || j == 3) ? 0.125f : 0.375f;
    if (method == NCDE_MIDPOINT) return j == 0 ? 0.0f : 1.0f;
    return 1.0f;
}
int piece_index(float t, int n_pieces) {
    int idx = (int)std::ceil(t) - 1;
    idx = idx < 0 ? 0 : idx;
    return idx > n_pieces - 1 ? n_pieces - 1 : idx;
}

// dX/dt(t) of sample b
void dxdt(const NcdeProblem* p, const Dims& d, int b, float t, float* dx) {
    const int idx = piece_index(t, d.n_pieces);
    const float frac = t - (float)idx;
    const float* cp = p->coeffs + (long long)b * p->coeffs_stride_b + (long long)idx * p->coeffs_stride_t;
    for (int c = 0; c < d.C; ++c) {
        if (p->interp == NCDE_INTERP_LINEAR) dx[c] = cp[p->coeffs_stride_t + c] - cp[c];
        else {
            const float inner = cp[2 * d.C + c] + cp[3 * d.C + c] * frac;
            dx[c] = cp[d.C + c] + inner * frac;
        }
    }
}

struct Scratch {        // per-thread buffers of one stage evaluation
    std::vector<float> x[NCDE_MAX_LAYERS + 1];   // x[0] = stage input, x[l] = relu output of layer l
    std::vector<float> m;                        // tanh(P) [H*C]
    std::vector<float> g0, g1, dp;
    void init(const Dims& d) {
        x[0].resize(d.H);
        int D = d.H;
        for (int l = 0; l < d.L; ++l) { x[l + 1].resize(d.dout[l]); D = std::max(D, d.dout[l]); }
        m.resize((size_t)d.H * d.C);
        dp.resize((size_t)d.H * d.C);
        g0.resize(D); g1.resize(D);
    }
};

// k = f_theta(y) . dx   (keeps the activations in sc for a following VJP)
void stage_forward(const NcdeProblem* p, const Dims& d, const float* y, const float* dx, Scratch& sc, float* k) {
    for (int h = 0; h < d.H; ++h) sc.x[0][h] = y[h];
    for (int l = 0; l < d.L; ++l) {
        const float* W = p->layer_W[l];
        const float* bb = p->layer_b[l];
        for (int n = 0; n < d.dout[l]; ++n) {
            float acc = bb[n];
            const float* w = W + (long long)n * d.din[l];
            const float* xin = sc.x[l].data();
            const int K = d.din[l];
#pragma omp simd reduction(+ : acc)
            for (int q = 0; q < K; ++q) acc += w[q] * xin[q];
            sc.x[l + 1][n] = acc > 0.0f ? acc : 0.0f;
        }
    }
    const float* xl = sc.x[d.L].data();
    for (int h = 0; h < d.H; ++h) {
        float acc = 0.0f;
        for (int c = 0; c < d.C; ++c) {
            const int row = h * d.C + c;
            float pre = p->bo[row];
            const float* w = p->Wo + (long long)row * d.dlast;
            const int K = d.dlast;
#pragma omp simd reduction(+ : pre)
            for (int q = 0; q < K; ++q) pre += w[q] * xl[q];
            const float mm = std::tanh(pre);
            sc.m[row] = mm;
            acc += mm * dx[c];
        }
        k[h] = acc;
    }
}

// VJP of the stage just evaluated with cotangent ct[H]: dy = J^T ct; parameter gradients += w * ...
void stage_vjp(const NcdeProblem* p, const Dims& d, const float* dx, const float* ct, float w, Scratch& sc, float* dy, float** gW, float** gb,
               float* gWo, float* gbo) {
    const float* xl = sc.x[d.L].data();
    float* g = sc.g0.data();
    for (int q = 0; q < d.dlast; ++q) g[q] = 0.0f;
    for (int h = 0; h < d.H; ++h)
        for (int c = 0; c < d.C; ++c) {
            const int row = h * d.C + c;
            const float mm = sc.m[row];
            const float dpv = (ct[h] * dx[c]) * (1.0f - mm * mm);
            sc.dp[row] = dpv;
            const float* wr = p->Wo + (long long)row * d.dlast;
            for (int q = 0; q < d.dlast; ++q) g[q] += wr[q] * dpv;
            if (w != 0.0f) {
                gbo[row] += w * dpv;
                float* gr = gWo + (long long)row * d.dlast;
                for (int q = 0; q < d.dlast; ++q) gr[q] += (w * dpv) * xl[q];
            }
        }
    float* gin = sc.g0.data();
    float* gout = sc.g1.data();
    for (int l = d.L - 1; l >= 0; --l) {
        const int N = d.dout[l], K = d.din[l];
        for (int n = 0; n < N; ++n) gin[n] = sc.x[l + 1][n] > 0.0f ? gin[n] : 0.0f;      // dL/dpre_l
        if (w != 0.0f) {
            float* gw = gW[d.slot[l]];
            float* gbv = gb[d.slot[l]];
            for (int n = 0; n < N; ++n) {
                gbv[n] += w * gin[n];
                for (int q = 0; q < K; ++q) gw[(long long)n * K + q] += (w * gin[n]) * sc.x[l][q];
            }
        }
        const float* W = p->layer_W[l];
        for (int q = 0; q < K; ++q) gout[q] = 0.0f;
        for (int n = 0; n < N; ++n)
            for (int q = 0; q < K; ++q) gout[q] += W[(long long)n * K + q] * gin[n];
        float* tmp = gin; gin = gout; gout = tmp;
    }
    for (int h = 0; h < d.H; ++h) dy[h] = gin[h];
}

// Butcher bookkeeping of one state (rk_common.py:106-114, fixed_grid.py:6-29, dt = 1)
float combine(int method, int j, float k, float& y0, float& k1, float& k2, bool& last) {
    last = false;
    if (method == NCDE_RK4_38) {
        if (j == 0) { k1 = k; return y0 + k * 0.333333343267440796f; }
        if (j == 1) { k2 = k; return y0 + (k - k1 * 0.333333343267440796f); }
        if (j == 2) { const float ys = y0 + ((k1 - k2) + k); k2 = k2 + k; return ys; }
        last = true;
        y0 = y0 + ((k1 + 3.0f * k2) + k) * 0.125f;
        return y0;
    }
    if (method == NCDE_MIDPOINT) {
        if (j == 0) return y0 + k * 0.5f;
        last = true;
        y0 = y0 + k;
        return y0;
    }
    last = true;
    y0 = y0 + k;
    return y0;
}

struct GradBufs {        // per-thread parameter-gradient accumulators
    std::vector<float> W[NCDE_MAX_LAYERS], b[NCDE_MAX_LAYERS], Wo, bo;
    float* pW[NCDE_MAX_LAYERS];
    float* pb[NCDE_MAX_LAYERS];
    void init(const Dims& d) {
        for (int s = 0; s < d.n_slots; ++s) {
            const int l = d.slot_layer[s];
            W[s].assign((size_t)d.dout[l] * d.din[l], 0.0f);
            b[s].assign(d.dout[l], 0.0f);
            pW[s] = W[s].data(); pb[s] = b[s].data();
        }
        Wo.assign((size_t)d.H * d.C * d.dlast, 0.0f);
        bo.assign((size_t)d.H * d.C, 0.0f);
    }
};

int forward_impl(const NcdeProblem* p, float* out, float* stages) {
    Dims d;
    int rc = setup(p, &d);
    if (rc != NCDE_OK) return rc;
    if (!out || !p->coeffs || !p->z0 || !p->Wo || !p->bo) return fail(NCDE_ERR_INVALID, "NULL pointer");
#pragma omp parallel
    {
        Scratch sc;
        sc.init(d);
        std::vector<float> y0(d.H), ys(d.H), k1(d.H), k2(d.H), k(d.H), dx(d.C);
#pragma omp for schedule(static)
        for (int b = 0; b < d.B; ++b) {
            for (int h = 0; h < d.H; ++h) {
                y0[h] = ys[h] = p->z0[(long long)b * d.H + h];
                out[((long long)b * d.n_out) * d.H + h] = y0[h];
            }
            for (int n = 0; n < d.T - 1; ++n)
                for (int j = 0; j < d.S; ++j) {
                    dxdt(p, d, b, (float)n + stage_offset(p->method, j), dx.data());
                    if (stages) memcpy(stages + ((long long)(n * d.S + j) * d.B + b) * d.H, ys.data(), sizeof(float) * d.H);
                    stage_forward(p, d, ys.data(), dx.data(), sc, k.data());
                    bool last = false;
                    for (int h = 0; h < d.H; ++h) ys[h] = combine(p->method, j, k[h], y0[h], k1[h], k2[h], last);
                    if (last) {
                        if (p->output == NCDE_OUT_KNOTS) memcpy(out + ((long long)b * d.n_out + (n + 1)) * d.H, y0.data(), sizeof(float) * d.H);
                        else if (n == d.T - 2) memcpy(out + ((long long)b * d.n_out + 1) * d.H, y0.data(), sizeof(float) * d.H);
                    }
                }
        }
    }
    return NCDE_OK;
}

// shared by the continuous adjoint (src = z_out) and the exact discrete backward (src = stage record)
int backward_impl(const NcdeProblem* p, const float* src, const float* grad_out, const NcdeGrads* g, bool discrete) {
    Dims d;
    int rc = setup(p, &d);
    if (rc != NCDE_OK) return rc;
    if (!src || !grad_out || !g || !g->grad_z0 || !g->grad_Wo || !g->grad_bo) return fail(NCDE_ERR_INVALID, "NULL pointer");
    for (int s = 0; s < d.n_slots; ++s)
        if (!g->grad_layer_W[d.slot_layer[s]] || !g->grad_layer_b[d.slot_layer[s]]) return fail(NCDE_ERR_INVALID, "NcdeGrads: NULL destination");
    GradBufs total;
    total.init(d);
#pragma omp parallel
    {
        Scratch sc;
        sc.init(d);
        GradBufs gb;
        gb.init(d);
        std::vector<float> y0(d.H), ys(d.H), ky1(d.H), ky2(d.H), a0(d.H), as(d.H), ka1(d.H), ka2(d.H), kd2(d.H), k(d.H), dy(d.H), dx(d.C);
        const int last_row = d.n_out - 1;
#pragma omp for schedule(static)
        for (int b = 0; b < d.B; ++b) {
            for (int h = 0; h < d.H; ++h) {
                const long long o = ((long long)b * d.n_out + last_row) * d.H + h;
                a0[h] = grad_out[o];
                if (discrete) as[h] = p->method == NCDE_RK4_38 ? a0[h] * 0.125f : a0[h];
                else { y0[h] = ys[h] = src[o]; as[h] = a0[h]; }
            }
            for (int n = d.T - 1; n >= 1; --n)
                for (int j = 0; j < d.S; ++j) {
                    const float t = discrete ? (float)(n - 1) + stage_offset(p->method, d.S - 1 - j) : -(-(float)n + stage_offset(p->method, j));
                    const float w = discrete ? 1.0f : stage_weight(p->method, j);
                    dxdt(p, d, b, t, dx.data());
                    if (discrete) memcpy(ys.data(), src + ((long long)((n - 1) * d.S + (d.S - 1 - j)) * d.B + b) * d.H, sizeof(float) * d.H);
                    stage_forward(p, d, ys.data(), dx.data(), sc, k.data());
                    stage_vjp(p, d, dx.data(), as.data(), w, sc, dy.data(), gb.pW, gb.pb, gb.Wo.data(), gb.bo.data());
                    if (discrete) {
                        // transpose of the Butcher step (see oracle/ncde_oracle.py solve_discrete_backward)
                        bool last = false;
                        for (int h = 0; h < d.H; ++h) {
                            const float dd = dy[h];
                            float next = 0.0f;
                            if (p->method == NCDE_RK4_38) {
                                const float c4 = a0[h] * 0.125f;
                                if (j == 0) { ka1[h] = dd; next = 3.0f * c4 + dd; }
                                else if (j == 1) { ka2[h] = dd; next = (3.0f * c4 - ka1[h]) + dd; }
                                else if (j == 2) { kd2[h] = dd; next = ((c4 + ka1[h]) - 0.333333343267440796f * ka2[h]) + 0.333333343267440796f * dd; }
                                else { a0[h] = (((a0[h] + ka1[h]) + ka2[h]) + kd2[h]) + dd; last = true; }
                            } else if (p->method == NCDE_MIDPOINT) {
                                if (j == 0) { ka1[h] = dd; next = 0.5f * dd; }
                                else { a0[h] = (a0[h] + ka1[h]) + dd; last = true; }
                            } else { a0[h] = a0[h] + dd; last = true; }
                            if (last) {
                                if (p->output == NCDE_OUT_KNOTS || n == 1) a0[h] += grad_out[((long long)b * d.n_out + (p->output == NCDE_OUT_KNOTS ? n - 1 : 0)) * d.H + h];
                                next = p->method == NCDE_RK4_38 ? a0[h] * 0.125f : a0[h];
                            }
                            as[h] = next;
                        }
                        continue;
                    }
                    bool last = false;
                    for (int h = 0; h < d.H; ++h) {
                        ys[h] = combine(p->method, j, -k[h], y0[h], ky1[h], ky2[h], last);      // dy/ds = -f (misc.py:152-159)
                        as[h] = combine(p->method, j, dy[h], a0[h], ka1[h], ka2[h], last);      // da/ds = +a^T df/dy
                    }
                    if (last) {
                        for (int h = 0; h < d.H; ++h) {
                            if (p->output == NCDE_OUT_KNOTS) {      // reset y to the stored value, add dL/dz at this knot (adjoint.py:132-133)
                                const long long o = ((long long)b * d.n_out + (n - 1)) * d.H + h;
                                y0[h] = src[o];
                                a0[h] += grad_out[o];
                            } else if (n == 1) {
                                a0[h] += grad_out[((long long)b * d.n_out) * d.H + h];
                            }
                            ys[h] = y0[h];
                            as[h] = a0[h];
                        }
                    }
                }
            for (int h = 0; h < d.H; ++h) g->grad_z0[(long long)b * d.H + h] = a0[h];
        }
#pragma omp critical
        {
            for (int s = 0; s < d.n_slots; ++s) {
                for (size_t i = 0; i < gb.W[s].size(); ++i) total.W[s][i] += gb.W[s][i];
                for (size_t i = 0; i < gb.b[s].size(); ++i) total.b[s][i] += gb.b[s][i];
            }
            for (size_t i = 0; i < gb.Wo.size(); ++i) total.Wo[i] += gb.Wo[i];
            for (size_t i = 0; i < gb.bo.size(); ++i) total.bo[i] += gb.bo[i];
        }
    }
    for (int s = 0; s < d.n_slots; ++s) {
        const int l = d.slot_layer[s];
        memcpy(g->grad_layer_W[l], total.W[s].data(), sizeof(float) * total.W[s].size());
        memcpy(g->grad_layer_b[l], total.b[s].data(), sizeof(float) * total.b[s].size());
    }
    memcpy(g->grad_Wo, total.Wo.data(), sizeof(float) * total.Wo.size());
    memcpy(g->grad_bo, total.bo.data(), sizeof(float) * total.bo.size());
    return NCDE_OK;
}

}  // namespace

extern "C" {

int ncde_version(void) { return NCDE_ABI_VERSION; }
// not part of include/ncde_hip.h: how many OpenMP threads the restatement uses (returns the previous maximum)
int ncde_cpu_set_threads(int n) {
    const int prev = omp_get_max_threads();
    if (n > 0) omp_set_num_threads(n);
    return prev;
}
const char* ncde_last_error_string(void) { return g_err; }
int ncde_num_outputs(const NcdeProblem* p) {
    Dims d;
    const int rc = setup(p, &d);
    return rc != NCDE_OK ? rc : d.n_out;
}
int64_t ncde_workspace_bytes(const NcdeProblem* p, int) {
    Dims d;
    const int rc = setup(p, &d);
    return rc != NCDE_OK ? rc : 0;
}
const char* ncde_kernel_name(const NcdeProblem* p, int) {
    Dims d;
    return setup(p, &d) == NCDE_OK ? "cpu_scalar_openmp" : nullptr;
}
int64_t ncde_stage_record_bytes(const NcdeProblem* p) {
    Dims d;
    const int rc = setup(p, &d);
    return rc != NCDE_OK ? rc : (int64_t)sizeof(float) * (d.T - 1) * d.S * (int64_t)d.B * d.H;
}
int ncde_forward(const NcdeProblem* p, float* out, void*, size_t, void*) { return forward_impl(p, out, nullptr); }
int ncde_forward_record(const NcdeProblem* p, float* out, float* stages, void*, size_t, void*) {
    if (!stages) return fail(NCDE_ERR_INVALID, "stages is NULL");
    return forward_impl(p, out, stages);
}
int ncde_adjoint(const NcdeProblem* p, const float* z_out, const float* grad_out, const NcdeGrads* g, void*, size_t, void*) {
    return backward_impl(p, z_out, grad_out, g, false);
}
int ncde_backward(const NcdeProblem* p, const float* stages, const float* grad_out, const NcdeGrads* g, void*, size_t, void*) {
    return backward_impl(p, stages, grad_out, g, true);
}

}  // extern "C"
