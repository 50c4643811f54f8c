// Host side of the general time axis: builds the time plan the generic / variant kernels walk (layout: ncde_common.h).
// Pure CPU code (no kernels): every time-like quantity of the reference's fixed-step solve is evaluated here, in the
// dtype of the caller's `t` and in torch's operation order, so that the kernels only consume (piece index, fraction,
// knot spacing) triples and per-step dt's.  Reference lines restated (relative to /root/reference/modules):
//   grid from step_size            torchdiffeq/torchdiffeq/_impl/solvers.py:78-87
//   output pick / interpolation    torchdiffeq/torchdiffeq/_impl/solvers.py:103-117, 166-172
//   stage times                    torchdiffeq/torchdiffeq/_impl/fixed_grid.py:6-29, rk_common.py:106-114
//   time cast to the state dtype   torchdiffeq/torchdiffeq/_impl/misc.py:181
//   knot index, fraction           torchcde/torchcde/interpolation_linear.py:212-219, interpolation_cubic.py:315-322
//   one reverse solve per output interval, own grid, negated time   torchdiffeq/_impl/adjoint.py:116-133, misc.py:262-271
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "ncde_common.h"
#include "ncde_timeplan.h"

namespace {

template <class T>
std::vector<T> fixed_grid(T start, T end, T step) {
    // niters = ceil((end - start) / step + 1); arange(0, niters) * step + start; last := end
    const T q = (end - start) / step;
    const T r = q + (T)1;
    const double niters = std::ceil((double)r);
    std::vector<T> g;
    const long long n = (long long)niters;
    g.reserve((size_t)(n > 0 ? n : 0));
    for (long long k = 0; k < n; ++k) {
        const T kk = (T)k;
        const T prod = kk * step;
        g.push_back(prod + start);
    }
    if (!g.empty()) g.back() = end;
    return g;
}

struct Knots {
    const double* user;   // NULL = default integer grid
    int n;
    float at(int i) const { return user ? (float)user[i] : (float)i; }
    // bucketize(t, knots, right=False) - 1, clamped to [0, n_pieces - 1]
    int piece(float t) const {
        int lo = 0, hi = n;               // first index with knots[i] >= t
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (at(mid) < t) lo = mid + 1;
            else hi = mid;
        }
        int idx = lo - 1;
        if (idx < 0) idx = 0;
        if (idx > n - 2) idx = n - 2;
        return idx;
    }
};

void put_stage(int* dst, const Knots& kn, float t) {
    const int idx = kn.piece(t);
    const float k0 = kn.at(idx), k1 = kn.at(idx + 1);
    const float frac = t - k0, kdt = k1 - k0;
    dst[0] = idx;
    memcpy(dst + 1, &frac, 4);
    memcpy(dst + 2, &kdt, 4);
}

// stage times of one step [t0, t1] in the arithmetic of dtype T; `neg` = the solve runs in negated time (adjoint)
template <class T>
void put_step_stages(int* dst, int method, T t0, T dt, T t1, bool neg, const Knots& kn) {
    T ts[4];
    int S = 1;
    ts[0] = t0;
    if (method == NCDE_RK4_38) {
        const T third = (T)(1.0 / 3.0), two_thirds = (T)(2.0 / 3.0);
        const T d1 = dt * third, d2 = dt * two_thirds;
        ts[1] = t0 + d1;
        ts[2] = t0 + d2;
        ts[3] = t1;
        S = 4;
    } else if (method == NCDE_MIDPOINT) {
        const T half = (T)0.5 * dt;
        ts[1] = t0 + half;
        S = 2;
    }
    for (int j = 0; j < S; ++j) {
        const T real_t = neg ? -ts[j] : ts[j];
        put_stage(dst + 3 * j, kn, (float)real_t);   // t.to(y.dtype): the control path sees the fp32 value
    }
}

template <class T>
int build(const NcdeProblem* p, const NcdeTimeSpec* ts, int* w, size_t cap_words, NcdeTimePlanInfo* info, char* err, size_t errn) {
    const int S = p->method == NCDE_RK4_38 ? 4 : (p->method == NCDE_MIDPOINT ? 2 : 1);
    const int pw = plan_step_words(S);
    const int nt = ts->n_t;
    std::vector<T> t(nt);
    for (int i = 0; i < nt; ++i) t[i] = (T)ts->t[i];
    for (int i = 1; i < nt; ++i)
        if (!(t[i] > t[i - 1])) {
            snprintf(err, errn, "t must be strictly increasing (decreasing output times are outside the fused path)");
            return NCDE_ERR_INVALID;
        }
    const T step = (T)ts->step_size;
    if (!(ts->step_size > 0.0)) { snprintf(err, errn, "step_size must be positive"); return NCDE_ERR_INVALID; }
    Knots kn{ts->knots, p->n_knots};
    if (ts->knots)
        for (int i = 1; i < p->n_knots; ++i)
            if (!((float)ts->knots[i] > (float)ts->knots[i - 1])) { snprintf(err, errn, "knot grid must be strictly increasing"); return NCDE_ERR_INVALID; }

    {   // the table is indexed with 32-bit word offsets (and built in host memory): refuse grids it cannot describe
        const double est = (ts->t[nt - 1] - ts->t[0]) / ts->step_size + (double)nt + 2.0;      // steps per direction, roughly
        const double cap = 2147483647.0 / (double)pw / 2.0 - 16.0;
        if (!(est < cap)) {
            snprintf(err, errn, "step_size %g over [%g, %g]: about %.3g steps, more than a time plan can hold (%.3g)", ts->step_size, ts->t[0], ts->t[nt - 1], est, cap);
            return NCDE_ERR_INVALID;
        }
    }
    const std::vector<T> grid = fixed_grid<T>(t[0], t[nt - 1], step);
    if (grid.size() < 2 || !(grid.front() == t[0])) { snprintf(err, errn, "degenerate time grid"); return NCDE_ERR_INVALID; }
    const int n_fwd = (int)grid.size() - 1;
    // reverse solves: one per output interval, each with its own grid in negated time
    std::vector<std::vector<T>> rgrids;
    long long n_adj = 0;
    for (int i = nt - 1; i >= 1; --i) {
        rgrids.push_back(fixed_grid<T>(-t[i], -t[i - 1], step));
        if (rgrids.back().size() < 2) { snprintf(err, errn, "degenerate reverse time grid"); return NCDE_ERR_INVALID; }
        n_adj += (long long)rgrids.back().size() - 1;
    }
    const long long words = (long long)NCDE_PLAN_HEADER + (long long)(n_fwd + n_adj) * pw + 2LL * nt;
    info->n_t_out = nt;
    info->n_steps_fwd = n_fwd;
    info->n_steps_adj = (int)n_adj;
    info->stages = S;
    info->bytes = words * 4;
    if (!w) return NCDE_OK;
    if ((long long)cap_words < words) { snprintf(err, errn, "time plan buffer %zu B < %lld B", cap_words * 4, words * 4); return NCDE_ERR_WORKSPACE; }
    memset(w, 0, (size_t)words * 4);
    const int off_fwd = plan_off_fwd(), off_out = plan_off_out(S, n_fwd), off_adj = plan_off_adj(S, n_fwd, nt);
    w[0] = NCDE_PLAN_MAGIC; w[1] = S; w[2] = n_fwd; w[3] = (int)n_adj; w[4] = nt; w[5] = off_fwd; w[6] = off_out; w[7] = off_adj;
    // forward steps + the outputs each one brackets
    int j = 1;
    for (int n = 0; n < n_fwd; ++n) {
        int* st = w + off_fwd + n * pw;
        const T t0 = grid[n], t1 = grid[n + 1];
        const T dt = t1 - t0;
        const float dtf = (float)dt;
        memcpy(st, &dtf, 4);
        put_step_stages<T>(st + 3, p->method, t0, dt, t1, false, kn);
        st[1] = j;
        int cnt = 0;
        while (j < nt && t1 >= t[j]) {
            int* o = w + off_out + 2 * j;
            float slope = 0.0f;
            if (t[j] == t0) o[0] = 0;
            else if (t[j] == t1) o[0] = 1;
            else {
                o[0] = 2;
                const T num = t[j] - t0, den = t1 - t0;
                slope = (float)(num / den);
            }
            memcpy(o + 1, &slope, 4);
            ++j; ++cnt;
        }
        st[2] = cnt;
    }
    if (j != nt) { snprintf(err, errn, "internal: %d of %d output times placed", j, nt); return NCDE_ERR_INVALID; }
    // reverse steps
    int r = 0;
    for (int k = 0; k < (int)rgrids.size(); ++k) {
        const std::vector<T>& g = rgrids[k];
        const int row = nt - 2 - k;     // output row the solve ends at
        for (size_t q = 0; q + 1 < g.size(); ++q, ++r) {
            int* st = w + off_adj + r * pw;
            const T s0 = g[q], s1 = g[q + 1];
            const T dt = s1 - s0;
            const float dtf = (float)dt;
            memcpy(st, &dtf, 4);
            st[1] = q + 2 == g.size() ? row : -1;
            put_step_stages<T>(st + 3, p->method, s0, dt, s1, true, kn);
        }
    }
    return NCDE_OK;
}

}  // namespace

int ncde_time_plan_build_impl(const NcdeProblem* p, const NcdeTimeSpec* ts, void* host_buffer, size_t bytes, NcdeTimePlanInfo* info,
                              char* err, size_t errn) {
    if (!ts || !ts->t || ts->n_t < 2) { snprintf(err, errn, "time spec: need >= 2 output times"); return NCDE_ERR_INVALID; }
    if (!info) { snprintf(err, errn, "info is NULL"); return NCDE_ERR_INVALID; }
    if (ts->time_is_f64) return build<double>(p, ts, (int*)host_buffer, bytes / 4, info, err, errn);
    return build<float>(p, ts, (int*)host_buffer, bytes / 4, info, err, errn);
}
