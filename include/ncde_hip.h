/*
 * ncde_hip.h -- C-ABI of libncde_hip.so, the MI355X (gfx950) Neural-CDE fixed-step integrator.
 *
 * The reference (jambo6/online-neural-cdes) has no FFI layer: its hot path is two Python callables,
 *   torchcde.cdeint(X, func, z0, t, adjoint=True, ...)        modules/torchcde/torchcde/solver.py:140-238
 *   NeuralCDE.forward(inputs)                                  src/ncde/ncde.py:214-243
 * which expand to the Python time loop of torchdiffeq's FixedGridODESolver.integrate
 * (modules/torchdiffeq/torchdiffeq/_impl/solvers.py:94-119) and, for the backward pass,
 * OdeintAdjointMethod.backward (modules/torchdiffeq/torchdiffeq/_impl/adjoint.py:37-145).
 * The entry points below are what a maintainer of the reference would bind (ctypes, see
 * INTEGRATION.md) to replace exactly those two loops.
 *
 * Conventions
 *   - every pointer inside NcdeProblem is a DEVICE pointer to fp32 data owned by the caller;
 *   - the library is stateless and re-entrant; work is enqueued on `stream` (a hipStream_t, NULL =
 *     default stream) and the calls never synchronise and never allocate;
 *   - scratch memory is passed in: ask ncde_workspace_bytes(), hand over `workspace` (per-workgroup gradient partials; for the
 *     specialised kernels also one range-fault word per 16-sample tile -- they multiply in 2-way split-fp16 and re-execute a tile
 *     in 3-way split-bf16, with a second launch on the same stream, when its operands leave the fp16 range: NCDE_FLAG_SPLIT_BF16);
 *   - return value 0 = success, negative = NcdeStatus; ncde_last_error_string() describes the last
 *     failure of the calling thread.  Nothing throws across the boundary.
 */
#ifndef NCDE_HIP_H
#define NCDE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NCDE_ABI_VERSION 4 /* version-1 structs (no trailing field_kind .. br members) and version-2 structs (no trailing
                              time_plan .. members) are still accepted */
#define NCDE_MAX_LAYERS 8

typedef enum NcdeStatus {
    NCDE_OK = 0,
    NCDE_ERR_INVALID = -1,      /* malformed problem (maps to ValueError / AssertionError in Python)    */
    NCDE_ERR_UNSUPPORTED = -2,  /* well-formed but outside what the kernels cover                      */
    NCDE_ERR_WORKSPACE = -3,    /* workspace too small                                                 */
    NCDE_ERR_HIP = -4           /* a HIP runtime call failed (launch error, no device, ...)            */
} NcdeStatus;

/* control-path evaluator: replaces LinearInterpolation.derivative (interpolation_linear.py:212-234)
 * and NaturalCubicSpline.derivative (interpolation_cubic.py:315-336) on the default integer grid. */
typedef enum NcdeInterp { NCDE_INTERP_LINEAR = 0, NCDE_INTERP_CUBIC = 1 } NcdeInterp;

/* fixed-step solver: replaces Euler / Midpoint / RK4 (3/8 rule) of fixed_grid.py:6-29,
 * rk_common.py:106-114 with options={'step_size': 1}. */
typedef enum NcdeMethod { NCDE_EULER = 0, NCDE_MIDPOINT = 1, NCDE_RK4_38 = 2 } NcdeMethod;

/* which times the solution is reported at: t = X.interval (2 outputs: z(0), z(T-1)) or
 * t = X.grid_points (every knot), the two cases src/ncde/ncde.py:219-225 uses on the default grid with step_size 1;
 * NCDE_OUT_TIMES = the general time axis (any increasing t, any step_size, user knot grids): needs a time plan. */
typedef enum NcdeOutput { NCDE_OUT_INTERVAL = 0, NCDE_OUT_KNOTS = 1, NCDE_OUT_TIMES = 2 } NcdeOutput;

/* kernel family selection (ncde_forward/ncde_adjoint `flags`) */
#define NCDE_FLAG_AUTO 0u
#define NCDE_FLAG_FORCE_GENERIC 1u  /* never use a shape-specialised kernel */
#define NCDE_FLAG_FORCE_FAST 2u     /* fail with NCDE_ERR_UNSUPPORTED if no specialised kernel fits */
#define NCDE_FLAG_FP32_MFMA 4u      /* specialised and batch-tiled kernels: plain fp32-input MFMA (and fp32 records) instead of the
                                       (default, fp32-equivalent) 3-way split-bf16 MFMA GEMMs */
#define NCDE_FLAG_ADJOINT_V1 8u     /* specialised adjoint: single-role kernel instead of the (default) chain+gradient
                                       wave-specialised one */
#define NCDE_FLAG_ADJOINT_V2 16u    /* specialised adjoint: chain+gradient kernel with an fp32-MFMA chain (default: split-bf16 chain) */
#define NCDE_FLAG_ADJOINT_V4 32u    /* specialised adjoint: decoupled y / cotangent waves (ncde_adj_fast4: y re-integration one stage ahead of
                                       the cotangent chain, split-bf16 dL/dx_L) instead of the (default, faster) chain+gradient kernel */
#define NCDE_FLAG_SPLIT_BF16 64u    /* specialised forward kernels: 3-way split-bf16 GEMMs (6 MFMAs per product, any operand magnitude) instead of
                                       the default 2-way split-fp16 ones (3 MFMAs per product; sample tiles whose operands leave the fp16 range
                                       are re-executed by the split-bf16 kernel, so results do not depend on this flag beyond fp32 round-off) */
#define NCDE_FLAG_ADJOINT_SPLIT_FP16 128u /* development builds (-DNCDE_DEV_KNOBS) only, ignored otherwise: split-fp16 GEMMs on the cotangent
                                       side of the specialised adjoint too (not reproducible run to run) */
#define NCDE_FLAG_DEBUG_PROFILE 0x100u /* development: instrumented kernel variant, cycle counters land in the workspace */
#define NCDE_FLAG_NO_COOP 0x400u       /* batch-tiled forward and backward: do not use the XCD-cooperative, weight-stationary output phase (round 5, large
                                          hidden sizes: workgroups of one launch exchange activations through L2 and spin on each other --
                                          it needs every workgroup resident, i.e. the GPU's CUs not held by another process's persistent kernel) */
#define NCDE_FLAG_COOP_FAULT_INJECT 0x800u /* verification (like the dopri5 replay): ONE workgroup of the first cooperative launch withholds its first
                                          arrival and the spin limit is shortened, so that launch gives up after ~1 ms: exercises the status word
                                          and the re-execution by the per-workgroup kernels (tests/test_gpu_parity.py); results must not change */
#define NCDE_FLAG_TILED_NS1 0x1000u    /* batch-tiled forward: force 1 / 2 / 4 sixteen-sample tiles per workgroup     */
#define NCDE_FLAG_TILED_NS2 0x2000u    /*   (default: the largest that still gives >= 256 workgroups)                 */
#define NCDE_FLAG_TILED_NS4 0x4000u
#define NCDE_FLAG_FORCE_TILED 0x8000u  /* use the batch-tiled family also where a shape-specialised kernel exists (default order:
                                          specialised, then batch-tiled wherever its shape constraints hold, then generic).
                                          Its backward keeps per-stage records for a WINDOW of steps only (sweep W steps, fold
                                          them into the output-layer gradient, reuse the record): workspace O(B H W), W sized to
                                          a 192 MB record budget (NCDE_FLAG_TILED_WINDOW_STEPS(n) overrides) */

#define NCDE_FLAG_TILED_WINDOW_STEPS(n) (((uint32_t)(n) & 0xFFu) << 16) /* batch-tiled backward: n (1..255) solver steps per time window
                                          instead of what the record budget allows (0 = budget).  The library reads no environment
                                          variable on any call path; development switches exist only in builds with -DNCDE_DEV_KNOBS */

typedef struct NcdeProblem {
    int32_t abi_version;  /* = NCDE_ABI_VERSION */
    int32_t batch;        /* B */
    int32_t n_knots;      /* T: knots of the control path (linear: coeffs rows; cubic: rows + 1) */
    int32_t channels;     /* C: control channels incl. time */
    int32_t hidden;       /* H: state size */
    int32_t interp;       /* NcdeInterp */
    int32_t method;       /* NcdeMethod */
    int32_t output;       /* NcdeOutput */
    uint32_t flags;       /* NCDE_FLAG_* */

    /* vector field f_theta (src/ncde/vector_fields/base.py:64-69, 83-104):
     *   x_0 = z;  x_l = relu(W_l x_{l-1} + b_l), l = 1..n_layers;  M = tanh(Wo x_n + bo) viewed [H, C]
     * layer_W[l] is [layer_out[l], layer_in[l]] row-major (torch Linear layout); the same pointer may
     * appear in several slots (the reference shares ONE inner layer nl-1 times, base.py:66-68).
     * Wo is [H*C, layer_out[n_layers-1]] with row index h*C + c. */
    int32_t n_layers;
    int32_t layer_in[NCDE_MAX_LAYERS];
    int32_t layer_out[NCDE_MAX_LAYERS];
    const float* layer_W[NCDE_MAX_LAYERS];
    const float* layer_b[NCDE_MAX_LAYERS];
    const float* Wo;
    const float* bo;

    /* control-path coefficients, row-major, element strides:
     *   linear/rectilinear: coeffs[b][t][c], t < T           (linear_interpolation_coeffs output)
     *   cubic:              coeffs[b][p][4C] = a|b|2c|3d, p < T-1 (natural_cubic_coeffs output)    */
    const float* coeffs;
    int64_t coeffs_stride_b;
    int64_t coeffs_stride_t;

    const float* z0; /* [B, H] contiguous */

    /* ---- ABI version 2: vector-field variants (read only when abi_version >= 2; version 1 = original / matmul) ----
     * field_kind  (src/ncde/vector_fields/gating.py:7-61)
     *   NCDE_FIELD_ORIGINAL  M = tanh(Wo hh + bo)
     *   NCDE_FIELD_MINIMAL   M = sigmoid(Wg hh + bg) * tanh(Wo hh + bo)
     *   NCDE_FIELD_GRU       M = sigmoid(Wg net(u) + bg) * tanh(Wo net(sigmoid(Wr u + br) * u) + bo)
     * field_input (vector_field_type of cdeint, modules/torchcde/torchcde/solver.py:112-137)
     *   NCDE_INPUT_MATMUL      u = z;             heads have H*C rows, dz/dt = M[H,C] . dX/dt
     *   NCDE_INPUT_EVALUATE    u = [z, X(t)];     heads have H rows,   dz/dt = M        (layer_in[0] = H + C)
     *   NCDE_INPUT_DERIVATIVE  u = [z, dX/dt(t)]; heads have H rows,   dz/dt = M
     * Wg, bg: the sigmoid head (same shape as Wo, bo); Wr [d0, d0], br [d0]: the GRU reset net, d0 = layer_in[0]. */
    int32_t field_kind;
    int32_t field_input;
    const float* Wg;
    const float* bg;
    const float* Wr;
    const float* br;

    /* ---- ABI version 3: general time axis (read only when abi_version >= 3 and output == NCDE_OUT_TIMES) ----
     * time_plan: DEVICE copy of the buffer ncde_time_plan_build() filled on the host; the three counts are the ones it
     * returned in NcdeTimePlanInfo.  The solution then has n_t_out rows per sample (row 0 = z0 = z(t[0])). */
    const void* time_plan;
    int32_t n_t_out;
    int32_t n_steps_fwd;
    int32_t n_steps_adj;
    int32_t reserved_;      /* padding; NOT an input -- the library ignores whatever the caller leaves here (it is zeroed on entry) */
} NcdeProblem;

typedef enum NcdeFieldKind { NCDE_FIELD_ORIGINAL = 0, NCDE_FIELD_MINIMAL = 1, NCDE_FIELD_GRU = 2 } NcdeFieldKind;
typedef enum NcdeFieldInput { NCDE_INPUT_MATMUL = 0, NCDE_INPUT_EVALUATE = 1, NCDE_INPUT_DERIVATIVE = 2 } NcdeFieldInput;

/* gradient outputs of the adjoint sweep; aliasing must mirror NcdeProblem.layer_W/layer_b
 * (a shared layer receives the SUM over its uses).  Buffers are overwritten, not accumulated. */
typedef struct NcdeGrads {
    float* grad_z0;                      /* [B, H] */
    float* grad_layer_W[NCDE_MAX_LAYERS];
    float* grad_layer_b[NCDE_MAX_LAYERS];
    float* grad_Wo;
    float* grad_bo;
    /* read only when the problem's abi_version >= 2 and the field kind has these parameters */
    float* grad_Wg;
    float* grad_bg;
    float* grad_Wr;
    float* grad_br;
} NcdeGrads;

/* General time axis of torchdiffeq's fixed-grid solvers, replacing
 *   the grid built from options['step_size']                     torchdiffeq/_impl/solvers.py:78-87
 *   the output pick / linear interpolation between grid states    solvers.py:103-117, 166-172
 *   user knot grids of the control path                           torchcde/interpolation_linear.py:186-202, interpolation_cubic.py:283-305
 *   one reverse solve per output interval on its own grid         torchdiffeq/_impl/adjoint.py:116-133
 * The caller describes the axis with HOST arrays; ncde_time_plan_build() (pure CPU code, exact torch arithmetic in the
 * dtype of t) turns it into a small table the kernels walk.  Copy the table to the device, point NcdeProblem.time_plan
 * at it, set output = NCDE_OUT_TIMES and the three counts from NcdeTimePlanInfo. */
typedef struct NcdeTimeSpec {
    int32_t n_t;          /* number of output times (>= 2); t[0] is where the solve starts (z0 = z(t[0])) */
    int32_t time_is_f64;  /* 0: the caller's t tensor is fp32 (grid arithmetic in fp32, as torch does); 1: fp64 */
    const double* t;      /* HOST, n_t strictly increasing values */
    double step_size;     /* options['step_size'] > 0 */
    const double* knots;  /* HOST, the n_knots knot times X was built on, or NULL = default integer grid 0..T-1 */
} NcdeTimeSpec;

typedef struct NcdeTimePlanInfo {
    int32_t n_t_out;      /* = n_t */
    int32_t n_steps_fwd;  /* solver steps of the forward solve */
    int32_t n_steps_adj;  /* solver steps of all reverse solves of the continuous adjoint */
    int32_t stages;       /* stages per step of the method */
    int64_t bytes;        /* size of the plan */
} NcdeTimePlanInfo;

/* host_buffer == NULL: only fills *info (sizes).  Uses p->method, p->n_knots. */
int ncde_time_plan_build(const NcdeProblem* p, const NcdeTimeSpec* ts, void* host_buffer, size_t bytes, NcdeTimePlanInfo* info);

/* ---- adaptive Dormand-Prince 5(4): method='dopri5' of torchdiffeq (the default method of cdeint; NeuralCDE(solver="dopri5")
 * passes options {'min_step': 0.5}, src/ncde/ncde.py:130-134).  Replaces Dopri5Solver / RKAdaptiveStepsizeODESolver
 * (torchdiffeq/_impl/dopri5.py:5-36, rk_common.py:117-313, misc.py:33-103, interp.py) and, for the backward pass, the
 * per-interval adaptive solves of OdeintAdjointMethod.backward over (vjp_t, y, a, g_theta) with the mixed error norm
 * (adjoint.py:37-145, 235-247).  The error norm is over the WHOLE batch (one accept / reject per attempt), the time is kept
 * in fp64 on the device.  These two calls DO synchronise the stream (the number of attempts is data dependent).
 * p->method and p->output are ignored; the solution has ts->n_t rows per sample (row 0 = z0 = z(t[0])); ts->step_size is
 * ignored.  Original field with the matmul input only. */
typedef struct NcdeAdaptiveOptions {
    double rtol, atol;      /* cdeint defaults: 1e-4, 1e-6 (torchcde/solver.py:193-196) */
    double min_step;        /* options['min_step'], default 0 */
    double max_step;        /* options['max_step'], 0 = infinity */
    double first_step;      /* options['first_step'], 0 = select automatically (misc.py:33-74) */
    double safety, ifactor, dfactor; /* 0 = the reference's defaults 0.9, 10, 0.2 */
    int32_t max_num_steps;  /* 0 = 2^31 - 1 */
    int32_t trace_capacity; /* diagnostics: number of attempts `trace` has room for (0 = none) */
    double* trace;          /* HOST buffer of trace_capacity x 4 doubles, filled after the solve with one row per attempt:
                               t0, dt, accepted (0/1), error ratio -- the step sequence, for comparison with the reference's */
    /* ---- read only when the problem's abi_version >= 4 ----
     * Replay of a recorded step sequence (verification): attempt i (counted over the whole call, all output intervals of an adjoint
     * solve included) takes dt = replay[2 i] and is accepted iff replay[2 i + 1] != 0 instead of what the controller would decide;
     * attempts beyond replay_count run free.  With the reference's own sequence (tests/golden/g10, g12: `trace_fwd` / `trace_bwd`)
     * the solve does the reference's arithmetic step for step, so the comparison with its outputs is tight instead of
     * solver-tolerance level. */
    int32_t replay_count;
    int32_t reserved_;
    const double* replay;   /* HOST, replay_count x 2 doubles */
} NcdeAdaptiveOptions;

typedef struct NcdeAdaptiveStats {
    int32_t nfe;            /* vector-field evaluations, as the reference's func.nfe counts them (base.py:90) */
    int32_t n_accepted, n_rejected;
    int32_t reserved_;
} NcdeAdaptiveStats;

int64_t ncde_dopri5_workspace_bytes(const NcdeProblem* p, const NcdeTimeSpec* ts, int pass /* 0 forward, 1 adjoint, 2 taped backward */);
/* Which kernels a dopri5 call of this problem runs (static string; NULL + last error if the problem is not supported): the fused
 * attempt kernels of round 4 -- one launch per attempt for the forward solve ("ncde_dpf_fwd<...>": hidden, hidden_hidden <= 32 with
 * C <= 20, or <= 64 with C <= 4), an attempt + a reduce launch for the adjoint ("ncde_dpf_adj<...> + ncde_dpf_reduce") and the persistent
 * reverse sweep of a taped solve ("ncde_dpf_tape<...>"; both: the (32, 32, 20) set, <= 3 layers -- a fourth layer's LDS images do
 * not fit 160 KB, so the library caps n_layers at 3 there) -- or the per-launch kernels
 * ("ncde_dp_stage x 6 + ncde_dp_control + ncde_dp_commit", "ncde_dp_tape_backward") for every other shape and under
 * NCDE_FLAG_FORCE_GENERIC. */
const char* ncde_dopri5_kernel_name(const NcdeProblem* p, int pass /* 0 forward, 1 adjoint, 2 taped backward */);
int ncde_dopri5_forward(const NcdeProblem* p, const NcdeTimeSpec* ts, const NcdeAdaptiveOptions* opt, float* out, void* workspace,
                        size_t workspace_bytes, void* stream, NcdeAdaptiveStats* stats);
int ncde_dopri5_adjoint(const NcdeProblem* p, const NcdeTimeSpec* ts, const NcdeAdaptiveOptions* opt, const float* z_out,
                        const float* grad_out, const NcdeGrads* grads, void* workspace, size_t workspace_bytes, void* stream,
                        NcdeAdaptiveStats* stats);

/* dopri5 with adjoint=False -- how the reference's shipped "interpolation" experiments run it (experiments/configurations/
 * configurations.json5:187-191 -> torchdiffeq.odeint under autograd, torchcde/solver.py:224-225).  Replaces the autograd tape of
 * RKAdaptiveStepsizeODESolver (rk_common.py:216-305): ncde_dopri5_forward_record is ncde_dopri5_forward that also keeps, per ACCEPTED
 * step, (t0, dt, the state at its start, the outputs interpolated in it) in a caller-owned device record; ncde_dopri5_backward is the
 * exact reverse-mode sweep over that record -- six stage VJPs per step with FSAL, the transpose of the 4th-order dense output
 * (interp.py:4-61), and the gradient of the FIRST step size through _select_initial_step (misc.py:33-74; every later step size is a
 * constant because _optimal_step_size is @torch.no_grad(), misc.py:84-97).  The backward first reads the record's 64-byte header back
 * (step count, overflow flag: one small copy + stream synchronisation; a second one uploads a user knot grid), then enqueues one
 * persistent launch + two small ones + the partial reduction; the forward synchronises like ncde_dopri5_forward.
 * Limits of the taped path (checked by ncde_dopri5_record_bytes / _forward_record, NCDE_ERR_UNSUPPORTED): hidden <= 128, last
 * hidden width <= 128, the reverse sweep's 7 [H][16] LDS arrays within 160 KB.
 * Record size: ncde_dopri5_record_bytes() is enough for every solve with options.min_step > 0 (never more than max_num_steps per
 * output interval); without a minimum step it is an estimate (about two steps per knot) that a solve can exceed: it then returns
 * NCDE_ERR_WORKSPACE ("pass a larger record") and may simply be repeated with a larger one (the Python host doubles it).  Workspace of the backward:
 * ncde_dopri5_workspace_bytes(p, ts, 2). */
int64_t ncde_dopri5_record_bytes(const NcdeProblem* p, const NcdeTimeSpec* ts, const NcdeAdaptiveOptions* opt);
int ncde_dopri5_forward_record(const NcdeProblem* p, const NcdeTimeSpec* ts, const NcdeAdaptiveOptions* opt, float* out, void* record,
                               size_t record_bytes, void* workspace, size_t workspace_bytes, void* stream, NcdeAdaptiveStats* stats);
int ncde_dopri5_backward(const NcdeProblem* p, const NcdeTimeSpec* ts, const NcdeAdaptiveOptions* opt, const void* record,
                         size_t record_bytes, const float* grad_out, const NcdeGrads* grads, void* workspace, size_t workspace_bytes,
                         void* stream);

int ncde_version(void);
const char* ncde_last_error_string(void);

/* number of solution rows per sample: 2 (interval) or T (knots) */
int ncde_num_outputs(const NcdeProblem* p);

/* scratch bytes needed by ncde_forward[_record] (pass = 0) / ncde_adjoint (pass = 1) / ncde_backward (pass = 2);
 * negative = NcdeStatus */
int64_t ncde_workspace_bytes(const NcdeProblem* p, int pass);

/* The XCD-cooperative kernels of the batch-tiled family (large hidden sizes) spin on each other and therefore carry a bounded wait: a
 * launch whose workgroups never all became resident (another process's persistent kernel, a CU mask) GIVES UP, and the per-workgroup
 * kernels the library always enqueues behind it re-execute the pass -- the caller's outputs are correct either way, only late.  That it
 * happened is recorded in a STATUS WORD inside the caller's workspace: uint32 at byte offset ncde_coop_status_offset(p, pass), zeroed
 * by the call, 1 after a cooperative launch of that call gave up.  Read it after the stream has finished the call (it is the caller's
 * memory; the library never synchronises); a caller that sees 1 should pass NCDE_FLAG_NO_COOP from then on (the Python host does).
 * Returns NCDE_ERR_UNSUPPORTED (-2) when the problem / pass launches nothing cooperative. */
int64_t ncde_coop_status_offset(const NcdeProblem* p, int pass);

/* name of the kernel family the call would dispatch to ("generic", "fast_h32_c20", ...); NULL on error */
const char* ncde_kernel_name(const NcdeProblem* p, int pass);

/* Forward solve.  out: [B, n_out, H]; row 0 is z0, then the solution at the requested times.
 * Replaces odeint(...)/FixedGridODESolver.integrate under torch.no_grad (adjoint.py:24-33). */
int ncde_forward(const NcdeProblem* p, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* Continuous-adjoint reverse sweep (adjoint.py:37-145): given the forward outputs z_out [B,n_out,H]
 * and dL/dz_out grad_out [B,n_out,H], writes dL/dz0 and dL/dtheta. */
int ncde_adjoint(const NcdeProblem* p, const float* z_out, const float* grad_out, const NcdeGrads* grads,
                 void* workspace, size_t workspace_bytes, void* stream);

/* ---- exact backward of the DISCRETISED solve: what cdeint(..., adjoint=False) + autograd computes in the reference
 * (modules/torchcde/torchcde/solver.py:224 picks torchdiffeq.odeint; autograd then tapes solvers.py:94-119 and
 * fixed_grid.py:6-29 / rk_common.py:106-114).  Instead of a tape, the forward records every stage input
 * ("stage record", [(T-1)*stages][B][H] fp32, caller-owned, ncde_stage_record_bytes() bytes) and the backward
 * transposes the solve stage by stage, re-evaluating f_theta at the recorded inputs.
 *   ncde_forward_record  = ncde_forward + the stage record
 *   ncde_backward        : stage record + dL/dz_out -> dL/dz0, dL/dtheta   (workspace: ncde_workspace_bytes(p, 2)) */
int64_t ncde_stage_record_bytes(const NcdeProblem* p);
int ncde_forward_record(const NcdeProblem* p, float* out, float* stages, void* workspace, size_t workspace_bytes, void* stream);
int ncde_backward(const NcdeProblem* p, const float* stages, const float* grad_out, const NcdeGrads* grads, void* workspace,
                  size_t workspace_bytes, void* stream);

/* Timing helper for benchmarks: runs `iters` back-to-back launches of the dominant kernel of the given
 * pass on `stream`, bracketed by HIP events on that same stream, and returns the mean milliseconds per
 * launch in *ms_per_launch (this call DOES synchronise).  pass = 1: `out` is z_out; pass = 2: `out` is the stage record. */
int ncde_time_kernel(const NcdeProblem* p, int pass, float* out, const float* grad_out, const NcdeGrads* grads,
                     void* workspace, size_t workspace_bytes, void* stream, int iters, float* ms_per_launch);

/* ---- control-path coefficient construction on the GPU (the step right before the hot path) ----------------
 * Default integer time grid, fp32.  x: [B, L, C] raw series (NaN = missing), device pointers.
 * ncde_prepare_linear   replaces torchcde.linear_interpolation_coeffs (interpolation_linear.py:131-180):
 *                       rectilinear_time_index >= 0 -> rectilinear preparation (out [B, 2L-1, C]), else plain
 *                       NaN-filled linear knots (out [B, L, C]).
 * ncde_prepare_cubic    replaces torchcde.natural_cubic_coeffs (interpolation_cubic.py:7-165, 170-190; NaN = missing:
 *                       ends filled from the first/last observation, spline through the observed knots):
 *                       out [B, L-1, 4C] = a|b|2c|3d.
 * kind = NcdeInterp for the workspace query. */
int64_t ncde_prepare_workspace_bytes(int kind, int B, int L, int C);
int ncde_prepare_linear(const float* x, int B, int L, int C, int rectilinear_time_index, float* out, void* stream);
int ncde_prepare_cubic(const float* x, int B, int L, int C, float* out, void* workspace, size_t workspace_bytes, void* stream);
/* The same builders with the observations on a USER time grid (the t= argument of torchcde.linear_interpolation_coeffs /
 * natural_cubic_coeffs: interpolation_linear.py:131-180, interpolation_cubic.py:56-165): t = DEVICE pointer to L strictly
 * increasing fp32 times, NULL = the integer grid.  Linear: the grid enters the interior-gap fill only (ignored by the rectilinear
 * preparation, whose output the caller pairs with a grid of 2L-1 times).  Cubic: non-uniform natural spline. */
int ncde_prepare_linear_grid(const float* x, const float* t, int B, int L, int C, int rectilinear_time_index, float* out, void* stream);
int ncde_prepare_cubic_grid(const float* x, const float* t, int B, int L, int C, float* out, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NCDE_HIP_H */
