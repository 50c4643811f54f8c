// XCD-cooperative, WEIGHT-STATIONARY output phase of the batch-tiled reverse sweep (round 5; included by ncde_tiled.hip).
//
// Why.  In the large-hidden regime (BASELINE cfg5: H = HH = 128, C = 80, |Wo| = 5.2 MB) a workgroup that owns 16 samples must
// stream ALL of Wo -- twice, once per operand form -- from L2 for every stage: 13 MB per stage and CU, 3.3 GB per stage over the
// chip, and every weight element fetched feeds 16 samples.  The sweep sat at 48 % of the L2 -> CU bandwidth and 62 % matrix-pipe
// occupancy on fp32-input MFMAs with no lever left inside that structure.
//
// What.  The sample tiles of one launch are partitioned into GROUPS of M workgroups (M = 32 at cfg5: the 32 CUs of one XCD, which
// share an L2).  Inside a group every workgroup plays two roles per stage:
//   * OWNER of its own 16-sample tile, exactly as before: hidden layers forward, Butcher bookkeeping, hidden layers backward,
//     records for the output-layer gradient pass;
//   * KEEPER of 1/M of the ROWS of Wo (20 row tiles = 4 state units x 80 channels at cfg5) in both MFMA operand forms (row-major
//     fragments for P = Wo x_L, K-major fragments for Wo^T dP; 2-way split-fp16, packed once per launch by ncde_coop_pack into a
//     per-member, per-wave fragment image of 40 KB), which it applies to the x_L of ALL M sample tiles of the group.  The 8 waves
//     split in two roles: waves 0-3 compute P / tanh / dP of tile i (5 row tiles each) and hand dP over through LDS, waves 4-7
//     form the partial of dL/dx_L of tile i-1 (2 column tiles each).  A wave's 40 fragments (160 registers) are re-read from the
//     image at the start of every stage's keeper loop and stay in registers over its M tiles: keeping them over the whole launch
//     (COOP_PIN) costs more in scratch reloads of the owner phase than the 40 KB L2 read per stage (measured, see COOP_PIN below).
// Per stage: owners publish (x_L as a scaled split-fp16 B-operand image, the cotangent a, dX/dt) -> group barrier -> keepers loop
// over the M tiles: P, tanh, f.dX slice, dP, partial of dL/dx_L over their rows -> publish -> group barrier -> owners sum the M
// partials of their tile in a fixed order.  L2 traffic per stage and CU drops from 13 MB of weights to ~1 MB of activations, the
// transposed product moves from fp32-input MFMA (32 cycles per 16x16x4) to the f16 matrix cores, and no fp32 copy of Wo is read at all.
//
// Arithmetic.  Both products are 2-way split-fp16 (ncde_bf3.h: operands to 2^-24, products to 2^-22) with EXACT power-of-two
// scaling per sample and per launch instead of the range-fault speculation of the forward kernels: x_L(.,s) is scaled by
// sx_s = 2^-e with max_j |x_L[j,s]| sx_s in [1/2, 1), dP(.,s) by sd_s chosen from max_h |a[h,s]| max_c |dX/dt[c,s]| >= max |dP[.,s]|,
// Wo by one factor from max |Wo| (ncde_coop_absmax); results are multiplied back by the exact reciprocals.  Entries far below a
// column's maximum keep an ABSOLUTE error of 2^-36 of that maximum -- below the fp32 rounding of the sums they enter.
//
// Inter-workgroup protocol (MI355X_MICROARCH.md, "inter-workgroup visibility"; cdna_hip_programming.md Guideline 16): payload is
// stored write-through (`sc1` buffer stores) -- or, when the members of the group found themselves on ONE XCD at start-up
// (coop_same_xcd: they exchange HW_REG_XCC_ID), with plain stores into the L2 they share --, every storing wave drains
// (`s_waitcnt vmcnt(0)`), the workgroup barrier-syncs, ONE lane adds 1 to the group's monotonic counter with an agent-scope atomic;
// consumers poll that ONE word with relaxed agent-scope loads
// (+ s_sleep), then read the payload with `sc1` loads (L1 bypassed; the keeper's x_L / dX/dt / a staging goes global -> LDS directly,
// `global_load_lds_dwordx4 ... sc1`, double-buffered one tile ahead).  Correct for ANY workgroup placement; same-XCD placement
// (block b -> XCD b % 8, observed) only makes it faster.  Every spin is bounded: on a timeout the workgroup raises the launch's
// abort word, every workgroup leaves at its next check, and the caller's gradients are poisoned with NaN (never silently wrong).
// All n_tiles workgroups must be co-resident (one per CU: the kernel takes ~157 KB of LDS): the host only selects this path for
// n_tiles <= the device's CU count, and a GPU shared with another process's persistent kernels is outside its contract.
//
// The FORWARD (ncde_fwd_tiled<.., COOP>) uses the same groups, weight images and protocol for its output phase P = Wo x_L: it has no
// transposed product, so all 8 waves play the P role (waves 0-3 on the group's even tiles, 4-7 on the odd ones, two tiles per keeper
// iteration), the exchange carries no partial sums (30 KB per tile: it stays in the XCD's L2), and -- its owner phases being light -- the
// fragments stay in registers for the whole launch (36 of a wave's 40; the last four wait in LDS).
#pragma once
#include "ncde_bf3.h"
#include "ncde_common.h"

#define COOP_NRT 5                      // row tiles of Wo per P-role wave (x 4 waves = 20 per workgroup)
#define COOP_RPM (4 * COOP_NRT)         // row tiles per member
#define COOP_NCH 4                      // K chunks of 32: last hidden width 128
#define COOP_SPIN_LIMIT (1 << 22)       // polls of a group counter before giving up (~1-2 s); the fault-injection flag shortens it

namespace {

typedef unsigned u32x2c __attribute__((ext_vector_type(2)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

// exchange area of one sample tile (floats): [XL split image | a | dX/dt | scales | f.dX slices | M partials of dL/dx_L]
struct CoopDims {
    int H, C, dlast, M, G;
    __host__ __device__ int xl() const { return dlast * 16; }                 // 2 pieces x fp16 = 4 bytes per element
    __host__ __device__ int off_as() const { return xl(); }
    __host__ __device__ int off_dx() const { return off_as() + H * 16; }
    __host__ __device__ int off_sc() const { return off_dx() + C * 16; }      // 4 x 16 floats: sx, 1/(sx sw), sd, 1/(sd sw)
    __host__ __device__ int off_ko() const { return off_sc() + 64; }
    __host__ __device__ int off_part() const { return off_ko() + H * 16; }
    __host__ __device__ long long per_tile() const { return (long long)off_part() + (long long)M * dlast * 16; }
    __host__ __device__ long long per_tile_fwd() const { return (long long)off_part(); }      // the forward exchanges no partials
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t coop_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7ffffffc, 0x00020000);
}
// Payload stores.  `same` (uniform over the group, decided once per launch by coop_same_xcd): every member of the group runs on ONE
// XCD, whose L2 they share -- a plain store is in that L2 once the storing wave's vmcnt has drained, and the readers' sc1 loads
// (L1 bypassed) find it there at L2-hit latency.  Otherwise the store is written through (sc1) so that readers on another XCD see
// it: correct under any placement, at fabric latency.  (byte offsets from the exchange base; 2 GB window)
__device__ __forceinline__ void coop_st16(__amdgpu_buffer_rsrc_t r, float* base, bool same, long long float_off, u32x4 v) {
    if (same) *reinterpret_cast<u32x4*>(base + float_off) = v;
    else __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)(float_off * 4), 0, 16);
}
__device__ __forceinline__ void coop_st4(__amdgpu_buffer_rsrc_t r, float* base, bool same, long long float_off, float v) {
    if (same) base[float_off] = v;
    else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)(float_off * 4), 0, 16);
}
__device__ __forceinline__ u32x4 coop_ld16(__amdgpu_buffer_rsrc_t r, long long float_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)(float_off * 4), 0, 16);
}
__device__ __forceinline__ float coop_ld4(__amdgpu_buffer_rsrc_t r, long long float_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)(float_off * 4), 0, 16));
}
__device__ __forceinline__ f32x4 coop_ld16f(__amdgpu_buffer_rsrc_t r, long long float_off) {
    return __builtin_bit_cast(f32x4, coop_ld16(r, float_off));
}

// power of two s with m s in [1/2, 1) (1 for m = 0 / non-finite), and exact reciprocals
__device__ __forceinline__ float coop_pow2_scale(float m) {
    if (!(m > 0.0f) || !(m < 3.0e38f)) return 1.0f;
    int e = (int)((__float_as_uint(m) >> 23) & 255u) - 126;      // m in [2^(e-1), 2^e)
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    return __uint_as_float((unsigned)(127 - e) << 23);
}
__device__ __forceinline__ float coop_pow2_inv(float s) { return __uint_as_float((254u - ((__float_as_uint(s) >> 23) & 255u)) << 23); }

// sync words: [0 .. G-1] group counters (monotonic within a launch), [G] abort word, [G+1 .. 2G] the groups' start-up counters,
// [2G+1 .. 2G+n_tiles] XCC id + 1 of every workgroup (all zeroed by the host before every launch)
struct CoopSync {
    unsigned* words;
    int G;
    unsigned* status;      // the call's status word (KArgs.coop_status): sticky, survives the per-window reset of `words`
    unsigned spin;         // polls before giving up
};
// giving up: the launch's abort word (every workgroup leaves at its next check) AND the call's status word (the cooperative launches of
// the later time windows return at once; the per-workgroup kernels enqueued behind them re-execute the pass)
__device__ __forceinline__ void coop_give_up(const CoopSync& sy) {
    __hip_atomic_store(sy.words + sy.G, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(sy.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__host__ __device__ inline int coop_sync_words(int G, int n_tiles) { return 2 * G + 1 + n_tiles; }
// Every storing wave has drained (s_waitcnt vmcnt(0)) and the workgroup has barrier-synced before this is called by thread 0.
__device__ __forceinline__ void coop_arrive(const CoopSync& sy, int g) {
    __hip_atomic_fetch_add(sy.words + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// thread 0 polls; returns false on abort / timeout (uniform over the workgroup through `flag_lds`)
__device__ __forceinline__ bool coop_wait(const CoopSync& sy, int g, unsigned target, int* flag_lds, int tid) {
    if (tid == 0) {
        int ok = 1;
        unsigned spins = 0;
        for (;;) {
            const unsigned c = __hip_atomic_load(sy.words + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c >= target) break;
            if ((spins & 63u) == 63u && __hip_atomic_load(sy.words + sy.G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
            if (++spins > sy.spin) {
                coop_give_up(sy);
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        *flag_lds = ok;
    }
    __syncthreads();
    const bool ok = *flag_lds != 0;
    __syncthreads();      // (flag_lds is rewritten by the next wait)
    return ok;
}

// Do all M members of this workgroup's group run on the same XCD (= share an L2)?  Every member publishes its XCC id with an
// agent-scope store, arrives on the group's start-up counter, waits for the M arrivals and reads the M ids back: every member sees
// the same M words, so the answer is uniform over the group.  -1 on timeout / abort.
__device__ __forceinline__ int coop_same_xcd(const CoopSync& sy, int g, int member, int M, int n_tiles_per_launch, int* flag_lds, int tid) {
    (void)n_tiles_per_launch;
    unsigned* ids = sy.words + 2 * sy.G + 1;
    if (tid == 0) {
        const unsigned me = (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) + 1u;      // HW_REG_XCC_ID[3:0] + 1
        int res = 1;
        // an earlier launch of this call gave up: nothing cooperative runs any more (the per-workgroup kernels behind redo the pass)
        if (__hip_atomic_load(sy.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) res = -1;
        if (res >= 0) {
            __hip_atomic_store(ids + g + sy.G * member, me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(sy.words + sy.G + 1 + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned spins = 0;
        while (res >= 0) {
            if (__hip_atomic_load(sy.words + sy.G + 1 + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)M) break;
            if ((spins & 63u) == 63u && __hip_atomic_load(sy.words + sy.G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { res = -1; break; }
            if (++spins > sy.spin) {
                coop_give_up(sy);
                res = -1;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        if (res >= 0)
            for (int mm = 0; mm < M; ++mm)
                if (__hip_atomic_load(ids + g + sy.G * mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != me) res = 0;
        *flag_lds = res;
    }
    __syncthreads();
    const int res = *flag_lds;
    __syncthreads();
    return res;
}

// The keeper's resident weights: 20 row tiles of Wo per workgroup, 160 registers per wave.  The workgroup has 8 waves, two per SIMD:
//   * waves 0..3 ("P role") hold ROW-major fragments of 5 row tiles each and compute P = Wo x_L, tanh, the f.dX slice and dP;
//   * waves 4..7 ("T role") hold K-major fragments of 2 COLUMN tiles of the 128 hidden units each, over all 20 row tiles (10 K-pairs),
//     and compute Wo^T dP one sample tile BEHIND the P role, from the dP the P role left in LDS.
// A P wave and a T wave share each SIMD: the tanh / split / LDS work of one runs under the MFMAs of the other.
struct CoopWeights {
    u32x4 f[40];      // P role: [row tile q][K chunk c][piece] = f[(q * 4 + c) * 2 + piece];  T role: [pair][column tile][piece] = f[(pr * 2 + ci) * 2 + piece]
};

// words of the packed images per workgroup: [P role: 4 waves x 40 fragments][T role: 4 waves x 40 fragments], 64 lanes x 4 words each
__host__ __device__ constexpr int coop_p_words() { return 4 * 40 * 64 * 4; }
__host__ __device__ constexpr int coop_t_words() { return 4 * 40 * 64 * 4; }

// Fragments [0, COOP_PIN) stay in registers for the whole launch; the other 40 - COOP_PIN are re-read from the packed image at the start
// of every keeper phase.  Measured at cfg5 (s_memtime phase counters, cycles per stage of 8 waves): all 40 pinned --
// the owner phases run out of registers, the compiler parks nine fragments in scratch and reloads them INSIDE the keeper loops (every
// scratch reload is an in-order vmcnt wait): 292k cycles in those loops, 478k per stage; 30 pinned + 10 re-read: the same nine reloads;
// all 40 re-read (the shipped setting): clean loops, 168k, but 10 MB per XCD and stage do not fit the 4 MB L2 and come from the
// Infinity Cache / HBM: +40k cycles waiting for them, 410k per stage (the per-workgroup sweep: 469k).
#ifndef COOP_PIN
#define COOP_PIN 0
#endif
template <int K0, int K1>
__device__ __forceinline__ void coop_load_weights(CoopWeights& w, const unsigned* img, int member, int wave, int lane) {
    const unsigned* p = img + (long long)member * (coop_p_words() + coop_t_words()) + wave * (40 * 64 * 4);      // (waves 4..7: behind the P images)
#pragma unroll
    for (int k = K0; k < K1; ++k) w.f[k] = *reinterpret_cast<const u32x4*>(p + (k * 64 + lane) * 4);
}

// The bias of the member's 20 row tiles as the P-role waves read it: dst[((rw * 5 + i) * 4 + lk) * 4 + r] = bo[row] for row tile
// rw * 5 + i of the member, state unit 4 hb + lk, channel 4 cq + r.  (LDS, not a global load in the keeper loop: `vmcnt` is in order, so
// a global load there waits for the staging loads of the NEXT tile issued just before it -- an L2 round trip per bias vector.)
__device__ __forceinline__ void coop_fill_bias(const float* bo, int C, int member, float* dst, int tid) {
    if (tid < 4 * COOP_NRT * 16) {
        const int r = tid & 3, lk = (tid >> 2) & 3, i = (tid >> 4) % COOP_NRT, rw = tid / (16 * COOP_NRT);
        const int ncq = C >> 2, t0 = member * COOP_RPM + rw * COOP_NRT;
        dst[tid] = bo[(4 * (t0 / ncq) + lk) * C + 4 * (t0 % ncq + i) + r];
    }
}

// 8 (or 4) fp32 values -> split-fp16 pieces (no range tracking: the caller has scaled them into [-1, 1])
__device__ __forceinline__ void coop_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
    const f32x2 x = {x0, x1};
    const h16x2 h = __builtin_convertvector(x, h16x2);
    f32x2 r;
    r[0] = __builtin_fmaf((float)h[0], -NCDE_H2_SCALE, x0 * NCDE_H2_SCALE);      // exact: (x - h1) * 2^11
    r[1] = __builtin_fmaf((float)h[1], -NCDE_H2_SCALE, x1 * NCDE_H2_SCALE);
    const h16x2 l = __builtin_convertvector(r, h16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

}  // namespace

#ifndef NCDE_COOP_NO_KERNELS      // (a second translation unit -- ncde_dwo2.hip -- includes this header for the helpers only)
// ------------------------------------------------------------------------------------------------------------------------------
// pack kernels (once per call: the parameters change every training step)
// ------------------------------------------------------------------------------------------------------------------------------
// out[0] = max |W| as float bits (non-negative floats order like unsigned integers); out must be zeroed first
__global__ __launch_bounds__(256) void ncde_coop_absmax(const float* __restrict__ W, long long n, unsigned* out) {
    float m = 0.0f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = fmaxf(m, fabsf(W[i]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// Row of Wo behind (row tile rt, tile row li): tiles are (4 state units) x (4 channels), rt = hb * (C/4) + cq  (as ncde_pack_panels)
__device__ __forceinline__ long long coop_row(int rt, int li, int C) {
    const int ncq = C >> 2, hb = rt / ncq, cq = rt - hb * ncq;
    return (long long)(4 * hb + (li >> 2)) * C + 4 * cq + (li & 3);
}

// One thread = one (member, wave, fragment, lane): 8 (4) weights, scaled by sw = 2^-e (max |Wo| sw in [1/2, 1)), split into two fp16
// pieces.  scale[0] = sw, scale[1] = 1 / sw (written by block 0 for the sweep to read).
__global__ __launch_bounds__(256) void ncde_coop_pack(const float* __restrict__ W, const unsigned* __restrict__ absmax, unsigned* __restrict__ img,
                                                      float* __restrict__ scale, int C, int dlast, int M) {
    const float sw = coop_pow2_scale(__uint_as_float(*absmax));
    if (blockIdx.x == 0 && threadIdx.x == 0) { scale[0] = sw; scale[1] = coop_pow2_inv(sw); }
    constexpr int PF = COOP_NRT * COOP_NCH, TF = (COOP_RPM / 2) * 2;      // fragments (both pieces handled by one thread) per wave
    const long long n = (long long)M * 4 * (PF + TF) * 64;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < n; v += (long long)gridDim.x * 256) {
        const int lane = (int)(v & 63);
        long long f = v >> 6;
        const int frag = (int)(f % (PF + TF));
        f /= (PF + TF);
        const int wave = (int)(f & 3), member = (int)(f >> 2);
        const int li = lane & 15, kg = lane >> 4;
        const int rt0 = member * COOP_RPM + wave * COOP_NRT;
        unsigned* base = img + (long long)member * (coop_p_words() + coop_t_words());
        float x[8];
        if (frag < PF) {
            const int q = frag / COOP_NCH, c = frag - q * COOP_NCH;
            const float* src = W + coop_row(rt0 + q, li, C) * dlast + 32 * c + 8 * kg;
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = src[e] * sw;
            unsigned h[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) coop_split2(x[2 * e], x[2 * e + 1], h[e], l[e]);
            unsigned* dst = base + wave * (PF * 2 * 64 * 4) + ((frag * 2) * 64 + lane) * 4;
            *reinterpret_cast<u32x4*>(dst) = (u32x4){h[0], h[1], h[2], h[3]};
            *reinterpret_cast<u32x4*>(dst + 256) = (u32x4){l[0], l[1], l[2], l[3]};
        } else {      // K-major: rows (tile 2 pr + (e >> 2), row 4 kg + (e & 3)) of the workgroup's pair pr, column 16 ct + li, ct = 2 wave + ci
            const int ft = frag - PF, pr = ft >> 1, ci = ft & 1, ct = 2 * wave + ci;
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = W[coop_row(member * COOP_RPM + 2 * pr + (e >> 2), 4 * kg + (e & 3), C) * dlast + 16 * ct + li] * sw;
            unsigned h[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) coop_split2(x[2 * e], x[2 * e + 1], h[e], l[e]);
            unsigned* dst = base + coop_p_words() + wave * (TF * 2 * 64 * 4) + ((ft * 2) * 64 + lane) * 4;
            *reinterpret_cast<u32x4*>(dst) = (u32x4){h[0], h[1], h[2], h[3]};
            *reinterpret_cast<u32x4*>(dst + 256) = (u32x4){l[0], l[1], l[2], l[3]};
        }
    }
}
#endif  // NCDE_COOP_NO_KERNELS
