// development / measurement (not part of the product path): round-trip time of a payload + flag exchange between TWO workgroups on
// different CUs, with the primitives csrc/ncde_coop.h uses (agent-scope atomics; plain stores + sc1 loads inside one XCD, sc1
// write-through stores across XCDs).  It answers VERDICT round 5 item 8 with a number: what one exchange per stage would cost if a
// 16-sample tile of cfg2 (stage: 1.28 us forward, 4.4 us adjoint) were split over two workgroups.
//   hipcc --offload-arch=gfx950 -O2 tools/xcd_pingpong.hip -o variants/xcd_pingpong && variants/xcd_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7ffffffc, 0x00020000);
}

// Workgroups `a` and `b` of the grid play; everyone else exits.  One round: a writes `words` 16-byte pieces (256 threads), drains,
// raises its flag; b sees the flag, reads the payload (sc1), writes its own payload back, drains, raises its flag; a reads it.
// out[0] = cycles of `rounds` round trips as seen by a, out[1] / out[2] = XCC id of a / b, out[3] = checksum (keeps the loads alive).
__global__ __launch_bounds__(256) void pingpong(unsigned* flags, unsigned* buf, unsigned long long* out, int a, int b, int pieces, int rounds, int write_through) {
    const int me = blockIdx.x == a ? 0 : (blockIdx.x == b ? 1 : -1);
    if (me < 0) return;
    const int tid = threadIdx.x;
    __shared__ unsigned seen;
    const __amdgpu_buffer_rsrc_t r = rsrc(buf);
    unsigned* mine = buf + me * 65536;                       // (256 KB apart)
    const int other_off = (1 - me) * 65536 * 4;              // byte offset of the partner's payload
    const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
    unsigned acc = 0;
    unsigned long long t0 = 0;
    for (int it = 0; it <= rounds; ++it) {
        if (it == 1 && tid == 0) t0 = __builtin_readcyclecounter();      // (round 0 warms up)
        for (int half = 0; half < 2; ++half) {
            if (half == me) {      // my turn to send
                for (int e = tid; e < pieces; e += 256) {
                    const u32x4 v = (u32x4){(unsigned)it, (unsigned)e, acc, 7u};
                    if (write_through) __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)((mine - buf) * 4 + e * 16), 0, 16);      // sc1
                    else *reinterpret_cast<u32x4*>(mine + e * 4) = v;
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(flags + me * 64, (unsigned)(2 * it + half + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {               // wait for the partner, then read what it sent
                if (tid == 0) {
                    unsigned spins = 0;
                    while (__hip_atomic_load(flags + (1 - me) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(2 * it + half + 1) && ++spins < (1u << 24)) __builtin_amdgcn_s_sleep(1);
                    seen = spins;
                }
                __syncthreads();
                for (int e = tid; e < pieces; e += 256) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, other_off + e * 16, 0, 16);      // sc1: L1 bypassed
                    acc += v[0] + v[1];
                }
            }
        }
    }
    if (me == 0 && tid == 0) out[0] = __builtin_readcyclecounter() - t0;
    if (tid == 0) out[1 + me] = xcc;
    atomicAdd(reinterpret_cast<unsigned*>(out + 3), acc + seen);
}

int main() {
    unsigned *flags, *buf;
    unsigned long long* out;
    CK(hipMalloc(&flags, 4096)); CK(hipMalloc(&buf, 2 * 65536 * 4)); CK(hipMalloc(&out, 64));
    int clock_khz = 0;
    CK(hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0));
    const int rounds = 2000;
    printf("round trip of a payload + flag exchange between two workgroups (256 threads each, %d rounds, shader clock %.2f GHz)\n", rounds, clock_khz * 1e-6);
    printf("%-28s %-14s %10s %12s %12s\n", "placement (XCC ids)", "stores", "payload", "counter/rt", "us/rt");
    // workgroups are dealt round-robin over the 8 XCDs: 0 and 8 share an XCD, 0 and 1 do not (the kernel reports the ids it saw)
    struct Case { int a, b, wt; } cases[] = {{0, 8, 0}, {0, 8, 1}, {0, 1, 1}, {0, 4, 1}};
    for (const Case& c : cases)
        for (int bytes : {0, 2048, 8192, 32768}) {
            CK(hipMemset(flags, 0, 4096)); CK(hipMemset(out, 0, 64));
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(pingpong, dim3(16), dim3(256), 0, 0, flags, buf, out, c.a, c.b, bytes / 16, rounds, c.wt);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h[4];
            CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
            char where[64];
            snprintf(where, sizeof(where), "blocks %d,%d (xcc %llu,%llu)", c.a, c.b, h[1], h[2]);
            const double cyc = (double)h[0] / rounds;
            printf("%-28s %-14s %8d B %12.0f %12.3f\n", where, c.wt ? "sc1 (through)" : "plain (L2)", bytes, cyc, ms * 1e3 / (rounds + 1));      // (us: HIP events around the launch)
        }
    return 0;
}
