// development probe (round 6): do the matrix pipe and the VALU of a SIMD work at the same time for TWO waves of one SIMD?
// One workgroup of 512 threads = 8 waves, waves w and w + 4 share a SIMD (tools/simd_map.hip).  Mode bits per wave half:
// waves 0..3 run `lo`, waves 4..7 run `hi`: 0 idle, 1 MFMA stream (independent accumulators), 2 VALU stream (independent FMAs),
// 3 transcendental stream.  Prints cycles per 1000 instructions of each stream alone and together.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o variants/mfma_valu_overlap && variants/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(512) void probe(int lo, int hi, int iters, unsigned long long* out, float* sink) {
    const int wave = threadIdx.x >> 6, mode = wave < 4 ? lo : hi;
    f32x4 acc[8];
    float v[16];
    h16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (i + 1)); }
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 16; ++i) v[i] = 0.001f * (threadIdx.x + i);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (mode == 1) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    } else if (mode == 2) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
    } else if (mode == 4) {      // one wave, interleaved: per MFMA two independent FMAs
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
                v[2 * i] = __builtin_fmaf(v[2 * i], 1.0001f, 0.5f);
                v[2 * i + 1] = __builtin_fmaf(v[2 * i + 1], 1.0001f, 0.5f);
            }
    } else if (mode == 5) {      // one wave, interleaved: per MFMA four independent FMAs
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[(4 * i + k) & 15] = __builtin_fmaf(v[(4 * i + k) & 15], 1.0001f, 0.5f);
            }
    } else if (mode == 6) {      // 32x32x16 f16 stream (four independent accumulators)
        f32x16 big[4];
        for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) big[i][k] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) big[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, big[i & 3], 0, 0, 0);
        for (int i = 0; i < 4; ++i) acc[i][0] += big[i][0] + big[i][15];
    } else if (mode == 7) {      // one wave: per 32x32x16 MFMA four independent FMAs
        f32x16 big[4];
        for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) big[i][k] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                big[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, big[i & 3], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[(4 * i + k) & 15] = __builtin_fmaf(v[(4 * i + k) & 15], 1.0001f, 0.5f);
            }
        for (int i = 0; i < 4; ++i) acc[i][0] += big[i][0] + big[i][15];
    } else if (mode == 8) {      // packed fp32 FMA stream (v_pk_fma_f32), eight independent pairs
        f32x2 pv[8];
        for (int i = 0; i < 8; ++i) pv[i] = (f32x2){v[2 * i], v[2 * i + 1]};
        const f32x2 k1 = {1.0001f, 1.0002f}, k2 = {0.5f, 0.25f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) pv[i & 7] = __builtin_elementwise_fma(pv[i & 7], k1, k2);
        for (int i = 0; i < 8; ++i) { v[2 * i] = pv[i][0]; v[2 * i + 1] = pv[i][1]; }
    } else if (mode == 9) {      // fp32-input MFMA 16x16x4 stream
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[i], v[i + 8], acc[i], 0, 0, 0);
    } else if (mode == 3) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 12345.678f) sink[threadIdx.x] = s;
}

int main() {
    unsigned long long* out; float* sink;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&sink, 4096));
    const int iters = 2000;
    const char* nm[10] = {"idle", "mfma 16x16x32 f16", "valu fma", "transcendental exp2", "1 mfma + 2 fma", "1 mfma + 4 fma", "mfma 32x32x16 f16", "1 mfma32 + 4 fma", "v_pk_fma_f32", "mfma 16x16x4 f32"};
    const int per[10] = {0, 8, 16, 16, 8, 8, 8, 8, 16, 8};      // (modes 4, 5: per MFMA group)
    printf("cycles per instruction of each stream (one workgroup, waves w / w + 4 share a SIMD)\n%-22s %-22s %10s %10s\n", "waves 0-3", "waves 4-7", "cyc/instr", "cyc/instr");
    const int cases[][2] = {{1, 0}, {2, 0}, {3, 0}, {1, 1}, {2, 2}, {1, 2}, {2, 1}, {1, 3}, {3, 2}, {4, 0}, {5, 0}, {4, 4}, {5, 5}, {6, 0}, {6, 2}, {2, 6}, {7, 0}, {7, 7}, {8, 0}, {8, 8}, {1, 8}, {9, 0}, {9, 2}};
    for (auto& c : cases) {
        CK(hipMemset(out, 0, 64));
        hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, c[0], c[1], iters, out, sink);
        CK(hipDeviceSynchronize());
        unsigned long long h[8]; CK(hipMemcpy(h, out, 64, hipMemcpyDeviceToHost));
        const double a = per[c[0]] ? (double)h[0] / (iters * per[c[0]]) : 0.0, b = per[c[1]] ? (double)h[4] / (iters * per[c[1]]) : 0.0;
        printf("%-22s %-22s %10.2f %10.2f\n", nm[c[0]], nm[c[1]], a, b);
    }
    return 0;
}
