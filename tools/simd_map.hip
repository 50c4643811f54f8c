// development probe: which SIMD each wave of a workgroup lands on (HW_REG_HW_ID): waves w and w + 4 of a 512-thread workgroup share a SIMD
//   hipcc --offload-arch=gfx950 -O2 tools/simd_map.hip -o variants/simd_map && variants/simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = hw;
}
int main() {
    unsigned* d; hipMalloc(&d, 4 * 16 * 8); hipMemset(d, 0, 4 * 16 * 8);
    for (int nt : {256, 512, 1024}) {
        hipLaunchKernelGGL(k, dim3(4), dim3(nt), 0, 0, d);
        unsigned h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int b = 0; b < 2; ++b) { printf("threads %d block %d: ", nt, b); for (int w = 0; w < nt / 64; ++w) printf("w%d:simd%u,wave%u,cu%u  ", w, (h[b*16+w] >> 4) & 3, h[b*16+w] & 15, (h[b*16+w] >> 8) & 15); printf("\n"); }
    }
    return 0;
}
