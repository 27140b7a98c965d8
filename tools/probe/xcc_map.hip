// Which XCD does block b of a 256-block launch land on? (development probe; grid of 512-thread blocks with ~154 KB LDS)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(512) k(unsigned* out) {
    extern __shared__ char smem[];
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = x & 0xf; out[blockIdx.x * 2 + 1] = (unsigned)(__builtin_amdgcn_s_memrealtime() & 0xffffffffu); }
    smem[threadIdx.x] = 1;
    for (volatile int i = 0; i < 20000; ++i) {}
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 157696);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 157696, 0, d);
        hipDeviceSynchronize();
        unsigned h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("rep %d xcc of blocks 0..63:", rep);
        for (int b = 0; b < 64; ++b) printf(" %u", h[2 * b]);
        int bad = 0; for (int b = 8; b < 256; ++b) bad += h[2 * b] != h[2 * (b - 8)];
        printf("\n blocks with xcc(b) != xcc(b-8): %d of 248; start spread %u ticks (100 MHz)\n", bad, h[2 * 255 + 1] - h[1]);
    }
    return 0;
}
