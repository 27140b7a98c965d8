// build: hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_mx_probe.hip -o tools/probe/mfma_mx_probe   (run on the GPU box)
// development probe: v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands and unit (e8m0 = 127) block scales:
// is D[i][j] = sum_k A[i][k] B[j][k] when lane (r = lane&15, g = lane>>4) supplies bytes k = 32g..32g+31 of row r?
#include <hip/hip_runtime.h>
#include <hip/hip_fp8.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned char* A, const unsigned char* B, float* D) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    i32x8 a = *reinterpret_cast<const i32x8*>(A + r * 128 + g * 32);
    i32x8 b = *reinterpret_cast<const i32x8*>(B + r * 128 + g * 32);
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int x = 0; x < 4; ++x) D[(4 * g + x) * 16 + r] = acc[x];
}
static float e4m3(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -f : f;
}
int main() {
    unsigned char hA[16 * 128], hB[16 * 128]; float hD[256], ref[256];
    srand(2);
    for (int i = 0; i < 2048; ++i) {
        unsigned char a = rand() & 0xff, b = rand() & 0xff;
        if ((a & 0x7f) == 0x7f) a &= 0xfe;      // no NaN
        if ((b & 0x7f) == 0x7f) b &= 0xfe;
        hA[i] = a; hB[i] = b;
    }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int kk = 0; kk < 128; ++kk) s += (double)e4m3(hA[i * 128 + kk]) * e4m3(hB[j * 128 + kk]); ref[i * 16 + j] = (float)s; }
    unsigned char *dA, *dB; float* dD;
    (void)hipMalloc(&dA, 2048); (void)hipMalloc(&dB, 2048); (void)hipMalloc(&dD, 1024);
    (void)hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    (void)hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    double worst = 0, scale = 0;
    for (int i = 0; i < 256; ++i) { worst = fmax(worst, fabs(hD[i] - ref[i])); scale = fmax(scale, fabs(ref[i])); }
    printf("max |D - ref| = %g at scale %g; D[0][0..3] = %g %g %g %g, ref %g %g %g %g\n", worst, scale, hD[0], hD[1], hD[2], hD[3], ref[0], ref[1], ref[2], ref[3]);
    return 0;
}
