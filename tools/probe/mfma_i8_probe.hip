// build: hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_i8_probe.hip -o tools/probe/mfma_i8_probe   (run on the GPU box)
// development probe: operand layout of v_mfma_i32_16x16x64_i8 (gfx950). D[i][j] = sum_k A[i][k] B[j][k]?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const signed char* A, const signed char* B, int* D) {
    // assumption under test: lane (r = lane&15, g = lane>>4) holds 16 consecutive k = 16g..16g+15 of row r
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    i32x4 a = *reinterpret_cast<const i32x4*>(A + r * 64 + g * 16);
    i32x4 b = *reinterpret_cast<const i32x4*>(B + r * 64 + g * 16);
    i32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc, 0, 0, 0);
    // assumption: acc[x] = D[4g + x][r]  (rows from the first operand, column from the second)
    for (int x = 0; x < 4; ++x) D[(4 * g + x) * 16 + r] = acc[x];
}
int main() {
    signed char hA[16 * 64], hB[16 * 64]; int hD[256], ref[256];
    srand(1);
    for (int i = 0; i < 1024; ++i) { hA[i] = rand() % 255 - 127; hB[i] = rand() % 255 - 127; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { int s = 0; for (int kk = 0; kk < 64; ++kk) s += hA[i * 64 + kk] * hB[j * 64 + kk]; ref[i * 16 + j] = s; }
    signed char *dA, *dB; int* dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0, badT = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { bad += hD[i * 16 + j] != ref[i * 16 + j]; badT += hD[i * 16 + j] != ref[j * 16 + i]; }
    printf("mismatches as D[i][j]=A_i.B_j: %d ; transposed: %d\n", bad, badT);
    return 0;
}
