// clipmi.hip — error plumbing and the small stand-alone entry points of libclipmi.so.
#include "common.hpp"

namespace clipmi {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int set_err(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

namespace {

// One wave per row: x / ||x||_2, rows with norm < 1e-9 untouched (query-index.py:13-17).
__global__ void __launch_bounds__(256) l2_normalize_rows_kernel(float* x, long long n, int E) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    float* p = x + row * E;
    float ss = 0.f;
    for (int i = lane; i < E; i += 64) { const float v = p[i]; ss += v * v; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
    const float nrm = sqrtf(ss);
    if (nrm < 1e-9f) return;
    for (int i = lane; i < E; i += 64) p[i] = p[i] / nrm;
}

}  // namespace
}  // namespace clipmi

using namespace clipmi;

extern "C" const char* clipmi_last_error(void) { return err_buf(); }
extern "C" int clipmi_abi_version(void) { return CLIPMI_ABI_VERSION; }

extern "C" int clipmi_l2_normalize_rows(float* x_dev, int64_t n, int E, void* stream) {
    if (!x_dev || n < 0 || E < 1) return set_err(CLIPMI_EINVAL, "l2_normalize_rows: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(l2_normalize_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, as_stream(stream), x_dev,
                       (long long)n, E);
    CLIPMI_CHECK_LAUNCH("l2_normalize_rows_kernel");
    return 0;
}
