// resize.hip — the geometry of the CLIP transform on the device (build-index.py:48): Pillow's bicubic
// resize of the shorter side to n_px + centre crop, bit for bit, for 8-bit RGB images that travel at full size.
//
// Pillow (the reference's dependency for this step; its published algorithm restated, not its code): 8-bit images are
// resampled in two passes, horizontal then vertical, each output a dot product of up to ksize taps with integer
// coefficients kk = trunc(w * 2^22 +- 0.5), accumulated in int32 from 2^21 and shifted right by 22, clipped to 0..255; the
// intermediate image between the passes is 8-bit. The coefficients depend on the sizes only; the HOST computes them
// (decode_worker.py: coeffs_window, the same float64 operations as Pillow's precompute_coeffs) for the n_px outputs per
// axis that survive the centre crop. These kernels do the integer part.
#include "common.hpp"

namespace clipmi {
namespace {

struct ResizeJob {             // mirrors clipmi_resize_job (include/clipmi.h)
    long long src_off;
    int w, h;
    int r0, nrows;
    int out_index;
    int need_h, need_v;
    int left, top;
    int hk, vk;
    long long hcoef_off, vcoef_off;
    long long tmp_off;
};

constexpr int RZ_PREC = 22;

__device__ __forceinline__ unsigned char rz_clip8(int v) {
    v >>= RZ_PREC;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: tmp[row][x][c] for the job's needed source rows and the n_px window columns.
// grid (row blocks, jobs); one thread per (row, x); per-axis coefficient block = [n_px] xmin | [n_px] cnt | [n_px][k] kk
__global__ void __launch_bounds__(256) resize_h_kernel(const unsigned char* __restrict__ raw, const ResizeJob* __restrict__ jobs,
                                                       const int* __restrict__ coef, int n_px, unsigned char* __restrict__ scratch) {
    const ResizeJob j = jobs[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int row = idx / n_px, x = idx - row * n_px;
    if (row >= j.nrows) return;
    const unsigned char* src = raw + j.src_off + (size_t)(j.r0 + row) * j.w * 3;
    unsigned char* dst = scratch + j.tmp_off + ((size_t)row * n_px + x) * 3;
    if (!j.need_h) {
        const unsigned char* p = src + (size_t)(j.left + x) * 3;
        dst[0] = p[0]; dst[1] = p[1]; dst[2] = p[2];
        return;
    }
    const int* cf = coef + j.hcoef_off;
    const int xmin = cf[x], cnt = cf[n_px + x];
    const int* kk = cf + 2 * n_px + (size_t)x * j.hk;
    int a0 = 1 << (RZ_PREC - 1), a1 = a0, a2 = a0;
    const unsigned char* p = src + (size_t)xmin * 3;
    for (int k = 0; k < cnt; ++k) {
        const int c = kk[k];
        a0 += p[3 * k] * c; a1 += p[3 * k + 1] * c; a2 += p[3 * k + 2] * c;
    }
    dst[0] = rz_clip8(a0); dst[1] = rz_clip8(a1); dst[2] = rz_clip8(a2);
}

// vertical pass + planar store: out[out_index][c][y][x]
__global__ void __launch_bounds__(256) resize_v_kernel(const ResizeJob* __restrict__ jobs, const int* __restrict__ coef, int n_px,
                                                       const unsigned char* __restrict__ scratch, unsigned char* __restrict__ out) {
    const ResizeJob j = jobs[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int y = idx / n_px, x = idx - y * n_px;
    if (y >= n_px) return;
    const unsigned char* tmp = scratch + j.tmp_off;
    unsigned char* o = out + (size_t)j.out_index * 3 * n_px * n_px + (size_t)y * n_px + x;
    const size_t plane = (size_t)n_px * n_px;
    if (!j.need_v) {
        const unsigned char* p = tmp + ((size_t)(j.top - j.r0 + y) * n_px + x) * 3;
        o[0] = p[0]; o[plane] = p[1]; o[2 * plane] = p[2];
        return;
    }
    const int* cf = coef + j.vcoef_off;
    const int ymin = cf[y], cnt = cf[n_px + y];
    const int* kk = cf + 2 * n_px + (size_t)y * j.vk;
    int a0 = 1 << (RZ_PREC - 1), a1 = a0, a2 = a0;
    const unsigned char* p = tmp + ((size_t)(ymin - j.r0) * n_px + x) * 3;
    const size_t rs = (size_t)n_px * 3;
    for (int k = 0; k < cnt; ++k) {
        const int c = kk[k];
        a0 += p[k * rs] * c; a1 += p[k * rs + 1] * c; a2 += p[k * rs + 2] * c;
    }
    o[0] = rz_clip8(a0); o[plane] = rz_clip8(a1); o[2 * plane] = rz_clip8(a2);
}

}  // namespace
}  // namespace clipmi

using namespace clipmi;

extern "C" int clipmi_resize_crop_rgb8(const void* raw_dev, const void* jobs_dev, int njobs, int max_rows, const int32_t* coef_dev,
                                       int n_px, void* out_dev, void* scratch_dev, void* stream) {
    static_assert(sizeof(ResizeJob) == 80, "clipmi_resize_job layout");
    if (njobs == 0) return 0;
    if (!raw_dev || !jobs_dev || !coef_dev || !out_dev || !scratch_dev || njobs < 0 || n_px < 1 || n_px > 4096 || max_rows < 1)
        return set_err(CLIPMI_EINVAL, "resize_crop_rgb8: bad arguments");
    hipStream_t st = as_stream(stream);
    const long long per_h = (long long)max_rows * n_px, per_v = (long long)n_px * n_px;
    hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)((per_h + 255) / 256), (unsigned)njobs), dim3(256), 0, st,
                       static_cast<const unsigned char*>(raw_dev), static_cast<const ResizeJob*>(jobs_dev), coef_dev, n_px,
                       static_cast<unsigned char*>(scratch_dev));
    CLIPMI_CHECK_LAUNCH("resize_h_kernel");
    hipLaunchKernelGGL(resize_v_kernel, dim3((unsigned)((per_v + 255) / 256), (unsigned)njobs), dim3(256), 0, st,
                       static_cast<const ResizeJob*>(jobs_dev), coef_dev, n_px, static_cast<const unsigned char*>(scratch_dev),
                       static_cast<unsigned char*>(out_dev));
    CLIPMI_CHECK_LAUNCH("resize_v_kernel");
    return 0;
}
