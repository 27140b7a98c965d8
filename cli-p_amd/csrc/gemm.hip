// gemm.hip — instantiation and launch of the bf16 NT GEMM (see gemm.hpp).
#include "gemm.hpp"

namespace clipmi {

template <int EPI>
static int launch_epi(const GemmArgs& g, hipStream_t st) {
    const int grid = (g.N / GEMM_BN) * ((g.M + GEMM_BM - 1) / GEMM_BM);
    hipLaunchKernelGGL(gemm_bf16_nt_kernel<EPI>, dim3(grid), dim3(256), GEMM_LDS_BYTES, st, g);
    CLIPMI_CHECK_LAUNCH("gemm_bf16_nt_kernel");
    return 0;
}

// Measurement probe (bench.py roofline): while active on this thread, every launch of the GEMM
// with epilogue `epi` is bracketed by a pair of HIP events on its own stream.
GemmProbe& gemm_probe() {
    static thread_local GemmProbe p;
    return p;
}

static int launch_gemm_inner(const GemmArgs& g, int epi, hipStream_t st);

int launch_gemm(const GemmArgs& g, int epi, hipStream_t st) {
    GemmProbe& p = gemm_probe();
    const bool hit = p.active && p.epi == epi && p.n < GemmProbe::MAX;
    if (hit) (void)hipEventRecord(p.ev[2 * p.n], st);
    const int rc = launch_gemm_inner(g, epi, st);
    if (hit) { (void)hipEventRecord(p.ev[2 * p.n + 1], st); ++p.n; }
    return rc;
}

static int launch_gemm_inner(const GemmArgs& g, int epi, hipStream_t st) {
    if (g.M < 1 || g.N < 1 || g.K < 1 || g.N % GEMM_BN != 0 || g.K % GEMM_BK != 0)
        return set_err(CLIPMI_EINVAL, "gemm: M=%d N=%d K=%d (need N %% 128 == 0, K %% 64 == 0)", g.M, g.N, g.K);
    if (!g.A || !g.W || !g.out) return set_err(CLIPMI_EINVAL, "gemm: NULL pointer");
    switch (epi) {
        case EPI_BIAS_BF16: return launch_epi<EPI_BIAS_BF16>(g, st);
        case EPI_BIAS_QGELU_BF16: return launch_epi<EPI_BIAS_QGELU_BF16>(g, st);
        case EPI_BIAS_RESID_F32: return launch_epi<EPI_BIAS_RESID_F32>(g, st);
        case EPI_F32: return launch_epi<EPI_F32>(g, st);
        case EPI_PATCH_F32: return launch_epi<EPI_PATCH_F32>(g, st);
    }
    return set_err(CLIPMI_EINVAL, "gemm: unknown epilogue %d", epi);
}

}  // namespace clipmi

using namespace clipmi;

extern "C" int clipmi_dbg_gemm_bf16(const void* a_dev, const void* w_dev, const float* bias_dev, void* out_dev, int M,
                                    int N, int K, int epi, void* stream) {
    if (epi < 0 || epi > 3) return set_err(CLIPMI_EINVAL, "dbg_gemm: epi %d", epi);
    GemmArgs g{};
    g.A = static_cast<const unsigned short*>(a_dev);
    g.W = static_cast<const unsigned short*>(w_dev);
    g.bias = bias_dev;
    g.out = out_dev;
    g.M = M; g.N = N; g.K = K;
    return launch_gemm(g, epi, as_stream(stream));
}
