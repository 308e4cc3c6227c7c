// f16 (FYC_F16 storage, v_mfma_f32_16x16x32_f16) plain GEMM with a pointwise activation in the epilogue (conditioning encoders: CLIP MLPs, Resampler feed-forward).
// Off the denoising hot path, so only three tile shapes are built.
#include "gemm_kernel.h"
namespace fycg {
int run_f16_act(const GemmP& p, int batch, int cfg, hipStream_t st) {
  if (!p.wide) return dispatch_cfg<f16_t, FYC_GEMM_PLAIN, EPI_LINEAR_ACT, false>(cfg, 2, p, batch, st);
  if (cfg == 6 || cfg == 5) return launch<f16_t, 128, 320, 2, 4, FYC_GEMM_PLAIN, EPI_LINEAR_ACT, 2, 128, true>(p, batch, st);
  if (cfg == 1 || cfg == 3 || cfg == 7) return launch<f16_t, 128, 128, 2, 2, FYC_GEMM_PLAIN, EPI_LINEAR_ACT, 2, 128, true>(p, batch, st);
  return launch<f16_t, 128, 64, 2, 2, FYC_GEMM_PLAIN, EPI_LINEAR_ACT, 2, 128, true>(p, batch, st);
}
}  // namespace fycg
