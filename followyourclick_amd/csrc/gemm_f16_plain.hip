// f16 (FYC_F16 storage, v_mfma_f32_16x16x32_f16) plain-GEMM instantiations (Linear / 1x1 conv / batched attention GEMMs), all epilogues.
#include "gemm_kernel.h"
namespace fycg {
int run_f16_plain(const GemmP& p, int batch, int cfg, int ns, hipStream_t st) {
  switch (p.epilogue) {
    case FYC_EPI_LINEAR: return dispatch_ns<f16_t, FYC_GEMM_PLAIN, FYC_EPI_LINEAR>(ns, cfg, p, batch, st);
    case FYC_EPI_GEGLU: return dispatch_ns<f16_t, FYC_GEMM_PLAIN, FYC_EPI_GEGLU>(ns, cfg, p, batch, st);
    case FYC_EPI_HEADS: return dispatch_ns<f16_t, FYC_GEMM_PLAIN, FYC_EPI_HEADS>(ns, cfg, p, batch, st);
  }
  FYC_FAIL(-2, "fyc_gemm: bad epilogue %d", p.epilogue);
}
}  // namespace fycg
