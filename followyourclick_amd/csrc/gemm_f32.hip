// f32 parity-mode instantiations (v_mfma_f32_16x16x4_f32): correctness path, two tile shapes, 2-deep ring.
#include "gemm_kernel.h"
namespace fycg {
template <int MODE, int EPI>
static int run_tile(const GemmP& p, int batch, int cfg, hipStream_t st) {
  return dispatch_cfg<float, MODE, EPI, false>(cfg, 2, p, batch, st);
}
int run_f32(const GemmP& p, int batch, int cfg, hipStream_t st) {
  if (p.mode == FYC_GEMM_CONV3X3) return run_tile<FYC_GEMM_CONV3X3, FYC_EPI_LINEAR>(p, batch, cfg, st);
  if (p.mode == FYC_GEMM_CONV3X3_UP2) return run_tile<FYC_GEMM_CONV3X3_UP2, FYC_EPI_LINEAR>(p, batch, cfg, st);
  if (p.epilogue == FYC_EPI_LINEAR && p.act != FYC_ACT_NONE) return run_tile<FYC_GEMM_PLAIN, EPI_LINEAR_ACT>(p, batch, cfg, st);
  switch (p.epilogue) {
    case FYC_EPI_LINEAR: return run_tile<FYC_GEMM_PLAIN, FYC_EPI_LINEAR>(p, batch, cfg, st);
    case FYC_EPI_GEGLU: return run_tile<FYC_GEMM_PLAIN, FYC_EPI_GEGLU>(p, batch, cfg, st);
    case FYC_EPI_HEADS: return run_tile<FYC_GEMM_PLAIN, FYC_EPI_HEADS>(p, batch, cfg, st);
  }
  FYC_FAIL(-2, "fyc_gemm: bad epilogue %d", p.epilogue);
}
}  // namespace fycg
