// bf16 instantiations of the register-staged GEMM kernel (gemm_staged.h): tile configs 12 (128x320, 8 waves) and 14 (128x128, 4 waves).
#include "gemm_staged.h"
namespace fycg {
template <int MODE, int EPI>
static int run_cfg(const GemmP& p, int batch, int cfg, hipStream_t st) {
  switch (cfg) {
    case 12: return launch_staged<bf16_t, 128, 320, 2, 4, MODE, EPI>(p, batch, st);
    case 14: return launch_staged<bf16_t, 128, 128, 2, 2, MODE, EPI>(p, batch, st);
  }
  FYC_FAIL(-2, "fyc_gemm: staged tile config %d not built", cfg);
}
int run_bf16_staged(const GemmP& p, int batch, int cfg, hipStream_t st) {
  if (p.mode == FYC_GEMM_CONV3X3) return run_cfg<FYC_GEMM_CONV3X3, FYC_EPI_LINEAR>(p, batch, cfg, st);
  if (p.mode == FYC_GEMM_CONV3X3_UP2) return run_cfg<FYC_GEMM_CONV3X3_UP2, FYC_EPI_LINEAR>(p, batch, cfg, st);
  switch (p.epilogue) {
    case FYC_EPI_LINEAR: return run_cfg<FYC_GEMM_PLAIN, FYC_EPI_LINEAR>(p, batch, cfg, st);
    case FYC_EPI_GEGLU: return run_cfg<FYC_GEMM_PLAIN, FYC_EPI_GEGLU>(p, batch, cfg, st);
    case FYC_EPI_HEADS: return run_cfg<FYC_GEMM_PLAIN, FYC_EPI_HEADS>(p, batch, cfg, st);
  }
  FYC_FAIL(-2, "fyc_gemm: bad epilogue %d", p.epilogue);
}
}  // namespace fycg
