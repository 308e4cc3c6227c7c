// bf16 plain-GEMM instantiations of the ping-pong main loop (gemm_pp_kernel.h), all epilogues.
#include "gemm_pp_kernel.h"
namespace fycg {
int run_pp_plain(const GemmP& p, int cfg, hipStream_t st) {
  switch (p.epilogue) {
    case FYC_EPI_LINEAR: return dispatch_pp<FYC_GEMM_PLAIN, FYC_EPI_LINEAR>(cfg, p, st);
    case FYC_EPI_GEGLU: return dispatch_pp<FYC_GEMM_PLAIN, FYC_EPI_GEGLU>(cfg, p, st);
    case FYC_EPI_HEADS: return dispatch_pp<FYC_GEMM_PLAIN, FYC_EPI_HEADS>(cfg, p, st);
  }
  FYC_FAIL(-2, "fyc_gemm: bad epilogue %d", p.epilogue);
}
}  // namespace fycg
