// bf16 implicit-GEMM 3x3 convolution instantiations of the ping-pong main loop (gemm_pp_kernel.h).
#include "gemm_pp_kernel.h"
namespace fycg {
int run_pp_conv(const GemmP& p, int cfg, hipStream_t st) {
  if (p.mode == FYC_GEMM_CONV3X3) return dispatch_pp<FYC_GEMM_CONV3X3, FYC_EPI_LINEAR>(cfg, p, st);
  return dispatch_pp<FYC_GEMM_CONV3X3_UP2, FYC_EPI_LINEAR>(cfg, p, st);
}
}  // namespace fycg
