// f16 (FYC_F16 storage, v_mfma_f32_16x16x32_f16) implicit-GEMM 3x3 convolution instantiations (stride 1/2 and nearest-2x-upsampled input).
#include "gemm_kernel.h"
namespace fycg {
int run_f16_conv(const GemmP& p, int batch, int cfg, int ns, hipStream_t st) {
  if (p.mode == FYC_GEMM_CONV3X3) return dispatch_ns<f16_t, FYC_GEMM_CONV3X3, FYC_EPI_LINEAR>(ns, cfg, p, batch, st);
  return dispatch_ns<f16_t, FYC_GEMM_CONV3X3_UP2, FYC_EPI_LINEAR>(ns, cfg, p, batch, st);
}
}  // namespace fycg
