// Instantiations of the overlapped-epilogue GEMM / convolution kernel (gemm_ov_kernel.h): tile config 31 = 128x320, LINEAR epilogue.
#include "gemm_ov_kernel.h"
namespace fycg {
int run_ov(const GemmP& p, int cfg, hipStream_t st) {
  if (cfg != 31) FYC_FAIL(-2, "fyc_gemm: overlapped-epilogue tile config %d not built", cfg);
  switch (p.mode) {
    case FYC_GEMM_PLAIN: return launch_ov<128, 320, 2, 4, FYC_GEMM_PLAIN>(p, st);
    case FYC_GEMM_CONV3X3: return launch_ov<128, 320, 2, 4, FYC_GEMM_CONV3X3>(p, st);
    case FYC_GEMM_CONV3X3_UP2: return launch_ov<128, 320, 2, 4, FYC_GEMM_CONV3X3_UP2>(p, st);
  }
  FYC_FAIL(-2, "fyc_gemm: bad mode %d", p.mode);
}
}  // namespace fycg
