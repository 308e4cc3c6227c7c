// fused attention instantiations, head dims 104..160
#include "attention_kernel.h"
namespace fyca {
int run_large(const AttnP& p, hipStream_t st) {
  switch (p.d) { FYC_ATTN_CASE(104, 2); FYC_ATTN_CASE(112, 2); FYC_ATTN_CASE(120, 2); FYC_ATTN_CASE(128, 2); FYC_ATTN_CASE(136, 2); FYC_ATTN_CASE(144, 2); FYC_ATTN_CASE(152, 2); FYC_ATTN_CASE(160, 2); }
  FYC_FAIL(-2, "fyc_attention: head dim %d not built", p.d);
}
}  // namespace fyca
