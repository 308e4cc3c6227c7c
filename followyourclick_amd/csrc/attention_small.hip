// fused attention instantiations, head dims 8..48
#include "attention_kernel.h"
namespace fyca {
int run_small(const AttnP& p, int qt, hipStream_t st) {
  if (qt == 4) switch (p.d) { FYC_ATTN_CASE(8, 4); FYC_ATTN_CASE(16, 4); FYC_ATTN_CASE(24, 4); FYC_ATTN_CASE(32, 4); FYC_ATTN_CASE(40, 4); FYC_ATTN_CASE(48, 4); }
  else if (qt == 3) switch (p.d) { FYC_ATTN_CASE(8, 3); FYC_ATTN_CASE(16, 3); FYC_ATTN_CASE(24, 3); FYC_ATTN_CASE(32, 3); FYC_ATTN_CASE(40, 3); FYC_ATTN_CASE(48, 3); }
  else switch (p.d) { FYC_ATTN_CASE(8, 2); FYC_ATTN_CASE(16, 2); FYC_ATTN_CASE(24, 2); FYC_ATTN_CASE(32, 2); FYC_ATTN_CASE(40, 2); FYC_ATTN_CASE(48, 2); }
  FYC_FAIL(-2, "fyc_attention: head dim %d not built", p.d);
}
}  // namespace fyca
