// fused attention instantiations, head dims 56..96
#include "attention_kernel.h"
namespace fyca {
int run_medium(const AttnP& p, int qt, hipStream_t st) {
  if (qt == 4 && p.d <= 80) switch (p.d) { FYC_ATTN_CASE(56, 4); FYC_ATTN_CASE(64, 4); FYC_ATTN_CASE(72, 4); FYC_ATTN_CASE(80, 4); }
  else if (qt == 3 && p.d <= 80) switch (p.d) { FYC_ATTN_CASE(56, 3); FYC_ATTN_CASE(64, 3); FYC_ATTN_CASE(72, 3); FYC_ATTN_CASE(80, 3); }
  else switch (p.d) { FYC_ATTN_CASE(56, 2); FYC_ATTN_CASE(64, 2); FYC_ATTN_CASE(72, 2); FYC_ATTN_CASE(80, 2); FYC_ATTN_CASE(88, 2); FYC_ATTN_CASE(96, 2); }
  FYC_FAIL(-2, "fyc_attention: head dim %d not built", p.d);
}
}  // namespace fyca
