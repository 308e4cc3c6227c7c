// Head-dim groups of the fused attention (kernel template: attention_kernel.h).  Each attention_*.hip instantiates ONE of these for ONE
// element type.
#pragma once
#include "attention_kernel.h"
namespace fyca {
// head dims 8..48
template <typename T> int run_small(const AttnP& p, int qt, hipStream_t st) {
  if (qt == 4) switch (p.d) { FYC_ATTN_CASE(8, 4); FYC_ATTN_CASE(16, 4); FYC_ATTN_CASE(24, 4); FYC_ATTN_CASE(32, 4); FYC_ATTN_CASE(40, 4); FYC_ATTN_CASE(48, 4); }
  else if (qt == 3) switch (p.d) { FYC_ATTN_CASE(8, 3); FYC_ATTN_CASE(16, 3); FYC_ATTN_CASE(24, 3); FYC_ATTN_CASE(32, 3); FYC_ATTN_CASE(40, 3); FYC_ATTN_CASE(48, 3); }
  else switch (p.d) { FYC_ATTN_CASE(8, 2); FYC_ATTN_CASE(16, 2); FYC_ATTN_CASE(24, 2); FYC_ATTN_CASE(32, 2); FYC_ATTN_CASE(40, 2); FYC_ATTN_CASE(48, 2); }
  FYC_FAIL(-2, "fyc_attention: head dim %d not built", p.d);
}
// head dims 56..96
template <typename T> int run_medium(const AttnP& p, int qt, hipStream_t st) {
  if (qt == 4 && p.d <= 80) switch (p.d) { FYC_ATTN_CASE(56, 4); FYC_ATTN_CASE(64, 4); FYC_ATTN_CASE(72, 4); FYC_ATTN_CASE(80, 4); }
  else if (qt == 3 && p.d <= 80) switch (p.d) { FYC_ATTN_CASE(56, 3); FYC_ATTN_CASE(64, 3); FYC_ATTN_CASE(72, 3); FYC_ATTN_CASE(80, 3); }
  else switch (p.d) { FYC_ATTN_CASE(56, 2); FYC_ATTN_CASE(64, 2); FYC_ATTN_CASE(72, 2); FYC_ATTN_CASE(80, 2); FYC_ATTN_CASE(88, 2); FYC_ATTN_CASE(96, 2); }
  FYC_FAIL(-2, "fyc_attention: head dim %d not built", p.d);
}
// head dims 104..160
template <typename T> int run_large(const AttnP& p, hipStream_t st) {
  switch (p.d) { FYC_ATTN_CASE(104, 2); FYC_ATTN_CASE(112, 2); FYC_ATTN_CASE(120, 2); FYC_ATTN_CASE(128, 2); FYC_ATTN_CASE(136, 2); FYC_ATTN_CASE(144, 2); FYC_ATTN_CASE(152, 2); FYC_ATTN_CASE(160, 2); }
  FYC_FAIL(-2, "fyc_attention: head dim %d not built", p.d);
}
}  // namespace fyca
