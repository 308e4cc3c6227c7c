// fused attention instantiations, bf16_t elements, the "small" head-dim group (attention_groups.h)
#include "attention_groups.h"
namespace fyca {
template int run_small<bf16_t>(const AttnP&, int, hipStream_t);
}  // namespace fyca
