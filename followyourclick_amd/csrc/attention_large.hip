// fused attention instantiations, bf16_t elements, the "large" head-dim group (attention_groups.h)
#include "attention_groups.h"
namespace fyca {
template int run_large<bf16_t>(const AttnP&, hipStream_t);
}  // namespace fyca
