// fused attention instantiations, f16_t elements, the "medium" head-dim group (attention_groups.h)
#include "attention_groups.h"
namespace fyca {
template int run_medium<f16_t>(const AttnP&, int, hipStream_t);
}  // namespace fyca
