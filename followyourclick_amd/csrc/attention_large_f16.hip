// fused attention instantiations, f16_t elements, the "large" head-dim group (attention_groups.h)
#include "attention_groups.h"
namespace fyca {
template int run_large<f16_t>(const AttnP&, hipStream_t);
}  // namespace fyca
