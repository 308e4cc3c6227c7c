// Host entry of the fused attention: argument validation and dispatch (kernel template: attention_kernel.h;
// instantiations: attention_small.hip, attention_medium.hip, attention_large.hip).
#include "attention_kernel.h"

extern "C" int fyc_attention(const fyc_attn_args* a, void* stream) {
  FYC_REQUIRE(a && a->q && a->k && a->vt && a->o, "fyc_attention: null pointer");
  FYC_REQUIRE(g_fyc_zero_page != nullptr, "fyc_attention: fyc_init() not called");
  FYC_REQUIRE(a->dtype == FYC_BF16 || a->dtype == FYC_F16, "fyc_attention: bf16 / f16 only (f32 parity mode uses materialised attention)");
  FYC_REQUIRE(a->batch > 0 && a->heads > 0 && a->n_q > 0 && a->n_k > 0, "fyc_attention: bad sizes");
  FYC_REQUIRE(a->d % 8 == 0 && a->d >= 8 && a->d <= 160, "fyc_attention: head dim %d (multiple of 8, <= 160)", a->d);
  FYC_REQUIRE(a->ldvt % 8 == 0 && a->ldvt >= a->n_k, "fyc_attention: ldvt=%d must be a multiple of 8 and >= n_k", a->ldvt);
  FYC_REQUIRE(a->ldo % 4 == 0, "fyc_attention: ldo must be a multiple of 4");
  FYC_REQUIRE(a->kv_batch_div >= 1, "fyc_attention: kv_batch_div");
  fyca::AttnP p;
  p.q = (const bf16_t*)a->q; p.k = (const bf16_t*)a->k; p.vt = (const bf16_t*)a->vt; p.o = (bf16_t*)a->o;
  p.batch = a->batch; p.heads = a->heads; p.n_q = a->n_q; p.n_k = a->n_k; p.d = a->d; p.ldo = a->ldo; p.ldvt = a->ldvt;
  p.kv_batch_div = a->kv_batch_div; p.o_accumulate = a->o_accumulate;
  p.sl2e = a->scale * 1.44269504088896340736f; p.o_scale = a->o_scale;
  p.nqb = 0; p.zero = (const char*)g_fyc_zero_page;
  hipStream_t st = (hipStream_t)stream;
  // Queries per wave = 16 * QT.  More queries amortise the K / V^T fragment reads over more MFMAs but cost registers: at d = 40
  // QT = 4 needs 170 VGPRs (2 waves / SIMD), QT = 3 needs 140 (3 waves / SIMD).  Large problems take QT = 3 (measured,
  // profiles/r02_attention_variants.txt); small ones QT = 2 so that the grid still fills the chip.  tuning key 3 forces QT.
  const long long wg3 = (long long)a->batch * a->heads * ((a->n_q + 191) / 192);
  int qt = (a->n_q >= 1024 && wg3 >= 512 && a->d <= 48) ? 3 : 2;      // larger head dims: 3 tiles no longer fit 2 waves / SIMD
  if (g_fyc_tuning[3] >= 2 && g_fyc_tuning[3] <= 4 && (a->d <= 80 || g_fyc_tuning[3] == 2)) qt = g_fyc_tuning[3];
  if (a->dtype == FYC_F16) {
    if (a->d <= 48) return fyca::run_small<f16_t>(p, qt, st);
    if (a->d <= 96) return fyca::run_medium<f16_t>(p, qt, st);
    return fyca::run_large<f16_t>(p, st);
  }
  if (a->d <= 48) return fyca::run_small<bf16_t>(p, qt, st);
  if (a->d <= 96) return fyca::run_medium<bf16_t>(p, qt, st);
  return fyca::run_large<bf16_t>(p, st);
}
