// Flash-style attention for gfx950: bf16 MFMA QK^T / PV with wave-level online softmax.
//
// Layouts (written by the GEMM "HEADS" epilogue): q,k [B*H][N][d] ; vt [B*H][d][ldvt] (V transposed,
// keys contiguous) ; o token-major [B*N][ldo] (channel = h*d + i) for the to_out GEMM.
//
// Work split: workgroup = 4 wave64 = 64*QT queries of one (batch, head); each wave owns QT tiles of
// 16 queries for the whole key loop.  K and V^T tiles of 64 keys are staged global->LDS by DMA
// (global_load_lds, 16 B/lane, double buffered) and shared by the 4 waves.
//
// MFMA orientation (v_mfma_f32_16x16x32_bf16, D[row][col]: col = lane&15, row = 4*(lane>>4)+reg):
//   S^T = K Q^T  : A = K rows (keys), B = Q rows (queries)  -> a lane holds, for ONE query (col),
//                  scores of 4 keys per tile; two tiles with interleaved key rows give it 8
//                  consecutive keys 8g..8g+7 of a 32-key block.
//   O^T = V^T P^T: B = P^T is exactly those 8 scores (exponentiated, bf16) - no cross-lane traffic,
//                  A = V^T rows (dv) x 8 consecutive keys = one 16-B LDS read.
// The softmax state (running max / sum) is per query = per lane column, so rescaling O^T is a
// lane-local multiply.  Row max needs two xor-shuffles (across the four lane quads) per 32 keys.
#include "fyc_common.h"

namespace {

struct AttnP {
  const bf16_t* q; const bf16_t* k; const bf16_t* vt; bf16_t* o;
  int batch, heads, n_q, n_k, d, ldo, ldvt, kv_batch_div, o_accumulate;
  float sl2e, o_scale;   // scale * log2(e)
  int nqb;               // query blocks per (b,h)
  const char* zero;
};

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int DP, int DVT, int QT>
__global__ void __launch_bounds__(256) fyc_attn_kernel(const AttnP p) {
  constexpr int KS = DP / 32;            // MFMA k-steps over the (padded) head dim
  constexpr int DC = DP / 8;             // 16-B chunks per K row
  constexpr bool KXOR = (DC == 8);       // 128-B rows: XOR swizzle; otherwise one pad chunk per row
  constexpr int PC = KXOR ? 8 : DC + 1;  // K row pitch in chunks
  constexpr int KB = 64;                 // keys per LDS tile
  constexpr int K_IT = (KB * PC + 255) / 256, K_BYTES = K_IT * 256 * 16;
  constexpr int V_ROWS = DVT * 16, V_IT = (V_ROWS * 8 + 255) / 256, V_BYTES = V_IT * 256 * 16;
  constexpr int STAGE = K_BYTES + V_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r16 = lane & 15;

  // block -> (bh, query block); keep all query blocks of one (b,h) on one XCD (block b runs on XCD b%8)
  const int BH = p.batch * p.heads;
  int bh, qb;
  if ((BH & 7) == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    bh = (j / p.nqb) * 8 + xcd;
    qb = j % p.nqb;
  } else {
    bh = blockIdx.x / p.nqb;
    qb = blockIdx.x % p.nqb;
  }
  const int b = bh / p.heads, h = bh - b * p.heads;
  const int kvb = (b / p.kv_batch_div) * p.heads + h;
  const bf16_t* Q = p.q + (long long)bh * p.n_q * p.d;
  const bf16_t* K = p.k + (long long)kvb * p.n_k * p.d;
  const bf16_t* VT = p.vt + (long long)kvb * p.d * p.ldvt;
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(p.zero);

  // ---- Q fragments (B operand): lane (query = r16, quad g) holds Q[query][32ks + 8g .. +8]
  const int qbase = qb * (64 * QT) + wave * (16 * QT);
  bf16x8 qf[QT][KS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int query = qbase + qt * 16 + r16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int dd = 32 * ks + 8 * g;
      const bf16_t* src = (query < p.n_q && dd < p.d) ? Q + (long long)query * p.d + dd : zero;
      qf[qt][ks] = *reinterpret_cast<const bf16x8*>(src);
    }
  }

  f32x4 o[QT][DVT];
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m_run[qt] = -INFINITY;
    l_run[qt] = 0.f;
#pragma unroll
    for (int dv = 0; dv < DVT; ++dv) o[qt][dv] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  const int ntiles = (p.n_k + KB - 1) / KB;
  const int dchunks = p.d >> 3;

  auto issue = [&](int tile, int stage) {
    char* sK = smem + stage * STAGE;
    char* sV = sK + K_BYTES;
    const int key0 = tile * KB;
#pragma unroll
    for (int it = 0; it < K_IT; ++it) {
      const int L = it * 256 + tid, row = L / PC, cc = L - row * PC;
      const int c = KXOR ? (cc ^ (row & 7)) : cc;
      const int key = key0 + row;
      const bool ok = row < KB && c < dchunks && key < p.n_k;
      glds16(ok ? K + (long long)key * p.d + c * 8 : zero, sK + (it * 256 + wave * 64) * 16);
    }
#pragma unroll
    for (int it = 0; it < V_IT; ++it) {
      const int L = it * 256 + tid, row = L >> 3, cc = L & 7;
      const int c = cc ^ (row & 7);
      const int key = key0 + c * 8;
      const bool ok = row < p.d && key < p.ldvt;
      glds16(ok ? VT + (long long)row * p.ldvt + key : zero, sV + (it * 256 + wave * 64) * 16);
    }
  };

  issue(0, 0);
  for (int tile = 0; tile < ntiles; ++tile) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tile + 1 < ntiles) issue(tile + 1, (tile + 1) & 1);
    const char* sK = smem + (tile & 1) * STAGE;
    const char* sV = sK + K_BYTES;
    const bool tail = (tile * KB + KB > p.n_k);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (tile * KB + kb * 32 >= p.n_k) break;  // whole 32-key block out of range (uniform)
      // ---- S^T tiles: tile t row i <-> key kb*32 + 8*(i>>2) + 4t + (i&3)
      f32x4 s[2][QT];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int krow = kb * 32 + 8 * (r16 >> 2) + 4 * t + (r16 & 3);
        bf16x8 kf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int chunk = 4 * ks + g;
          const int off = krow * (PC * 16) + (KXOR ? (chunk ^ (krow & 7)) : chunk) * 16;
          kf[ks] = *reinterpret_cast<const bf16x8*>(sK + off);
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks], qf[qt][ks], a, 0, 0, 0);
          s[t][qt] = a;
        }
      }
      // ---- online softmax; lane holds keys kb*32 + 8g + 4t + r of query r16
      bf16x8 pf[QT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        if (tail) {
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (tile * KB + kb * 32 + 8 * g + 4 * t + r >= p.n_k) s[t][qt][r] = -INFINITY;
        }
        float mx = fmaxf(fmaxf(fmaxf(s[0][qt][0], s[0][qt][1]), fmaxf(s[0][qt][2], s[0][qt][3])),
                         fmaxf(fmaxf(s[1][qt][0], s[1][qt][1]), fmaxf(s[1][qt][2], s[1][qt][3])));
        // max over the 4 lane quads that share this query: xor 16 inside each 32-lane half (ds_swizzle
        // bit-mode, no LDS traffic), then across the halves (v_permlane32_swap)
        mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, mx), 0x401F)));
        {
          const unsigned u = __builtin_bit_cast(unsigned, mx);
          auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
          mx = fmaxf(__builtin_bit_cast(float, sw[0]), __builtin_bit_cast(float, sw[1]));
        }
        // lazy rescale: the running max only grows in the first few key tiles; skip the O^T rescale
        // (12-40 multiplies + one exp per query tile) whenever no lane of the wave saw a new maximum
        if (__builtin_amdgcn_ballot_w64(mx > m_run[qt]) != 0) {
          const float m_new = fmaxf(m_run[qt], mx);
          const float alpha = __builtin_amdgcn_exp2f((m_run[qt] - m_new) * p.sl2e);
          m_run[qt] = m_new;
          l_run[qt] *= alpha;
#pragma unroll
          for (int dv = 0; dv < DVT; ++dv) {
            o[qt][dv][0] *= alpha; o[qt][dv][1] *= alpha; o[qt][dv][2] *= alpha; o[qt][dv][3] *= alpha;
          }
        }
        const float msl = m_run[qt] * p.sl2e;
        float pv[8], sum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][qt][r], p.sl2e, -msl));
            pv[4 * t + r] = e;
            sum += e;
          }
        l_run[qt] += sum;
        u32x4 pk;
#pragma unroll
        for (int i = 0; i < 4; ++i) pk[i] = pack_bf16x2(pv[2 * i], pv[2 * i + 1]);
        pf[qt] = __builtin_bit_cast(bf16x8, pk);
      }
      // ---- O^T += V^T P^T
#pragma unroll
      for (int dv = 0; dv < DVT; ++dv) {
        const int vrow = dv * 16 + r16;
        const int chunk = kb * 4 + g;
        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sV + vrow * 128 + ((chunk ^ (vrow & 7)) * 16));
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) o[qt][dv] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qt], o[qt][dv], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: lane holds O[query r16][dv*16 + 4g + r]
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float l = l_run[qt];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    const int query = qbase + qt * 16 + r16;
    if (query >= p.n_q) continue;
    bf16_t* orow = p.o + ((long long)b * p.n_q + query) * p.ldo + h * p.d;
#pragma unroll
    for (int dv = 0; dv < DVT; ++dv) {
      const int dd = dv * 16 + 4 * g;
      if (dd >= p.d) continue;
      float v[4] = {o[qt][dv][0] * inv, o[qt][dv][1] * inv, o[qt][dv][2] * inv, o[qt][dv][3] * inv};
      if (p.o_accumulate) {
        float prev[4];
        ElemIO<bf16_t>::ld4(orow + dd, prev);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = prev[r] + p.o_scale * v[r];
      }
      ElemIO<bf16_t>::st4(orow + dd, v);
    }
  }
}

template <int DP, int DVT, int QT>
int launch_attn(const AttnP& p0, hipStream_t st) {
  constexpr int DC = DP / 8;
  constexpr int PC = (DC == 8) ? 8 : DC + 1;
  constexpr int K_BYTES = ((64 * PC + 255) / 256) * 256 * 16;
  constexpr int V_BYTES = ((DVT * 16 * 8 + 255) / 256) * 256 * 16;
  constexpr int smem = 2 * (K_BYTES + V_BYTES);
  auto kern = fyc_attn_kernel<DP, DVT, QT>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_done = true;
  }
  AttnP p = p0;
  p.nqb = (p.n_q + 64 * QT - 1) / (64 * QT);
  dim3 grid(p.batch * p.heads * p.nqb);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, p);
  FYC_CHECK_LAUNCH("fyc_attention");
  return 0;
}

}  // namespace

extern "C" int fyc_attention(const fyc_attn_args* a, void* stream) {
  FYC_REQUIRE(a && a->q && a->k && a->vt && a->o, "fyc_attention: null pointer");
  FYC_REQUIRE(g_fyc_zero_page != nullptr, "fyc_attention: fyc_init() not called");
  FYC_REQUIRE(a->dtype == FYC_BF16, "fyc_attention: bf16 only (f32 parity mode uses materialised attention)");
  FYC_REQUIRE(a->batch > 0 && a->heads > 0 && a->n_q > 0 && a->n_k > 0, "fyc_attention: bad sizes");
  FYC_REQUIRE(a->d % 8 == 0 && a->d >= 8 && a->d <= 160, "fyc_attention: head dim %d (multiple of 8, <= 160)", a->d);
  FYC_REQUIRE(a->ldvt % 8 == 0 && a->ldvt >= a->n_k, "fyc_attention: ldvt=%d must be a multiple of 8 and >= n_k", a->ldvt);
  FYC_REQUIRE(a->ldo % 4 == 0, "fyc_attention: ldo must be a multiple of 4");
  FYC_REQUIRE(a->kv_batch_div >= 1, "fyc_attention: kv_batch_div");
  AttnP p;
  p.q = (const bf16_t*)a->q; p.k = (const bf16_t*)a->k; p.vt = (const bf16_t*)a->vt; p.o = (bf16_t*)a->o;
  p.batch = a->batch; p.heads = a->heads; p.n_q = a->n_q; p.n_k = a->n_k; p.d = a->d; p.ldo = a->ldo; p.ldvt = a->ldvt;
  p.kv_batch_div = a->kv_batch_div; p.o_accumulate = a->o_accumulate;
  p.sl2e = a->scale * 1.44269504088896340736f; p.o_scale = a->o_scale;
  p.nqb = 0; p.zero = (const char*)g_fyc_zero_page;
  hipStream_t st = (hipStream_t)stream;
  // 64 queries per wave (QT=4) amortises the K / V^T fragment reads over twice the MFMAs; it needs n_q large
  // enough to still fill the chip.  tuning key 3 forces QT (A/B measurements).
  const long long wg4 = (long long)a->batch * a->heads * ((a->n_q + 255) / 256);
  bool qt4 = a->n_q >= 1024 && wg4 >= 512 && a->d <= 80;
  if (g_fyc_tuning[3] == 2) qt4 = false;
  if (g_fyc_tuning[3] == 4 && a->d <= 80) qt4 = true;
  if (a->d <= 32) return qt4 ? launch_attn<32, 2, 4>(p, st) : launch_attn<32, 2, 2>(p, st);
  if (a->d <= 48) return qt4 ? launch_attn<64, 3, 4>(p, st) : launch_attn<64, 3, 2>(p, st);
  if (a->d <= 64) return qt4 ? launch_attn<64, 4, 4>(p, st) : launch_attn<64, 4, 2>(p, st);
  if (a->d <= 80) return qt4 ? launch_attn<96, 5, 4>(p, st) : launch_attn<96, 5, 2>(p, st);
  if (a->d <= 96) return launch_attn<96, 6, 2>(p, st);
  if (a->d <= 128) return launch_attn<128, 8, 2>(p, st);
  return launch_attn<160, 10, 2>(p, st);
}
