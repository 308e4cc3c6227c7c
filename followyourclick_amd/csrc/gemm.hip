// Host entry of the GEMM family: argument validation, tile/ring selection, dispatch.
// Kernel template: gemm_kernel.h; instantiations: gemm_bf16_plain.hip, gemm_bf16_conv.hip, gemm_f32.hip.
// The round-4 main-loop experiments that measured slower on every shape (ping-pong wave groups, tile configs 21 / 22 / 23; epilogue
// under the next tile's K loop, config 31) live in tools/exp/gemm_variants/ and are only part of a library built with
// FYC_GEMM_VARIANTS=1 (python -m followyourclick_amd._build; tests: tools/exp/gemm_variants/test_gemm_variants_gpu.py).
#ifdef FYC_GEMM_VARIANTS
#include "../../tools/exp/gemm_variants/gemm_pp_kernel.h"
#include "../../tools/exp/gemm_variants/gemm_ov_kernel.h"
#else
#include "gemm_kernel.h"
namespace fycg {
constexpr bool pp_cfg(int) { return false; }
constexpr bool ov_cfg(int) { return false; }
inline int run_pp_plain(const GemmP&, int, hipStream_t) { return -2; }
inline int run_pp_conv(const GemmP&, int, hipStream_t) { return -2; }
inline int run_ov(const GemmP&, int, hipStream_t) { return -2; }
}  // namespace fycg
#endif

using fycg::GemmP;

namespace {
// ---- split-K: small M with a long K (the 8x8-latent level: M = 2048, K up to 23040) -----------------------------------------
// A 128x64 tile grid (320 tiles) re-fetches 3.4x the operand bytes per FLOP of the 320-wide tiles, and 128x320 tiles alone
// leave 3/4 of the CUs idle (64 tiles).  Each output tile is therefore cut into `splitk` K slices (one work item each, raw f32
// partials to the caller's workspace) and this kernel adds the slices and applies the LINEAR epilogue.
template <typename T>
__global__ void __launch_bounds__(256) splitk_finish_kernel(const float* __restrict__ ws, int S, const float* __restrict__ bias,
                                                            const float* __restrict__ rowbias, int rows_per_batch, int ldrb,
                                                            const T* __restrict__ residual, int ldr, T* __restrict__ out, int ldo,
                                                            int M, int N, float out_scale) {
  const int n8 = N >> 3;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < (long long)M * n8; idx += gridDim.x * 256ll) {
    const int m = (int)(idx / n8), n = (int)(idx - (long long)m * n8) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    for (int s = 0; s < S; ++s) {
      const float* src = ws + ((long long)s * M + m) * N + n;
      const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] += a[e]; v[4 + e] += b[e]; }
    }
    if (bias) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bias[n + e];
    }
    if (rowbias) {
      const float* rb = rowbias + (long long)(m / rows_per_batch) * ldrb + n;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += rb[e];
    }
    if (residual) {
      float r[8];
      load8<T>(residual + (long long)m * ldr + n, r);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += r[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= out_scale;
    store8<T>(out + (long long)m * ldo + n, v);
  }
}

// The same finish with the OUTPUT STATISTICS of the LINEAR epilogue (fyc_gemm chan_parts; round 6: the split-K convolutions of the
// 8x8 level used to send their consumers to the separate fyc_gn_stats pass - 40 launches per forward).  One block = 128 rows (the
// row tile both split tile configs use: fyc_gemm_stat_layout) x 64 columns: the values go out as above, their rounded copies are
// parked in LDS and thread (slot, column) adds the rows of its sample slot IN ROW ORDER - no atomics, bitwise repeatable - and writes
// chan_parts[tile][slot][n] = {sum, sum of squares} of the values as stored.
template <typename T>
__global__ void __launch_bounds__(256) splitk_finish_stats_kernel(const float* __restrict__ ws, int S, const float* __restrict__ bias,
                                                                  const float* __restrict__ rowbias, int rows_per_batch, int ldrb,
                                                                  const T* __restrict__ residual, int ldr, T* __restrict__ out, int ldo,
                                                                  int M, int N, float out_scale, float* __restrict__ chan_parts, int cs_rows, int cs_slots) {
  __shared__ float tile[128][64];
  const int tile_m = blockIdx.x, n0 = blockIdx.y * 64;
  const int cch = threadIdx.x & 7, rsub = threadIdx.x >> 3;
  const int n = n0 + cch * 8;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {      // (unrolled: the partial-sum loads of the four passes are independent)
    const int lr = pass * 32 + rsub, m = tile_m * 128 + lr;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (m < M && n < N) {
      for (int s = 0; s < S; ++s) {
        const float* src = ws + ((long long)s * M + m) * N + n;
        const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += a[e]; v[4 + e] += b[e]; }
      }
      if (bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bias[n + e];
      }
      if (rowbias) {
        const float* rb = rowbias + (long long)(m / rows_per_batch) * ldrb + n;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rb[e];
      }
      if (residual) {
        float r[8];
        load8<T>(residual + (long long)m * ldr + n, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += r[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = round_through<T>(v[e] * out_scale);     // the value as stored
      store8<T>(out + (long long)m * ldo + n, v);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[lr][cch * 8 + e] = v[e];                     // rows / columns outside the problem: 0
  }
  __syncthreads();
  const int col = threadIdx.x & 63, slot = threadIdx.x >> 6;                       // one wave per sample slot (cs_slots <= 4)
  if (slot < cs_slots && n0 + col < N) {
    const int first = (tile_m * 128) / cs_rows;
    const long long lo = (long long)(first + slot) * cs_rows - (long long)tile_m * 128, hi = lo + cs_rows;
    const int r0 = lo < 0 ? 0 : (int)lo, r1 = hi > 128 ? 128 : (int)hi;
    float sx = 0.f, sq = 0.f;
    for (int r = r0; r < r1; ++r) { const float x = tile[r][col]; sx += x; sq = __builtin_fmaf(x, x, sq); }
    *reinterpret_cast<float2*>(chan_parts + (((long long)tile_m * cs_slots + slot) * N + n0 + col) * 2) = make_float2(sx, sq);
  }
}

// Tile / ring-depth choice.  g_fyc_tuning[1] / [2] force a config / depth (bench sweeps, tests).
void choose(const GemmP& p, int batch, int tile, int& cfg, int& ns) {
  // measured on MI355X (tools/gemm_bench.py, profiles/r01_gemm_tile_sweep.txt): the DMA fill rate of the
  // LDS ring, not the MFMA rate, bounds this kernel, so the widest tile that still fills the chip wins.
  ns = 2;
  if (p.N % 320 == 0) {
    // One or two column tiles per row and a short K loop (N = 320, K <= 320; N = 640, K <= 640): the epilogue is most of the
    // tile, and 128-row tiles give every CU twice as many epilogues to overlap with the next tile's fill.  Measured with the
    // operands coming from HBM and the epilogue features the UNet uses (tools/gemm_probe.py, profiles/r02_gemm_probe_cold_sweep.txt):
    // 95 (config 6) vs 111 (8) / 122 (5) us at M = 131072, N = K = 320 + residual, 67 vs 75 (5) at M = 32768, N = K = 640; the
    // Infinity-Cache-hot sweep had ranked 8 first.  (Shape-only rule: fyc_gemm_stat_layout must predict the tile from M, N, K.)
    // Round 6 (profiles/r06_gemm_tile_sweep.txt, after the packed epilogue of round 4 and this round's epilogue work): with a residual the
    // two tiles are level (84 / 58 us either way), WITHOUT one the 256-row tile is 10-18 % faster (q2 head projection 131072x320x320: 72 vs
    // 88 us, 32768x640x640: 51 vs 61; proj_in 32768x640x640: 44 vs 49) - the short-K rule now only holds for problems with a residual.
    const bool short_k = p.mode == FYC_GEMM_PLAIN && (p.N == 320 || p.N == 640) && p.K <= p.N && p.residual != nullptr;
    if (p.M >= 16384) cfg = short_k ? 6 : 5;
    else if (p.M >= 4096) cfg = (p.N >= 5120) ? 5 : 6;
    // M < 4096 (the 8x8 latent level): the widest tile that still gives the chip ~200+ work items.  Cold-operand probe at
    // M = 2048 (profiles/r04_gemm_small_m_ring_depth.txt): N = 10240 GEGLU 256x320 59 us vs 95 us on 128x64 tiles, N = 3840
    // 128x128 35 vs 47 us, N = 1280 stays on 128x64 (24 vs 27 / 34 us).  Deeper rings on the small tiles measured equal
    // (3-deep) or 1.6x slower (4-deep: one workgroup per CU) - more workgroups in flight, not a deeper ring, hide the latency.
    else cfg = (p.N >= 5120) ? 5 : (p.N >= 2560 ? 1 : 2);
  } else if (p.N % 256 == 0 && p.M >= 16384) {
    cfg = 7;
  } else if (p.N % 128 == 0 || p.N > 512) {
    cfg = (p.M >= 16384) ? 3 : 1;
  } else {
    cfg = 2;
  }
  if (tile > 0) { cfg = tile & 0xff; if (tile >> 8) ns = tile >> 8; }
  if (g_fyc_tuning[1] > 0) cfg = g_fyc_tuning[1];
  if (g_fyc_tuning[2] > 0) ns = g_fyc_tuning[2];
  // GEGLU pairs 16-column value / gate blocks inside a wave: config 6 gives a wave 5 column blocks (128x320 over 2x4 waves) and
  // used to leave the output unwritten (found by tools/gemm_diag.py at M = 4096 / 8192, N = 2560 - shapes the UNet never issued)
  if ((cfg == 6 || cfg == 11) && p.epilogue == FYC_EPI_GEGLU) cfg = 5;
  if (cfg == 22 && p.epilogue == FYC_EPI_GEGLU) cfg = 21;
  // ping-pong main loop (gemm_pp_kernel.h) for the 8-wave tiles: fyc_set_tuning key 9 = 1 keeps the one-phase loop, 2 forces the
  // ping-pong one wherever it is built; fyc_gemm() falls back to the one-phase twin when the problem does not qualify
#ifdef FYC_GEMM_VARIANTS
  if (g_fyc_tuning[9] == 2 && tile <= 0 && g_fyc_tuning[1] <= 0) {
    if (cfg == 5) cfg = 21;
    else if (cfg == 6) cfg = 22;
    else if (cfg == 7) cfg = 23;
  }
  // overlapped-epilogue kernel (gemm_ov_kernel.h, tile config 31 = 128x320): key 9 = 3 takes it for every N = 320 k problem the
  // library would give a 256x320 / 128x320 tile; fyc_gemm() falls back to 6 when the problem does not qualify
  if (g_fyc_tuning[9] == 3 && tile <= 0 && g_fyc_tuning[1] <= 0 && (cfg == 5 || cfg == 6) && p.epilogue == FYC_EPI_LINEAR) cfg = 31;
#endif
  if (!((cfg == 1 && ns == 3) || (cfg == 2 && (ns == 3 || ns == 4) && p.mode == FYC_GEMM_PLAIN))) ns = 2;   // deeper rings: config 1 (3) and, for linears, config 2 (3, 4)
}
// the one-phase twin of a ping-pong tile config (same tile, same wave grid)
int pp_twin(int cfg) { return cfg == 21 ? 5 : cfg == 22 ? 6 : cfg == 23 ? 7 : cfg == 31 ? 6 : cfg; }
// column-tile width / row-tile height of a tile config (gemm_kernel.h::dispatch_cfg)
int tile_bn(int cfg) {
  switch (cfg) { case 2: case 4: return 64; case 5: case 6: case 8: case 12: case 21: case 22: case 31: return 320; case 7: case 13: case 23: return 256; case 11: return 160; default: return 128; }
}
int tile_bm(int cfg) {
  switch (cfg) { case 3: case 4: case 5: case 7: case 12: case 13: case 14: case 21: case 23: return 256; default: return 128; }
}
// the 32x32x16-instruction twin of a tile config (same tile, same wave grid, same epilogues): 0 = none built
#ifdef FYC_GEMM_MI32
int mi32_twin(int cfg) { return cfg == 5 ? 12 : cfg == 7 ? 13 : cfg == 3 ? 14 : 0; }
#else
int mi32_twin(int) { return 0; }
#endif
// sample slots a row tile of bm rows can touch when a sample has cs_rows rows
int stat_slots(int bm, int cs_rows) {
  if (cs_rows % bm == 0) return 1;
  if (bm % cs_rows == 0) return bm / cs_rows;
  return (bm - 1) / cs_rows + 2;
}
// `stats`: the epilogue also writes output statistics - not built for the 64-byte K-tile configs (their ring stage is too small
// for the accumulators), which fall back to the 128-byte ones
inline bool is16(int dtype) { return dtype == FYC_BF16 || dtype == FYC_F16; }   // the 16-bit storage formats share every tile / epilogue decision
void pick(const fyc_gemm_args* a, int& cfg, int& ns, bool stats) {
  if (a->dtype == FYC_F32) { cfg = (a->N % 128 == 0) ? 1 : 2; ns = 2; return; }
  GemmP q;
  memset(&q, 0, sizeof(q));
  q.M = a->M; q.N = a->N; q.K = a->K; q.mode = a->mode; q.epilogue = a->epilogue;
  // (the tile of a problem with output statistics must follow from its shape alone - fyc_gemm_stat_layout has no pointers: such problems keep
  // the residual-independent short-K rule)
  q.residual = stats ? (const char*)a : (const char*)a->residual;
  choose(q, a->batch > 0 ? a->batch : 1, a->tile, cfg, ns);
  if (stats && cfg == 8) cfg = 6;
  if (stats && cfg == 10) cfg = 1;
}
// K slices per output tile (1 = no split) and the tile config a split problem uses
int split_of(const fyc_gemm_args* a, int& cfg) {
  if (!is16(a->dtype) || a->epilogue != FYC_EPI_LINEAR || a->act != FYC_ACT_NONE || a->batch > 1 || a->tile != 0 || g_fyc_tuning[1] > 0 || g_fyc_tuning[0] == 1) return 1;
  if (a->ln_stats != nullptr || a->row_parts != nullptr) return 1;      // (chan_parts: the finish kernel writes them, round 6)
  // fyc_set_tuning key 10 = v > 0 (A/B): at least v K tiles per slice instead of 16, and K >= 128 v instead of 2048 - the K = 1280 linears
  // of the 8x8 latent level (profiles/r06_gemm_small_m_split_k.txt)
  const int min_kt = g_fyc_tuning[10] > 0 ? g_fyc_tuning[10] : 10;     // (16 until round 6: the K = 2560 shortcut of the 8x8 level now splits 4 ways, 39 -> 33 us)
  if (a->M > 4096 || a->K < (g_fyc_tuning[10] > 0 ? 128 * min_kt : 2048) || a->N % 8 != 0 || a->N < 256) return 1;
  const int c = (a->N % 320 == 0) ? 6 : 1;                     // 128x320 or 128x128 tiles
  const int bn = (c == 6) ? 320 : 128;
  const long long tiles = (long long)((a->M + 127) / 128) * ((a->N + bn - 1) / bn);
  const int kt = (a->K + 63) / 64;
  int s = (int)(256 / tiles);
  if (s > 8) s = 8;
  while (s > 1 && kt / s < min_kt) --s;                        // keep >= min_kt K tiles per slice
  if (s < 2) return 1;
  cfg = c;
  return s;
}
}  // namespace

extern "C" int64_t fyc_gemm_workspace_bytes(const fyc_gemm_args* a) {
  if (a == nullptr) return 0;
  int cfg = 0;
  const int s = split_of(a, cfg);
  return s > 1 ? (int64_t)s * a->M * a->N * 4 : 0;
}

extern "C" int fyc_gemm_row_parts(const fyc_gemm_args* a) {
  if (a == nullptr || a->N <= 0) return 0;
  int cfg = 1, ns = 2;
  pick(a, cfg, ns, true);
  const int bn = tile_bn(cfg);
  return (a->N + bn - 1) / bn;
}

extern "C" int fyc_gemm_stat_layout(const fyc_gemm_args* a, int32_t* tile_rows, int32_t* slots) {
  if (a == nullptr || a->M <= 0 || a->cs_rows <= 0) return 0;
  int cfg = 1, ns = 2;
  pick(a, cfg, ns, true);
  int scfg = 0;
  fyc_gemm_args q = *a;                     // (the query carries no pointers: split_of() must see the call as the engine will make it)
  q.chan_parts = nullptr; q.row_parts = nullptr; q.ln_stats = nullptr; q.act = FYC_ACT_NONE; q.epilogue = FYC_EPI_LINEAR;
  const int bm = split_of(&q, scfg) > 1 ? 128 : tile_bm(cfg);      // split-K problems: the finish kernel's 128-row blocks
  if (tile_rows) *tile_rows = bm;
  if (slots) *slots = stat_slots(bm, a->cs_rows);
  return (a->M + bm - 1) / bm;
}

extern "C" int fyc_gemm(const fyc_gemm_args* a, void* stream) {
  FYC_REQUIRE(a != nullptr, "fyc_gemm: null args");
  FYC_REQUIRE(g_fyc_zero_page != nullptr, "fyc_gemm: fyc_init() not called");
  FYC_REQUIRE(a->dtype == FYC_F32 || is16(a->dtype), "fyc_gemm: bad dtype %d", a->dtype);
  const int es = is16(a->dtype) ? 2 : 4, ch = 16 / es;
  FYC_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "fyc_gemm: empty problem M=%d N=%d K=%d", a->M, a->N, a->K);
  FYC_REQUIRE((a->ln_stats == nullptr) == (a->ln_colsum == nullptr) && (a->ln_stats == nullptr || (a->mode == FYC_GEMM_PLAIN && a->batch <= 1 && ((uintptr_t)a->ln_stats % 8) == 0 && ((uintptr_t)a->ln_colsum % 16) == 0)),
              "fyc_gemm: ln_stats / ln_colsum must come together (PLAIN mode, no batch, 8-/16-byte aligned)");
  FYC_REQUIRE(a->act >= FYC_ACT_NONE && a->act <= FYC_ACT_QUICK_GELU && (a->act == FYC_ACT_NONE || a->epilogue == FYC_EPI_LINEAR),
              "fyc_gemm: act=%d needs the LINEAR epilogue", a->act);
  FYC_REQUIRE(a->K % ch == 0, "fyc_gemm: K=%d must be a multiple of %d", a->K, ch);
  FYC_REQUIRE(a->ldw % ch == 0 && a->stride_w % ch == 0, "fyc_gemm: ldw/stride_w must keep 16-B alignment");
  FYC_REQUIRE(((uintptr_t)a->a % 16) == 0 && ((uintptr_t)a->w % 16) == 0, "fyc_gemm: operands must be 16-B aligned");
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.a = (const char*)a->a; p.a2 = (const char*)a->a2; p.w = (const char*)a->w;
  p.k_split = a->k_split; p.lda2 = a->lda2; p.bias = a->bias; p.rowbias = a->rowbias;
  p.residual = (const char*)a->residual; p.out = (char*)a->out;
  p.M = a->M; p.N = a->N; p.K = a->K; p.lda = a->lda; p.ldw = a->ldw; p.ldo = a->ldo; p.ldr = a->ldr;
  p.stride_a = a->stride_a; p.stride_w = a->stride_w; p.stride_o = a->stride_o;
  p.mode = a->mode; p.epilogue = a->epilogue;
  p.Hout = a->Hout; p.Wout = a->Wout; p.Hin = a->Hin; p.Win = a->Win; p.Cin = a->Cin; p.conv_stride = a->conv_stride; p.conv_pad = a->conv_pad;
  p.rows_per_batch = a->rows_per_batch > 0 ? a->rows_per_batch : 1;
  p.ldrb = a->ldrb > 0 ? a->ldrb : a->N;
  p.out_scale = a->out_scale;
  p.act = a->act;
  p.ln_stats = a->ln_stats; p.ln_colsum = a->ln_colsum;
  p.ln_nparts = a->ln_nparts; p.ln_eps = a->ln_eps;
  FYC_REQUIRE(a->ln_nparts >= 0 && (a->ln_nparts == 0 || (a->ln_stats != nullptr && a->a2 == nullptr)), "fyc_gemm: ln_nparts=%d needs ln_stats (and no a2)", a->ln_nparts);
  p.chan_parts = a->chan_parts; p.cs_rows = a->cs_rows; p.row_parts = a->row_parts; p.row_nparts = a->row_nparts;
  if (a->chan_parts != nullptr || a->row_parts != nullptr) {
    FYC_REQUIRE(a->epilogue == FYC_EPI_LINEAR && (a->batch <= 1) && (a->dtype == FYC_F32 || a->act == FYC_ACT_NONE), "fyc_gemm: output statistics need the LINEAR epilogue without batch (bf16: without activation)");
    int32_t bm = 0, slots = 0;
    if (a->chan_parts != nullptr) (void)fyc_gemm_stat_layout(a, &bm, &slots);
    p.cs_slots = slots;
    FYC_REQUIRE(a->chan_parts == nullptr || (a->cs_rows > 0 && a->cs_rows % 16 == 0 && slots >= 1 && slots <= 4 && a->M % a->cs_rows == 0 && ((uintptr_t)a->chan_parts % 8) == 0),
                "fyc_gemm: chan_parts needs cs_rows (=%d) a multiple of 16 dividing M=%d, and at most 4 samples per row tile (%d)", a->cs_rows, a->M, slots);
    FYC_REQUIRE(a->row_parts == nullptr || (((uintptr_t)a->row_parts % 8) == 0 && a->row_nparts == fyc_gemm_row_parts(a)),
                "fyc_gemm: row_parts needs row_nparts == fyc_gemm_row_parts() = %d (got %d)", fyc_gemm_row_parts(a), a->row_nparts);
  }
  p.zero = (const char*)g_fyc_zero_page;
  const int batch = a->batch > 0 ? a->batch : 1;
  {   // the loaders keep 32-bit element offsets inside one batch element
    const long long lim = 0xffffffffll;
    const bool small = (long long)a->N * a->ldw < lim &&
                       (a->mode == FYC_GEMM_PLAIN ? ((long long)a->M * a->lda < lim && (a->a2 == nullptr || (long long)a->M * a->lda2 < lim))
                                                  : (a->Hout > 0 && a->Wout > 0 && (long long)a->M / ((long long)a->Hout * a->Wout) * a->Hin * a->Win * a->Cin < lim));
    FYC_REQUIRE(small, "fyc_gemm: an operand has 2^32 elements or more (M=%d lda=%d N=%d ldw=%d): split the call", a->M, a->lda, a->N, a->ldw);
  }
  if (a->mode == FYC_GEMM_PLAIN) {
    FYC_REQUIRE(a->lda % ch == 0 && a->stride_a % ch == 0, "fyc_gemm: lda/stride_a must keep 16-B alignment");
    if (a->a2 != nullptr) {
      FYC_REQUIRE(a->k_split > 0 && a->k_split < a->K && a->k_split % (8 * ch) == 0, "fyc_gemm: k_split=%d must be a multiple of %d inside (0, K)", a->k_split, 8 * ch);
      FYC_REQUIRE(a->lda2 % ch == 0 && ((uintptr_t)a->a2 % 16) == 0 && batch == 1, "fyc_gemm: a2 alignment / batch");
    }
  } else {
    const int bk = 8 * ch;
    FYC_REQUIRE(a->mode == FYC_GEMM_CONV3X3 || a->mode == FYC_GEMM_CONV3X3_UP2, "fyc_gemm: bad mode %d", a->mode);
    FYC_REQUIRE(a->Cin > 0 && a->Cin % bk == 0, "fyc_gemm conv: Cin=%d must be a multiple of %d", a->Cin, bk);
    FYC_REQUIRE(a->K == 9 * a->Cin, "fyc_gemm conv: K=%d != 9*Cin", a->K);
    FYC_REQUIRE(a->Hout > 0 && a->Wout > 0 && a->Hin > 0 && a->Win > 0, "fyc_gemm conv: bad spatial dims");
    FYC_REQUIRE(a->M % (a->Hout * a->Wout) == 0, "fyc_gemm conv: M=%d not a multiple of Hout*Wout", a->M);
    FYC_REQUIRE(batch == 1, "fyc_gemm conv: batch must be 1");
    FYC_REQUIRE(a->a2 == nullptr, "fyc_gemm conv: a2 is a PLAIN-mode feature");
    if (a->mode == FYC_GEMM_CONV3X3) {
      FYC_REQUIRE(a->conv_stride == 1 || a->conv_stride == 2, "fyc_gemm conv: stride %d", a->conv_stride);
      FYC_REQUIRE(a->conv_pad == 0 || a->conv_pad == 1, "fyc_gemm conv: conv_pad must be 0 or 1");
      FYC_REQUIRE(a->Hout == (a->Hin + a->conv_pad + 1 - 3) / a->conv_stride + 1 && a->Wout == (a->Win + a->conv_pad + 1 - 3) / a->conv_stride + 1,
                  "fyc_gemm conv: output size mismatch");
    } else {
      p.conv_stride = 1;
      p.conv_pad = 1;
      FYC_REQUIRE(a->Hout >= a->Hin && a->Wout >= a->Win, "fyc_gemm upconv: output must not be smaller than the input");
      p.up_exact2 = (a->Hout == 2 * a->Hin && a->Wout == 2 * a->Win) ? 1 : 0;
      p.up_sh = (float)a->Hin / (float)a->Hout;
      p.up_sw = (float)a->Win / (float)a->Wout;
    }
  }
  if (a->epilogue == FYC_EPI_GEGLU) {
    FYC_REQUIRE(a->N % 32 == 0, "fyc_gemm GEGLU: N=%d must be a multiple of 32", a->N);
    FYC_REQUIRE(a->residual == nullptr && a->rowbias == nullptr, "fyc_gemm GEGLU: residual/rowbias unsupported");
  } else if (a->epilogue == FYC_EPI_HEADS) {
    FYC_REQUIRE(a->seg_cols > 0 && a->heads > 0 && a->tokens > 0, "fyc_gemm HEADS: bad seg_cols/heads/tokens");
    FYC_REQUIRE(a->N % a->seg_cols == 0 && a->N / a->seg_cols <= 3, "fyc_gemm HEADS: N=%d vs seg_cols=%d", a->N, a->seg_cols);
    FYC_REQUIRE(a->seg_cols % a->heads == 0 && (a->seg_cols / a->heads) % 4 == 0, "fyc_gemm HEADS: head_dim must be a multiple of 4");
    FYC_REQUIRE(a->M % a->tokens == 0, "fyc_gemm HEADS: M not a multiple of tokens");
    FYC_REQUIRE(a->residual == nullptr && batch == 1, "fyc_gemm HEADS: residual/batch unsupported");
    p.seg_cols = a->seg_cols; p.heads = a->heads; p.tokens = a->tokens; p.head_dim = a->seg_cols / a->heads;
    p.inv_seg_cols = 1.0f / (float)p.seg_cols; p.inv_head_dim = 1.0f / (float)p.head_dim;      // (fdiv_small: N < 2^16, divisors <= 2^12)
    FYC_REQUIRE(a->N < 65536 && p.seg_cols <= 4096, "fyc_gemm HEADS: N=%d / seg_cols=%d beyond the epilogue's index arithmetic", a->N, p.seg_cols);
    for (int s = 0; s < a->N / a->seg_cols; ++s) {
      FYC_REQUIRE(a->seg_out[s] != nullptr, "fyc_gemm HEADS: seg_out[%d] null", s);
      p.seg_out[s] = (char*)a->seg_out[s];
      p.seg_transposed[s] = a->seg_transposed[s];
      p.seg_ld[s] = a->seg_ld[s] > 0 ? a->seg_ld[s] : a->tokens;
      FYC_REQUIRE(p.seg_ld[s] >= a->tokens, "fyc_gemm HEADS: seg_ld[%d] < tokens", s);
    }
  } else {
    FYC_REQUIRE(a->epilogue == FYC_EPI_LINEAR, "fyc_gemm: bad epilogue %d", a->epilogue);
    FYC_REQUIRE(a->out != nullptr, "fyc_gemm: out is null");
  }
  p.wide = (is16(a->dtype) && a->epilogue != FYC_EPI_HEADS && a->N % (a->epilogue == FYC_EPI_GEGLU ? 32 : 8) == 0 && a->ldo % 8 == 0 && a->stride_o % 8 == 0 &&
            ((uintptr_t)a->out % 16) == 0 && (a->residual == nullptr || (a->ldr % 8 == 0 && ((uintptr_t)a->residual % 16) == 0)) &&
            (a->bias == nullptr || ((uintptr_t)a->bias % 16) == 0) &&
            (a->rowbias == nullptr || (p.ldrb % 4 == 0 && ((uintptr_t)a->rowbias % 16) == 0)) && g_fyc_tuning[6] == 0) ? 1 : 0;
  // bias / colsum / rowbias rows may be fetched as 16-byte vectors and staged through LDS (always true for the engine's buffers)
  p.colc = (is16(a->dtype) && a->epilogue != FYC_EPI_GEGLU && a->N % 4 == 0 && ((uintptr_t)a->bias % 16) == 0 && ((uintptr_t)a->ln_colsum % 16) == 0 &&
            (a->rowbias == nullptr || (p.ldrb % 4 == 0 && ((uintptr_t)a->rowbias % 16) == 0))) ? 1 : 0;
  if (p.wide) p.colc = 1;
  if (is16(a->dtype) && a->epilogue == FYC_EPI_HEADS && p.colc && g_fyc_tuning[6] == 0 && g_fyc_tuning[7] == 0 && p.head_dim % 8 == 0 && a->tokens % 16 == 0 && a->N % 8 == 0) {
    bool ok = true;                                   // wide head-split epilogue: 16-byte runs into every segment
    for (int s = 0; s < a->N / a->seg_cols; ++s)
      ok = ok && ((uintptr_t)a->seg_out[s] % 16) == 0 && (!a->seg_transposed[s] || p.seg_ld[s] % 8 == 0);
    p.wide = ok ? 1 : 0;
  }
  p.rb_tile = 0;
  hipStream_t st = (hipStream_t)stream;
  int cfg = 1, ns = 2;
  pick(a, cfg, ns, a->chan_parts != nullptr || a->row_parts != nullptr);
  // the packed bf16 LINEAR epilogue stages every per-column input through LDS; two combinations stay with the narrow per-lane epilogue
  // (tile configs 1 / 2): a residual next to a LayerNorm fold, and row-bias groups it cannot stage (not multiples of 16 rows, more than 4
  // per row tile, or the 64-byte-K tiles whose ring stage is too small)
  if (p.wide && a->epilogue == FYC_EPI_LINEAR && a->act == FYC_ACT_NONE) {
    const int bm = tile_bm(cfg);
    const bool rb_multi = a->rowbias != nullptr && p.rows_per_batch % bm != 0;
    if (rb_multi && (cfg == 8 || cfg == 10)) cfg = (cfg == 8) ? 6 : 1;
    if ((a->residual != nullptr && a->ln_stats != nullptr) || (rb_multi && fycg::rowbias_slots(tile_bm(cfg), p.rows_per_batch) == 0)) {
      FYC_REQUIRE(a->chan_parts == nullptr && a->row_parts == nullptr, "fyc_gemm: output statistics need a row-bias layout / LayerNorm + residual combination the 16-byte epilogue covers");
      p.wide = 0;
    }
  }
  const bool f16 = a->dtype == FYC_F16;
  // the ping-pong main loop is built for bf16 problems with the 16-byte epilogues, whole 64-element K tiles (at least two) and no batch
  const bool pp_ok = a->dtype == FYC_BF16 && p.wide && batch == 1 && a->act == FYC_ACT_NONE && a->K % 64 == 0 && a->K >= 128;
#ifndef FYC_GEMM_VARIANTS
  cfg = pp_twin(cfg);                                   // tile configs 21 / 22 / 23 / 31 are not in this build: their one-phase twins run
#endif
  if (fycg::pp_cfg(cfg) && !pp_ok) cfg = pp_twin(cfg);
  if (fycg::ov_cfg(cfg)) {
    const int bm = tile_bm(cfg);
    int scfg0 = 0;
    const bool ov_ok = pp_ok && a->epilogue == FYC_EPI_LINEAR && a->K >= 5 * 64 && a->M % 2 == 0 && a->ln_nparts == 0 && a->row_parts == nullptr &&
                       !(a->ln_stats != nullptr && a->residual != nullptr) && (a->chan_parts == nullptr || a->cs_rows % bm == 0) &&
                       (a->rowbias == nullptr || p.rows_per_batch % bm == 0 || fycg::rowbias_slots(bm, p.rows_per_batch) > 0) && split_of(a, scfg0) <= 1;
    if (!ov_ok) cfg = pp_twin(cfg);
  }
  if ((cfg == 6 || cfg == 11) && a->epilogue == FYC_EPI_GEGLU) cfg = 5;      // (a fallback above may land on the one tile GEGLU is not built for)
  {
    int scfg = 0;
    const int sk = split_of(a, scfg);
    FYC_REQUIRE(!(sk > 1 && a->chan_parts != nullptr) || (p.wide && a->workspace != nullptr && a->workspace_bytes >= (int64_t)sk * a->M * a->N * 4 && ((uintptr_t)a->workspace % 16) == 0),
                "fyc_gemm: chan_parts of a split-K problem (fyc_gemm_workspace_bytes() > 0) are laid out for its finish kernel: pass the workspace (and 16-byte aligned operands)");
    if (sk > 1 && p.wide && a->workspace != nullptr && a->workspace_bytes >= (int64_t)sk * a->M * a->N * 4 && ((uintptr_t)a->workspace % 16) == 0) {
      GemmP q = p;
      q.splitk = sk; q.ws = (float*)a->workspace;
#ifdef FYC_GEMM_VARIANTS
      if (scfg == 6 && g_fyc_tuning[9] == 2 && pp_ok) scfg = 22;
#endif
      const int rc = fycg::pp_cfg(scfg) ? (a->mode == FYC_GEMM_PLAIN ? fycg::run_pp_plain(q, scfg, st) : fycg::run_pp_conv(q, scfg, st))
                     : f16 ? ((a->mode == FYC_GEMM_PLAIN) ? fycg::run_f16_plain(q, batch, scfg, 2, st) : fycg::run_f16_conv(q, batch, scfg, 2, st))
                     : (a->mode == FYC_GEMM_PLAIN) ? fycg::run_bf16_plain(q, batch, scfg, 2, st) : fycg::run_bf16_conv(q, batch, scfg, 2, st);
      if (rc != 0) return rc;
      if (a->chan_parts != nullptr) {
        const dim3 grid((a->M + 127) / 128, (a->N + 63) / 64);
        if (f16)
          hipLaunchKernelGGL(splitk_finish_stats_kernel<f16_t>, grid, dim3(256), 0, st, (const float*)a->workspace, sk, a->bias, a->rowbias, p.rows_per_batch, p.ldrb,
                             (const f16_t*)a->residual, a->ldr, (f16_t*)a->out, a->ldo, a->M, a->N, a->out_scale, a->chan_parts, a->cs_rows, p.cs_slots);
        else
          hipLaunchKernelGGL(splitk_finish_stats_kernel<bf16_t>, grid, dim3(256), 0, st, (const float*)a->workspace, sk, a->bias, a->rowbias, p.rows_per_batch, p.ldrb,
                             (const bf16_t*)a->residual, a->ldr, (bf16_t*)a->out, a->ldo, a->M, a->N, a->out_scale, a->chan_parts, a->cs_rows, p.cs_slots);
        FYC_CHECK_LAUNCH("fyc_gemm split-K finish + statistics");
        return 0;
      }
      const long long items = (long long)a->M * (a->N / 8);
      const int blocks = (int)((items + 255) / 256 < 4096 ? (items + 255) / 256 : 4096);
      if (f16)
        hipLaunchKernelGGL(splitk_finish_kernel<f16_t>, dim3(blocks), dim3(256), 0, st, (const float*)a->workspace, sk, a->bias, a->rowbias, p.rows_per_batch, p.ldrb,
                           (const f16_t*)a->residual, a->ldr, (f16_t*)a->out, a->ldo, a->M, a->N, a->out_scale);
      else
        hipLaunchKernelGGL(splitk_finish_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const float*)a->workspace, sk, a->bias, a->rowbias, p.rows_per_batch, p.ldrb,
                           (const bf16_t*)a->residual, a->ldr, (bf16_t*)a->out, a->ldo, a->M, a->N, a->out_scale);
      FYC_CHECK_LAUNCH("fyc_gemm split-K finish");
      return 0;
    }
  }
  FYC_REQUIRE(a->chan_parts == nullptr || a->dtype == FYC_F32 || p.wide || cfg == 1 || cfg == 2, "fyc_gemm: chan_parts in bf16 needs the 16-byte aligned layout or tile config 1 / 2");
  FYC_REQUIRE(a->row_parts == nullptr || a->dtype == FYC_F32 || p.wide, "fyc_gemm: row_parts in bf16 needs the 16-byte aligned layout (N, ldo, ldr multiples of 8; aligned pointers)");
  if (a->dtype == FYC_F32) return fycg::run_f32(p, batch, cfg, st);
  if (p.act != FYC_ACT_NONE) {
    FYC_REQUIRE(a->mode == FYC_GEMM_PLAIN, "fyc_gemm: act needs the PLAIN mode");
    return f16 ? fycg::run_f16_act(p, batch, cfg, st) : fycg::run_bf16_act(p, batch, cfg, st);
  }
  if (fycg::ov_cfg(cfg)) return fycg::run_ov(p, cfg, st);
  if (fycg::pp_cfg(cfg)) return a->mode == FYC_GEMM_PLAIN ? fycg::run_pp_plain(p, cfg, st) : fycg::run_pp_conv(p, cfg, st);
  // 32x32x16 matrix instruction in the K loop of the tiles that have such a twin (fyc_set_tuning key 14: A/B switch)
  if (p.wide && mi32_twin(cfg) != 0 && g_fyc_tuning[14] == 1) cfg = mi32_twin(cfg);
  if (f16) return a->mode == FYC_GEMM_PLAIN ? fycg::run_f16_plain(p, batch, cfg, ns, st) : fycg::run_f16_conv(p, batch, cfg, ns, st);
  if (a->mode == FYC_GEMM_PLAIN) return fycg::run_bf16_plain(p, batch, cfg, ns, st);
  return fycg::run_bf16_conv(p, batch, cfg, ns, st);
}
