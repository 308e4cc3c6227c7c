// GEMM / implicit-GEMM 3x3 convolution on MFMA for gfx950 (CDNA4): kernel template.
//
//   out[m][n] = (sum_k A[m][k] * W[n][k] + bias[n] + rowbias[m/rpb][n] + residual[m][n]) * out_scale
//
// Data layout: activations channels-last, so a 3x3 convolution is a GEMM whose A rows are
// gathered from shifted pixels: K is ordered (channel slab of 128 B, ky, kx, channel-in-slab), so every
// K-tile lies inside one filter tap - an A-tile row is one contiguous 128-B channel slice of one input
// pixel (or the zero page at the image border / beyond M) - and the 9 taps that re-read the same pixels
// are consecutive K tiles: the ~6 image rows a block touches per slab stay in its XCD's 4 MiB L2
// (tap-major order measured 58 % L2 hits on the 64x64 layers, profiles/r01_gemm_pmc.txt).  Nearest-2x upsampling (Upsample3D) is folded into
// that gather: no 4x tensor is ever written.
//
// Tile: BM x BN x 128 B of K per step, WGM x WGN wave64s, each wave a (BM/WGM) x (BN/WGN) sub-tile
// of 16x16 MFMA blocks.  Operands are staged global -> LDS by direct DMA (global_load_lds, 16 B per
// lane): the LDS image is lane-linear, so the bank-conflict XOR swizzle is applied to the *source*
// chunk each lane fetches and again on the ds_read_b128 address (same involution on both sides).
// NS-deep LDS ring: the DMA of tiles k+1 .. k+NS-1 is in flight while tile k feeds the matrix cores;
// waits are counted (s_waitcnt vmcnt(N), never a drain in steady state) and the barrier is the raw
// s_barrier so that in-flight DMA survives it (cdna_hip_programming.md: "Pipelining across barriers").
//
// MFMA orientation: acc = mfma(Wfrag, Afrag) computes the transposed block D[n][m], so every lane
// ends up with 4 *consecutive output channels* of one row: the epilogue (bias, time-embedding row,
// residual, GEGLU gate, head split) is vectorised over channels and stores 8/16 B per lane.
//
// f32 parity mode uses the same kernel with v_mfma_f32_16x16x4_f32 (exact f32, 1/16 rate).
#pragma once
#include <mutex>
#include <type_traits>

#include "fyc_common.h"

namespace fycg {

constexpr int FYC_MAX_DEVICES = 64;

struct GemmP {
  const char* a; const char* a2; const char* w; const float* bias; const float* rowbias; const char* residual; char* out;
  char* seg_out[3]; int seg_transposed[3]; int seg_ld[3];
  int M, N, K, lda, ldw, ldo, ldr, ldrb, k_split, lda2;
  long long stride_a, stride_w, stride_o;
  int mode, epilogue;
  int Hout, Wout, Hin, Win, Cin, conv_stride, conv_pad;
  int rows_per_batch, seg_cols, heads, tokens, head_dim;
  float out_scale;
  int act;    // FYC_ACT_* (LINEAR epilogue)
  const float* ln_stats; const float* ln_colsum;   // folded LayerNorm (see fyc.h): acc := rstd*(acc - mean*colsum[n])
  int ln_nparts; float ln_eps;                     // ln_nparts > 0: ln_stats holds [M][ln_nparts] partial {sum, sum sq} (row_parts of the producer)
  // statistics of the OUTPUT for the GroupNorm / LayerNorm that consumes it (LINEAR epilogue), see fyc.h
  float* chan_parts; int cs_rows, cs_slots;        // [tiles_m][cs_slots][N][2] = {sum, sum sq} of the stored values per (row tile, sample slot, channel)
  float* row_parts; int row_nparts;                // [M][row_nparts][2] = per row, per column tile {sum, sum sq}
  int tiles_m, tiles_n;
  int strip;  // > 0: tiles are walked in column strips of this many tiles (row-major inside a strip), see tile_coords
  int up_exact2; float up_sh, up_sw;   // nearest-upsample source mapping
  int colc;   // bias / colsum / tile-uniform rowbias may be fetched 16 B at a time and staged through LDS once per tile
  int rb_tile; // rowbias row is the same for every row of a tile (rows_per_batch % BM == 0): folded into the staged bias
  int rb_slots; // bf16 LINEAR wide epilogue: a tile touches up to this many rowbias rows (rows_per_batch % 16 == 0), staged through LDS next to the bias
  int splitk; float* ws;   // split-K (small M, long K): `splitk` work items per output tile, raw f32 partials to ws[splitk][M][N]
  int stagger; // 8-wave tiles: the upper half of the waves issues its DMAs between its two MFMA k-steps (fyc_set_tuning key 5 = 1: off)
  int wide;   // bf16 linear epilogue may use 16-B row accesses (N, ldo, ldr multiples of 8; bias/rowbias 16-B aligned)
  const char* zero;
  int res_acc;   // bf16 LINEAR wide epilogue: the residual tile is loaded INTO the accumulators before the K loop (see epilogue_linear_packed)
  float inv_seg_cols, inv_head_dim;   // HEADS epilogue: reciprocals for fdiv_small (host: fyc_gemm)
  int fast1;   // packed LINEAR epilogue: specialised pass 1 (fyc_set_tuning key 13 = 1: the generic one, A/B)
  int heads_pk; // head-split epilogue: the pack-first form (epilogue_heads_packed) - M <= 8192 only, see launch()
  int pre;     // round 6: the epilogue's per-row / per-column inputs are already in LDS (issue_consts in fyc_gemm_kernel), see pre_bytes()
  int phase_delay;   // round 6 (A/B, fyc_set_tuning key 11): every other block of an XCD starts this many x 1024 cycles late, see fyc_gemm_kernel
  unsigned long long* trace;   // timing builds (-DFYC_TRACE, tools/gemm_phase_probe.py): per-block s_memtime stamps; nullptr otherwise
};

// timing builds: waves 0 and 4 of a block append the shader clock to their list ([block][2][128] u64, zeroed by the host; the
// count lives in a register so that a stamp is one store and no load)
#ifdef FYC_TRACE
#define FYC_STAMP_DECL int fyc_trace_n = 0
#define FYC_STAMP(p, wave, lane)                                                                                         \
  do {                                                                                                                   \
    if ((p).trace != nullptr && ((wave) & 3) == 0 && (lane) == 0 && fyc_trace_n < 128)                                   \
      (p).trace[((size_t)blockIdx.x * 2 + ((wave) >> 2)) * 128 + fyc_trace_n] = __builtin_amdgcn_s_memtime();            \
    ++fyc_trace_n;                                                                                                       \
  } while (0)
// ... and inside the packed LINEAR epilogue (second half of the buffer, own counter): behind its first barrier, its inputs, pass 1, pass 2
#define FYC_STAMP_E(p, wave, lane, cnt)                                                                                  \
  do {                                                                                                                   \
    if ((p).trace != nullptr && ((wave) & 3) == 0 && (lane) == 0 && (cnt) < 128)                                         \
      (p).trace[65536 + ((size_t)blockIdx.x * 2 + ((wave) >> 2)) * 128 + (cnt)] = __builtin_amdgcn_s_memtime();          \
    ++(cnt);                                                                                                             \
  } while (0)
#else
#define FYC_STAMP_DECL
#define FYC_STAMP(p, wave, lane) do { } while (0)
#define FYC_STAMP_E(p, wave, lane, cnt) do { } while (0)
#endif

// Linear tile index (after the XCD remap) -> tile coordinates.  The 32 CUs of an XCD work on ~32 consecutive indices at
// any time.  With plain row-major order (n fastest) and many column tiles that window is 1-2 tile rows x 16-32 columns:
// per K slab the XCD's L2 then has to hold 2 A panels + 16-32 weight panels.  Walking column strips of `strip` tiles makes
// the window 8 rows x 4 columns: (8*BM + 4*BN) instead of (BM + 32*BN) rows per slab, i.e. up to 3x less L2 fill traffic
// for the wide FF1 layers (N = 2560 / 5120 / 10240).
__device__ __forceinline__ void tile_coords(const GemmP& p, int t, int& tile_m, int& tile_n) {
  if (p.strip <= 0) { tile_n = t % p.tiles_n; tile_m = t / p.tiles_n; return; }
  const int per_strip = p.strip * p.tiles_m;
  const int s = t / per_strip, r = t - s * per_strip;
  const int n0 = s * p.strip;
  const int w = min(p.strip, p.tiles_n - n0);        // the last strip may be narrower
  tile_m = r / w;
  tile_n = n0 + r - tile_m * w;
}

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  typedef bf16x8 Frag;
  __device__ static __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<f16_t> {
  typedef f16x8 Frag;
  __device__ static __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  typedef f32x4 Frag;
  // lane quad g holds k = 4*chunk + {0..3}; MFMA #j contracts element j of all four quads.
  __device__ static __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
    return c;
  }
};

// ---- 32x32x16 main loop (round 6) ----------------------------------------------------------------------------------------------------------
// v_mfma_f32_32x32x16_{bf16,f16} is a 32-cycle instruction: half as many matrix instructions per K tile as the 16x16x32 form, and every one
// covers twice as many issue cycles of the partner wave's DMA / LDS instructions (profiles/r03_mfma_fill_microbench.txt: a wave interleaving
// LDS reads gets 26 cycles per 16 cycles of matrix work with it against 41 with the 16-cycle form).  Used as D^T = W x^T like the 16x16 form:
// a lane of the 32x32 block holds output row m = lane % 32 and columns n = 8 q + 4 (lane / 32) + e for register 4 q + e.
// The epilogues are written for the 16x16 layout (lane = 16 g + r: row r, columns 4 g + e of a 16x16 block); acc32_to_acc16 re-sorts a block in
// registers with gfx950's v_permlane32_swap / v_permlane16_swap: (lane bit 5, lane bit 4, register bit q0) of the 32x32 layout are
// (h, row bit 4, column bit 3); the 16x16 layout wants (column bit 3, h, -) in the lane and row bit 4 in the register index - a 3-cycle
// of one register bit and two lane bits = the two swaps.  16 VALU instructions per 32x32 block, no LDS.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <typename T> struct Mma32;
template <> struct Mma32<bf16_t> {
  __device__ static __forceinline__ f32x16 mma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Mma32<f16_t> {
  __device__ static __forceinline__ f32x16 mma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mma32<float> {     // (never instantiated with MI = 32: the f32 parity mode keeps v_mfma_f32_16x16x4_f32)
  __device__ static __forceinline__ f32x16 mma(f32x4, f32x4, f32x16 c) { return c; }
};
// X = {X.lo, Y.lo}, Y = {X.hi, Y.hi} (halves of 32 lanes) / X = {X.r0, Y.r0, X.r2, Y.r2}, Y = {X.r1, Y.r1, X.r3, Y.r3} (rows of 16 lanes)
__device__ __forceinline__ void swap32(float& x, float& y) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]); y = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float& x, float& y) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]); y = __uint_as_float(r[1]);
}
// one 32x32 block -> its four 16x16 blocks o[im][jn] (rows 16 im.., columns 16 jn..)
__device__ __forceinline__ void acc32_to_acc16(const f32x16& c, f32x4& o00, f32x4& o01, f32x4& o10, f32x4& o11) {
#pragma unroll
  for (int q1 = 0; q1 < 2; ++q1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x = c[8 * q1 + e], y = c[8 * q1 + 4 + e];      // q0 = 0 / 1
      swap32(x, y);                                          // lane bit 5 <-> q0: registers now indexed by h
      swap16(x, y);                                          // lane bit 4 <-> h:  registers now indexed by row bit 4
      if (q1 == 0) { o00[e] = x; o10[e] = y; } else { o01[e] = x; o11[e] = y; }
    }
  }
}
// the inverse (residual tile loaded in the 16x16 layout -> 32x32 accumulators)
__device__ __forceinline__ void acc16_to_acc32(f32x16& c, const f32x4& o00, const f32x4& o01, const f32x4& o10, const f32x4& o11) {
#pragma unroll
  for (int q1 = 0; q1 < 2; ++q1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x = q1 == 0 ? o00[e] : o01[e], y = q1 == 0 ? o10[e] : o11[e];
      swap16(x, y);
      swap32(x, y);
      c[8 * q1 + e] = x; c[8 * q1 + 4 + e] = y;
    }
  }
}

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// internal epilogue id: LINEAR with the pointwise activation p.act compiled in.  A separate instantiation so that the
// hot-path LINEAR kernels do not carry the GELU code (it costs the 256x320 tile 80 B/lane of scratch).
constexpr int EPI_LINEAR_ACT = 3;

__device__ __forceinline__ float activate(float x, int act) {
  return act == FYC_ACT_GELU ? gelu_erf_f(x) : x / (1.0f + __expf(-1.702f * x));
}

// floor(n / d) for 0 <= n < 2^16, 1 <= d <= 2^12 from inv = 1.0f / d: (n + 0.5) / d is at least 0.5 / d away from every integer, the f32
// product is off by < 2^-7 of that.  3 VALU instructions where hipcc's 32-bit division by a run-time value is ~25: the head-split epilogue
// decomposed EVERY 16-byte chunk it stored with two such divisions (round 6: ~3 000 VALU instructions per lane and 256x320 tile).
__device__ __forceinline__ int fdiv_small(int n, float inv) { return (int)(((float)n + 0.5f) * inv); }

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// swizzle key of an LDS row: 128-B rows (8 chunks) use row&7; 64-B rows (4 chunks) use a 4-entry table over (row>>2)&3 -
// both make the ds_read_b128 of 16 consecutive rows x one chunk conflict-free under the real 16-lane service groups
template <int RB, int MI = 16> __device__ __forceinline__ int swz_key(int row) {
  // 32x32x16 fragments: a 16-lane service group of ds_read_b128 holds 16 DIFFERENT rows (8 even, 8 odd) x ONE chunk - (row >> 1) & 7 gives the
  // 8 rows of a parity 8 different chunk slots in every group ({0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}); row & 7 would pair rows 0 | 24, 12 | 20, ...
  if (MI == 32) return (row >> 1) & 7;
  if (RB == 128) return row & 7;
  return (0x78 >> (((row >> 2) & 3) * 2)) & 3;   // {0, 2, 3, 1}
}

// ---- output statistics (fused GroupNorm / LayerNorm statistics passes) ----------------------------------------------------
// LDS accumulators of one output tile, placed behind the column constants: cacc[stat_arrays][BN][2] per column and racc[BM][2]
// per row.  A tile of BM rows touches at most STAT_SLOTS samples (host: cs_rows % 16 == 0, fyc_gemm_stat_layout).
//
// The column sums are ORDER-FREE (bitwise repeatable, no atomics): every (wave row wm, sample slot) pair that occurs in a tile
// owns one accumulator array, index wm + slot (rows, hence both wm and slot, only grow, so the sum is unique; at most
// WGM + STAT_SLOTS - 1 pairs), ONE wave writes a given (array, column) with plain stores in program order, and stats_flush adds
// the arrays of a slot in wave-row order.  (Round 3 used LDS float atomics here: the order the waves arrived in moved the last
// bits of the GroupNorm statistics from run to run.)  The per-row sums (row_parts, off in the engine) still use LDS atomics.
constexpr int STAT_SLOTS = 4;
constexpr int RB_SLOTS = 4;     // rowbias rows a tile may touch in the packed LINEAR epilogue (GemmP::rb_slots)
template <int WGM> constexpr int stat_arrays() { return WGM + STAT_SLOTS - 1; }
template <int BM, int BN, int WGM> constexpr int stat_bytes() { return (stat_arrays<WGM>() * BN * 2 + BM * 2) * 4; }

__device__ __forceinline__ void lds_add(float* p, float v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int BM, int BN, int WGM, int NT>
__device__ __forceinline__ void stats_zero(float* cacc, int tid) {
  for (int i = tid; i < stat_arrays<WGM>() * BN * 2 + BM * 2; i += NT) cacc[i] = 0.f;
}

// after every wave finished accumulating: the arrays of a sample slot are added in wave-row order and stored as the tile's partial
// sums with plain stores (no device atomics: 640 f64 atomics per tile, 327 K per launch, cost the 64x64 convs +77 us each);
// fyc_chan_stats_reduce adds the row tiles of a sample up
template <int BM, int BN, int WGM, int NT>
__device__ __forceinline__ void stats_flush(const GemmP& p, const float* cacc, int tile_m, int tile_n, int tid) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const float* racc = cacc + stat_arrays<WGM>() * BN * 2;
  if (p.chan_parts != nullptr) {
    constexpr int R = BM / WGM;
    const int first = (tile_m * BM) / p.cs_rows;
    int lo[WGM], hi[WGM];                              // sample slots wave row w touches
#pragma unroll
    for (int w = 0; w < WGM; ++w) {
      lo[w] = (tile_m * BM + w * R) / p.cs_rows - first;
      hi[w] = (tile_m * BM + w * R + R - 1) / p.cs_rows - first;
    }
    for (int i = tid; i < p.cs_slots * BN; i += NT) {
      const int slot = i / BN, col = i - slot * BN, n = tile_n * BN + col;
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < WGM; ++w) {
        if (slot >= lo[w] && slot <= hi[w]) {
          const float2 v = *reinterpret_cast<const float2*>(cacc + ((w + slot) * BN + col) * 2);
          s += v.x; q += v.y;
        }
      }
      if (n < p.N) *reinterpret_cast<float2*>(p.chan_parts + (((long long)tile_m * p.cs_slots + slot) * p.N + n) * 2) = make_float2(s, q);
    }
  }
  if (p.row_parts != nullptr) {
    for (int r = tid; r < BM; r += NT) {
      const int m = tile_m * BM + r;
      if (m < p.M) *reinterpret_cast<float2*>(p.row_parts + ((long long)m * p.row_nparts + tile_n) * 2) = *reinterpret_cast<const float2*>(racc + 2 * r);
    }
  }
}

// {mean, rstd} of row m for the folded LayerNorm: either stored as such (fyc_row_stats) or derived from the producer's partial sums
__device__ __forceinline__ void ln_row(const GemmP& p, int m, float& mu, float& rs) {
  if (p.ln_nparts <= 0) {
    const float2 ms = *reinterpret_cast<const float2*>(p.ln_stats + 2ll * m);
    mu = ms.x; rs = ms.y;
    return;
  }
  const float2* q = reinterpret_cast<const float2*>(p.ln_stats) + (long long)m * p.ln_nparts;
  float2 t = q[0];
  for (int i = 1; i < p.ln_nparts; ++i) { const float2 u = q[i]; t.x += u.x; t.y += u.y; }
  const float inv = 1.0f / (float)p.K;
  mu = t.x * inv;
  rs = rsqrtf(fmaxf(t.y * inv - mu * mu, 0.f) + p.ln_eps);
}

template <typename T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<bf16_t>(float v) { return round_through<bf16_t>(v); }
template <> __device__ __forceinline__ float round_to<f16_t>(float v) { return round_through<f16_t>(v); }

// Column constants of one output tile, staged through LDS once per tile: colc[0..BN) = bias (+ the rowbias row when it is the
// same for the whole tile), colc[BN..2BN) = LayerNorm column sums.  Every global load in the epilogue is followed by an
// s_waitcnt that also drains the next tile's K-tile DMA (loads return in order) and, with one block per CU, nothing hides
// that latency - so the constants are fetched by BN/4 lanes in one go instead of once per (row block, column tile) by all.
template <int BN, bool LN>
__device__ __forceinline__ void stage_col_constants(const GemmP& p, float* colc, int tile_m_row0, int tile_n, int tid, float* rbc = nullptr) {
  const int t4 = tid * 4;
  if (t4 < BN) {
    const int n = tile_n * BN + t4;
    const bool ok = n < p.N;
    f32x4 b = (ok && p.bias) ? *reinterpret_cast<const f32x4*>(p.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
    if (ok && p.rb_tile) {
      const f32x4 r4 = *reinterpret_cast<const f32x4*>(p.rowbias + (long long)(tile_m_row0 / p.rows_per_batch) * p.ldrb + n);
      b[0] += r4[0]; b[1] += r4[1]; b[2] += r4[2]; b[3] += r4[3];
    }
    *reinterpret_cast<f32x4*>(colc + t4) = b;
    *reinterpret_cast<f32x4*>(colc + BN + t4) = (LN && ok && p.ln_stats) ? *reinterpret_cast<const f32x4*>(p.ln_colsum + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
    if (rbc != nullptr) {                    // the rowbias rows of the (up to RB_SLOTS) batch elements this tile touches
      const int b0 = tile_m_row0 / p.rows_per_batch, nb = (p.M + p.rows_per_batch - 1) / p.rows_per_batch;
      for (int sl = 0; sl < p.rb_slots; ++sl)
        *reinterpret_cast<f32x4*>(rbc + sl * BN + t4) = (ok && b0 + sl < nb) ? *reinterpret_cast<const f32x4*>(p.rowbias + (long long)(b0 + sl) * p.ldrb + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// ---- epilogue inputs pre-staged by DMA (round 6) -------------------------------------------------------------------------------------------
// profiles/r06_gemm_epilogue_ablation.txt: of the 103 us the epilogues of the 131072x960x320 projection cost (178 us per launch), 54 are
// "pass 1" - and that is not its arithmetic but two exposed round trips: the column constants (stage_col_constants) and the LayerNorm row
// statistics (ln_row) are global loads, loads retire in order, and the next tile's first K-tile DMA is in flight in front of them - every
// tile of every launch paid the DMA's HBM latency plus its own.  The wide epilogues of the 2-deep-ring kernels now find their inputs in LDS:
// in front of a tile's K loop, behind a barrier (every wave has left the previous tile's epilogue), the waves fetch, by global_load_lds,
//   [BM] {mean, rstd} | [BN] bias | [BN] LayerNorm column sums | [<= RB_SLOTS][BN] row-bias rows
// into a region behind the ring; the vmcnt(0) + barrier of the tile's first K-loop iteration retires them.  No global load and no vmcnt wait is left in those epilogues.
template <int BM, int BN> constexpr int pre_bytes() { return BM * 8 + (2 + RB_SLOTS) * BN * 4; }
template <int BM> __device__ __forceinline__ void pre_ln_row(const char* pre, int row_in_tile, float& mu, float& rs) {
  const float2 ms = *reinterpret_cast<const float2*>(pre + row_in_tile * 8);
  mu = ms.x; rs = ms.y;
}

// ---- bf16 LINEAR epilogue, "pack first" (round 4) ---------------------------------------------------------------------------------
// s_memtime stamps (profiles/r04_gemm_phase_trace.txt) showed the epilogue of a 256x320 tile with residual + output statistics at
// 125 000 cycles against 175 000 for its whole 45-K-tile loop: 160 live accumulators + two sets of prefetched residual rows + the
// statistics registers spilled (the same epilogue on the 128x320 tile, no spills: 34 000 per 128 rows).  This version never holds
// more than the accumulators:
//   * the RESIDUAL is not an epilogue input any more: the kernel loads the residual tile into the accumulators before the K loop
//     (GemmP::res_acc; MFMA layout, 8 bytes per lane) - it rides under the K loop instead of costing exposed round trips here, and
//     the sum stays one f32 accumulation with a single rounding;
//   * there is no global load left in it: bias, LayerNorm column sums and the time-embedding / positional rows (also when they change
//     inside a tile) are staged through LDS once per tile; fyc_gemm() sends the two combinations this does not cover (a residual next
//     to a LayerNorm fold, row-bias groups that are not multiples of 16 rows or too many per tile) to the narrow per-lane epilogue;
//   * pass 1 turns the accumulators into the final values (LayerNorm fold, bias / time-embedding row, scale) and PACKS them to
//     bf16 pairs in place: 160 registers become 80;
//   * pass 2 transposes 16 rows x all of the wave's columns per step through a bf16 staging slice (4 steps per wave for the
//     256x320 tile instead of 8 half-width f32 ones) and stores / accumulates the statistics from 16-byte row segments as before.
template <typename T, int BM, int BN, int WGM, int WGN, int STG_BYTES, int MODE>
__device__ __forceinline__ void epilogue_linear_packed(const GemmP& p, f32x4 (&acc)[BM / WGM / 16][BN / WGN / 16], int tile_m, int tile_n,
                                                       long long bz, char* stg_stage, int wave, int lane, const char* pre, int& ecnt) {
  static_assert(sizeof(T) == 2, "16-bit outputs only (bf16 / f16)");
  constexpr int WTM = BM / WGM / 16, WTN = BN / WGN / 16;
#if defined(FYC_ABL_EPI) && FYC_ABL_EPI == 3      // TIMING BUILDS (wrong results), tools/gpu_r6.sh epi_ablation: 3 = no epilogue at all
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j) asm volatile("" :: "v"(acc[i][j][0]), "v"(acc[i][j][1]), "v"(acc[i][j][2]), "v"(acc[i][j][3]));
  return;
#endif
  constexpr bool LN = (MODE == FYC_GEMM_PLAIN);
  constexpr int PITCH = WTN * 32 + 16;               // bytes per staged row: the wave's WTN * 16 bf16 columns + 16 (keeps the 16-byte reads aligned)
  constexpr int CPR = WTN * 2;                       // 16-byte chunks per staged row
  constexpr int RPP = 64 / CPR;                      // rows per store instruction; a lane keeps ONE chunk and walks rows lrow, lrow + RPP, ...
  constexpr int NQ = (16 + RPP - 1) / RPP;           // store instructions per 16-row block
  // a wave's staging slice: 16 staged rows, and at least the 64 lanes x 8 floats the column-statistics reduction parks in it
  constexpr int SLICE = 16 * PITCH > 2048 ? 16 * PITCH : 2048;
  static_assert(WGM * WGN * SLICE + 2 * BN * 4 <= STG_BYTES, "staging + column constants must fit in one ring stage");
  constexpr bool STATS_FIT = WGM * WGN * SLICE + 2 * BN * 4 + stat_bytes<BM, BN, WGM>() <= STG_BYTES;
  const int wm = wave / WGN, wn = wave % WGN;
  const int g = lane >> 4, r16 = lane & 15;
  T* O = reinterpret_cast<T*>(p.out) + bz * p.stride_o;   // batched problems (materialised attention of the VAE): one output per batch element
  __builtin_amdgcn_s_barrier();                      // every wave is done reading the stage we reuse
  char* stg = stg_stage + wave * SLICE;
  const bool use_pre = pre != nullptr;                 // (wave-uniform, the same for every tile of a launch)
  float* colc = reinterpret_cast<float*>(stg_stage + WGM * WGN * SLICE);   // [2][BN], see stage_col_constants
  float* cacc = colc + 2 * BN;
  float* racc = cacc + stat_arrays<WGM>() * BN * 2;
  const bool do_cs = STATS_FIT && p.chan_parts != nullptr, do_rp = STATS_FIT && p.row_parts != nullptr;
  // rowbias rows that change inside the tile (per-frame rows at the 8x8 level): behind the statistics accumulators
  constexpr bool RB_FIT = WGM * WGN * SLICE + 2 * BN * 4 + stat_bytes<BM, BN, WGM>() + RB_SLOTS * BN * 4 <= STG_BYTES;
  float* rbc = (RB_FIT && p.rb_slots > 0) ? racc + BM * 2 : nullptr;
  if (do_cs || do_rp) stats_zero<BM, BN, WGM, WGM * WGN * 64>(cacc, wave * 64 + lane);
  if (use_pre) {                                       // inputs already in LDS: [bias | colsum | row-bias rows] behind the row statistics
    colc = reinterpret_cast<float*>(const_cast<char*>(pre) + BM * 8);
    rbc = p.rowbias != nullptr ? colc + 2 * BN : nullptr;
    if (do_cs || do_rp) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }   // the zeroed accumulators
  } else {
    stage_col_constants<BN, LN>(p, colc, tile_m * BM, tile_n, wave * 64 + lane, rbc);
  }
  const int n_w0 = tile_n * BN + wn * WTN * 16;      // first GEMM column of this wave
  const int nl_w0 = wn * WTN * 16;                   // ... inside the tile
  // ---- pass 1: final values, packed in place --------------------------------------------------------------------------------------
  // (column block outermost: the column constants of a block are read once and die with it - with the row block outermost the
  // compiler kept all 2 x WTN constant vectors alive across the row blocks, 80 registers on the 256x320 tile, and spilled)
  u32x2 pk[WTM][WTN];
  float mu[WTM], rs[WTM];
  int rbo[WTM];                                      // float offset of row block i's rowbias row inside rbc (wave-uniform)
#pragma unroll
  for (int i = 0; i < WTM; ++i) {
    const int m0 = tile_m * BM + (wm * WTM + i) * 16;
    mu[i] = 0.f; rs[i] = 1.f;
    if (LN && p.ln_stats) {
      if (use_pre) pre_ln_row<BM>(pre, (wm * WTM + i) * 16 + r16, mu[i], rs[i]);      // (rows beyond M: {0, 0} from the zero page, never stored)
      else if (m0 + r16 < p.M) ln_row(p, m0 + r16, mu[i], rs[i]);
    }
    rbo[i] = rbc != nullptr ? (m0 / p.rows_per_batch - (tile_m * BM) / p.rows_per_batch) * BN : 0;
  }
  // (scheduling fences: left alone, hipcc hoists the constant reads of ALL column blocks to the top - 80 registers - and sinks the
  // packing into pass 2, i.e. keeps all 160 accumulators alive to the end: the spills this function exists to remove.  The constants
  // of block j + 1 are read while block j is packed.)
  //
  // Round 6 (profiles/r06_gemm_epilogue_ablation.txt): this pass was 54 of the 178 us of the 131072x960x320 projection - not its loads but
  // its instruction stream: per 4 values two wave-uniform branches (LayerNorm? row bias?), an LDS read with its own lgkmcnt(0), 10 VALU
  // for the LayerNorm fold, 4 adds, 2 scale multiplies.  The common combinations now run specialised copies chosen ONCE per tile: unit
  // out_scale, the row-bias row the same for all row blocks of the wave (one read per column block, folded into the bias), the LayerNorm
  // fold as two packed FMAs per value pair: out = fma(acc, rstd, fma(-rstd * mean, colsum, bias')).  4-6 VALU per 4 values instead of ~20.
  FYC_STAMP_E(p, wave, lane, ecnt);
  bool rb_same = true;
#pragma unroll
  for (int i = 1; i < WTM; ++i) rb_same = rb_same && rbo[i] == rbo[0];
  const bool has_ln = LN && p.ln_stats != nullptr;
  auto pass1_fast = [&](auto ln_c, auto rb_c) __attribute__((always_inline)) {
    constexpr bool HAS_LN = decltype(ln_c)::value, HAS_RB = decltype(rb_c)::value;
    f32x2 rs2[WTM], nm2[WTM];
#pragma unroll
    for (int i = 0; i < WTM; ++i) { rs2[i] = (f32x2){rs[i], rs[i]}; nm2[i] = (f32x2){-rs[i] * mu[i], -rs[i] * mu[i]}; }
    const float* rb0 = HAS_RB ? rbc + rbo[0] : colc;
    f32x4 b4n = *reinterpret_cast<const f32x4*>(colc + nl_w0 + g * 4), s4n = (f32x4){0.f, 0.f, 0.f, 0.f}, r4n = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (HAS_LN) s4n = *reinterpret_cast<const f32x4*>(colc + BN + nl_w0 + g * 4);
    if (HAS_RB) r4n = *reinterpret_cast<const f32x4*>(rb0 + nl_w0 + g * 4);
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
      const int nl = nl_w0 + j * 16 + g * 4;
      f32x4 b4 = b4n;
      const f32x4 s4 = s4n;
      if (HAS_RB) { b4[0] += r4n[0]; b4[1] += r4n[1]; b4[2] += r4n[2]; b4[3] += r4n[3]; }
      __builtin_amdgcn_sched_barrier(0);
      if (j + 1 < WTN) {
        b4n = *reinterpret_cast<const f32x4*>(colc + nl + 16);
        if (HAS_LN) s4n = *reinterpret_cast<const f32x4*>(colc + BN + nl + 16);
        if (HAS_RB) r4n = *reinterpret_cast<const f32x4*>(rb0 + nl + 16);
      }
      const f32x2 b01 = {b4[0], b4[1]}, b23 = {b4[2], b4[3]}, s01 = {s4[0], s4[1]}, s23 = {s4[2], s4[3]};
#pragma unroll
      for (int i = 0; i < WTM; ++i) {
        f32x2 v01 = {acc[i][j][0], acc[i][j][1]}, v23 = {acc[i][j][2], acc[i][j][3]};
        if (HAS_LN) {
          v01 = __builtin_elementwise_fma(v01, rs2[i], __builtin_elementwise_fma(nm2[i], s01, b01));
          v23 = __builtin_elementwise_fma(v23, rs2[i], __builtin_elementwise_fma(nm2[i], s23, b23));
        } else {
          v01 = v01 + b01; v23 = v23 + b23;
        }
        unsigned lo = Pair16<T>::pack(v01[0], v01[1]), hi = Pair16<T>::pack(v23[0], v23[1]);
        asm volatile("" : "+v"(lo), "+v"(hi));       // pin the conversion HERE (see below)
        pk[i][j] = (u32x2){lo, hi};
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (p.out_scale == 1.0f && (rbc == nullptr || rb_same) && p.fast1) {
    if (has_ln) { if (rbc != nullptr) pass1_fast(std::true_type{}, std::true_type{}); else pass1_fast(std::true_type{}, std::false_type{}); }
    else { if (rbc != nullptr) pass1_fast(std::false_type{}, std::true_type{}); else pass1_fast(std::false_type{}, std::false_type{}); }
  } else {
  f32x4 b4n = *reinterpret_cast<const f32x4*>(colc + nl_w0 + g * 4), s4n = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (LN && p.ln_stats) s4n = *reinterpret_cast<const f32x4*>(colc + BN + nl_w0 + g * 4);
#pragma unroll
  for (int j = 0; j < WTN; ++j) {
    const int nl = nl_w0 + j * 16 + g * 4;
    const f32x4 b4 = b4n, s4 = s4n;
    __builtin_amdgcn_sched_barrier(0);
    if (j + 1 < WTN) {
      b4n = *reinterpret_cast<const f32x4*>(colc + nl + 16);
      if (LN && p.ln_stats) s4n = *reinterpret_cast<const f32x4*>(colc + BN + nl + 16);
    }
#pragma unroll
    for (int i = 0; i < WTM; ++i) {
      f32x4 v = acc[i][j];
      if (LN && p.ln_stats) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = rs[i] * (v[r] - mu[i] * s4[r]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += b4[r];
      if (rbc != nullptr) {                          // wave-uniform branch, LDS only
        const f32x4 r4 = *reinterpret_cast<const f32x4*>(rbc + rbo[i] + nl);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += r4[r];
      }
      unsigned lo = Pair16<T>::pack(v[0] * p.out_scale, v[1] * p.out_scale), hi = Pair16<T>::pack(v[2] * p.out_scale, v[3] * p.out_scale);
      asm volatile("" : "+v"(lo), "+v"(hi));       // pin the conversion HERE (LLVM otherwise sinks it to the staging write of pass 2 and carries 4 floats instead of 2 words)
      pk[i][j] = (u32x2){lo, hi};
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  }
#if defined(FYC_ABL_EPI) && FYC_ABL_EPI == 2      // 2 = pass 1 only (final values packed in registers, nothing staged or stored)
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j) asm volatile("" :: "v"(pk[i][j][0]), "v"(pk[i][j][1]));
  return;
#endif
  FYC_STAMP_E(p, wave, lane, ecnt);
  // ---- pass 2: 16 rows x (WTN * 16) columns per step through the wave's staging slice ---------------------------------------------
  const int lrow = lane / CPR, lch = lane - lrow * CPR;
  const bool lact = lrow < RPP;
  const int n_lane = n_w0 + lch * 8;                 // this lane's 8 output columns
  const int first_sample = do_cs ? (tile_m * BM) / p.cs_rows : 0;
  const bool one_slot = p.cs_slots == 1;
  float cs8[8], cq8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) cs8[e] = cq8[e] = 0.f;
  int cur_slot = one_slot ? 0 : -1;
  // The wave's column sums of one sample slot go to the accumulator array this (wave row, slot) pair owns (see stats_flush): lanes
  // park their 8 sums in the idle staging slice ([lrow][lch][8] = lane * 8 floats), then one lane per column adds the rows of its
  // column in order and stores the total - no atomics, a handful of registers (round 2: a shuffle tree cost the K loop its registers).
  // Called where the slot changes (tiles that span samples: 8x8 / 12x12 / 24x24 frames) and after the last row block.
  auto flush_tree = [&](int slot) __attribute__((always_inline)) {
    float* red = reinterpret_cast<float*>(stg);
    float* own = cacc + (wm + slot) * (BN * 2);
    constexpr int NCOL = CPR * 8, ROWS_LIVE = RPP < 16 ? RPP : 16;
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      if (lact) {
        const float* src = ph ? cq8 : cs8;
        *reinterpret_cast<f32x4*>(red + lane * 8) = (f32x4){src[0], src[1], src[2], src[3]};
        *reinterpret_cast<f32x4*>(red + lane * 8 + 4) = (f32x4){src[4], src[5], src[6], src[7]};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
      for (int c = lane; c < NCOL; c += 64) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < ROWS_LIVE; ++r) t += red[r * NCOL + c];
        own[(nl_w0 + c) * 2 + ph] = t;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) cs8[e] = cq8[e] = 0.f;
  };
#pragma unroll
  for (int i = 0; i < WTM; ++i) {
    if (do_cs && !one_slot) {                          // sample slot of this 16-row block (wave-uniform); the staging slice is idle here
      const int slot = (tile_m * BM + (wm * WTM + i) * 16) / p.cs_rows - first_sample;
      if (slot != cur_slot) { if (cur_slot >= 0) flush_tree(cur_slot); cur_slot = slot; }
    }
#pragma unroll
    for (int j = 0; j < WTN; ++j) *reinterpret_cast<u32x2*>(stg + r16 * PITCH + j * 32 + g * 8) = pk[i][j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // round 6: ALL reads of the block first, by every lane (rows clamped into the staged block), ONE wait, then the predicated stores -
    // inside the `if (live)` each of the NQ reads was followed by its own lgkmcnt(0): 24 exposed LDS round trips per 256x320 tile and wave
    // (in groups of at most QG reads: all six of the 256x320 tile at once cost its kernels 60 more bytes of scratch and the convolutions 1.5 %)
    // (QG = 1: read, wait, store - the round-5 order.  Batching the reads of a block - all NQ, or groups of 3 - measured 5-7 % faster on the
    // short-K linears in the cold-operand probe and 0.1 ms per DDIM step SLOWER in the pipeline, where the convolutions' epilogues lost more than
    // the linears' gained: profiles/r06_gemm_epilogue_ab.txt.  -DFYC_P2_QG=n rebuilds the batched form.)
#ifdef FYC_P2_QG
    constexpr int QG = FYC_P2_QG;
#else
    constexpr int QG = 1;
#endif
#pragma unroll
    for (int q0 = 0; q0 < NQ; q0 += QG) {
    u32x4 v4q[QG];
#pragma unroll
    for (int qq = 0; qq < QG; ++qq) {
      const int q = q0 + qq;
      const int rowc = (q * RPP + lrow) < 15 ? (q * RPP + lrow) : 15;
      if (q < NQ) v4q[qq] = *reinterpret_cast<const u32x4*>(stg + rowc * PITCH + lch * 16);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int qq = 0; qq < QG; ++qq) {
      const int q = q0 + qq;
      if (q >= NQ) continue;
      const int row = q * RPP + lrow;
      const int m = tile_m * BM + (wm * WTM + i) * 16 + row;
      float rsum = 0.f, rsq = 0.f;
      const bool live = lact && row < 16 && m < p.M && n_lane < p.N;
      if (live) {
        const u32x4 v4 = v4q[qq];
#if defined(FYC_ABL_EPI) && FYC_ABL_EPI == 1      // 1 = everything but the global stores
        asm volatile("" :: "v"(v4[0]), "v"(v4[1]), "v"(v4[2]), "v"(v4[3]));
#else
        *reinterpret_cast<u32x4*>(O + (long long)m * p.ldo + n_lane) = v4;   // (nt stores: same GEMM time, consumers +10 %: profiles/r04_epilogue_nontemporal_ab.txt)
#endif
        if (do_cs || do_rp) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x0 = Pair16<T>::lo(v4[e]), x1 = Pair16<T>::hi(v4[e]);
            if (do_cs) {
              cs8[2 * e] += x0; cq8[2 * e] = __builtin_fmaf(x0, x0, cq8[2 * e]);
              cs8[2 * e + 1] += x1; cq8[2 * e + 1] = __builtin_fmaf(x1, x1, cq8[2 * e + 1]);
            }
            if (do_rp) { rsum += x0 + x1; rsq = __builtin_fmaf(x0, x0, __builtin_fmaf(x1, x1, rsq)); }
          }
        }
      }
      if (do_rp && live) {
        float* dst = racc + ((wm * WTM + i) * 16 + row) * 2;
        lds_add(dst, rsum); lds_add(dst + 1, rsq);
      }
    }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  FYC_STAMP_E(p, wave, lane, ecnt);
  if (do_cs) flush_tree(cur_slot);
  if (do_cs || do_rp) stats_flush<BM, BN, WGM, WGM * WGN * 64>(p, cacc, tile_m, tile_n, wave * 64 + lane);
}

// ---- head-split epilogue, "pack first" (round 6) ---------------------------------------------------------------------------------------------
// q | k | V^T of the spatial attentions and the q of the cross attentions: 55 tile rounds per DDIM step whose round-2 epilogue staged f32 in two
// half-width passes and decomposed every chunk it stored (profiles/r06_gemm_epilogue_ablation.txt: 18-20 us per 256x320 tile round against 15
// for LINEAR and 7 for GEGLU).  Same scheme as epilogue_linear_packed: pass 1 packs the final values (LayerNorm fold as packed FMAs, bias);
// pass 2 stages 16 rows x the wave's columns as 16-bit words and
//   * a lane keeps ONE 8-column chunk for the whole tile - its (segment, head, offset) is computed once, not per store - and writes its rows
//     of the q / k style segments as 16-byte runs along the head dim;
//   * the transposed segments (V^T [b*H][d][ld]) gather 8 consecutive tokens of one column from the staged rows (8 ds_read_u16) and leave as
//     16-byte runs along the token axis.
// Host guarantees (GemmP::wide with FYC_EPI_HEADS): head_dim % 8 == 0, tokens % 16 == 0, transposed pitches % 8 == 0, 16-byte aligned bases.
template <typename T, int BM, int BN, int WGM, int WGN, int STG_BYTES, int MODE>
__device__ __forceinline__ void epilogue_heads_packed(const GemmP& p, f32x4 (&acc)[BM / WGM / 16][BN / WGN / 16], int tile_m, int tile_n,
                                                      char* stg_stage, int wave, int lane, const char* pre) {
  static_assert(sizeof(T) == 2, "16-bit outputs only (bf16 / f16)");
  constexpr int WTM = BM / WGM / 16, WTN = BN / WGN / 16;
  constexpr bool LN = (MODE == FYC_GEMM_PLAIN);
  constexpr int PITCH = WTN * 32 + 16, CPR = WTN * 2, RPP = 64 / CPR, NQ = (16 + RPP - 1) / RPP;
  constexpr int SLICE = 16 * PITCH;
  static_assert(WGM * WGN * SLICE + 2 * BN * 4 <= STG_BYTES, "staging + column constants must fit in one ring stage");
  const int wm = wave / WGN, wn = wave % WGN;
  const int g = lane >> 4, r16 = lane & 15;
  __builtin_amdgcn_s_barrier();                      // every wave is done reading the stage we reuse
  char* stg = stg_stage + wave * SLICE;
  float* colc = reinterpret_cast<float*>(stg_stage + WGM * WGN * SLICE);
  if (pre != nullptr) colc = reinterpret_cast<float*>(const_cast<char*>(pre) + BM * 8);
  else stage_col_constants<BN, LN>(p, colc, tile_m * BM, tile_n, wave * 64 + lane);
  const bool has_ln = LN && p.ln_stats != nullptr;
  f32x2 rs2[WTM], nm2[WTM];
#pragma unroll
  for (int i = 0; i < WTM; ++i) {
    float mu = 0.f, rs = 1.f;
    if (has_ln) {
      if (pre != nullptr) pre_ln_row<BM>(pre, (wm * WTM + i) * 16 + r16, mu, rs);
      else if (tile_m * BM + (wm * WTM + i) * 16 + r16 < p.M) ln_row(p, tile_m * BM + (wm * WTM + i) * 16 + r16, mu, rs);
    }
    rs2[i] = (f32x2){rs * p.out_scale, rs * p.out_scale}; nm2[i] = (f32x2){-rs * mu, -rs * mu};      // out = fma(acc, rs * scale, scale * fma(-rs mu, colsum, bias))
  }
  const int nl_w0 = wn * WTN * 16, n_w0 = tile_n * BN + nl_w0;
  const f32x2 sc2 = {p.out_scale, p.out_scale};
  // ---- pass 1 ------------------------------------------------------------------------------------------------------------------------
  u32x2 pk[WTM][WTN];
  auto pass1 = [&](auto ln_c) __attribute__((always_inline)) {
    constexpr bool HAS_LN = decltype(ln_c)::value;
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
      const int nl = nl_w0 + j * 16 + g * 4;
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(colc + nl);
      f32x4 s4 = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (HAS_LN) s4 = *reinterpret_cast<const f32x4*>(colc + BN + nl);
      __builtin_amdgcn_sched_barrier(0);
      const f32x2 b01 = {b4[0], b4[1]}, b23 = {b4[2], b4[3]}, s01 = {s4[0], s4[1]}, s23 = {s4[2], s4[3]};
#pragma unroll
      for (int i = 0; i < WTM; ++i) {
        f32x2 v01 = {acc[i][j][0], acc[i][j][1]}, v23 = {acc[i][j][2], acc[i][j][3]};
        if (HAS_LN) {
          v01 = __builtin_elementwise_fma(v01, rs2[i], sc2 * __builtin_elementwise_fma(nm2[i], s01, b01));
          v23 = __builtin_elementwise_fma(v23, rs2[i], sc2 * __builtin_elementwise_fma(nm2[i], s23, b23));
        } else {
          v01 = (v01 + b01) * sc2; v23 = (v23 + b23) * sc2;
        }
        unsigned lo = Pair16<T>::pack(v01[0], v01[1]), hi = Pair16<T>::pack(v23[0], v23[1]);
        asm volatile("" : "+v"(lo), "+v"(hi));
        pk[i][j] = (u32x2){lo, hi};
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (has_ln) pass1(std::true_type{}); else pass1(std::false_type{});
  // ---- pass 2 ------------------------------------------------------------------------------------------------------------------------
  // (a) this lane's 8-column chunk of the row-major segments, decomposed once
  const int lrow = lane / CPR, lch = lane - lrow * CPR;
  const int n_a = n_w0 + lch * 8;
  const int seg_a = fdiv_small(n_a < p.N ? n_a : 0, p.inv_seg_cols), cs_a = (n_a < p.N ? n_a : 0) - seg_a * p.seg_cols;
  const int hh_a = fdiv_small(cs_a, p.inv_head_dim), di_a = cs_a - hh_a * p.head_dim;
  const bool act_a = lrow < RPP && n_a < p.N && !p.seg_transposed[seg_a];
  T* const S_a = reinterpret_cast<T*>(p.seg_out[seg_a]);
  // (b) the (column, 8-token half) pairs of the transposed segments this lane gathers: u = lane + 64 k over the wave's WTN * 16 columns x 2.
  // One word per pair: segment << 28 | head << 16 | offset in the head, -1 = nothing to do (registers: this epilogue runs beside 80 packed
  // accumulators at the 256-register cap)
  constexpr int NU = (WTN * 32 + 63) / 64;
  int dec_b[NU];
  bool any_b = false;
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    const int u = lane + 64 * k;
    const int n = n_w0 + (u >> 1);
    const bool in = u < WTN * 32 && n < p.N;
    const int seg = fdiv_small(in ? n : 0, p.inv_seg_cols), cs = (in ? n : 0) - seg * p.seg_cols;
    const int hh = fdiv_small(cs, p.inv_head_dim), di = cs - hh * p.head_dim;
    dec_b[k] = (in && p.seg_transposed[seg]) ? ((seg << 28) | (hh << 16) | di) : -1;
    any_b = any_b || dec_b[k] != -1;
  }
  const bool wave_b = __builtin_amdgcn_ballot_w64(any_b) != 0;
#pragma unroll
  for (int i = 0; i < WTM; ++i) {
    const int m0 = tile_m * BM + (wm * WTM + i) * 16;            // first of the 16 token rows of this block (same batch element: tokens % 16 == 0)
    const int b = m0 / p.tokens, tok0 = m0 - b * p.tokens;
#pragma unroll
    for (int j = 0; j < WTN; ++j) *reinterpret_cast<u32x2*>(stg + r16 * PITCH + j * 32 + g * 8) = pk[i][j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (m0 < p.M) {
      if (act_a) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int row = q * RPP + lrow;
          if (row < 16) {
            const u32x4 v4 = *reinterpret_cast<const u32x4*>(stg + row * PITCH + lch * 16);
            *reinterpret_cast<u32x4*>(S_a + ((long long)(b * p.heads + hh_a) * p.tokens + tok0 + row) * p.head_dim + di_a) = v4;
          }
        }
      }
      if (wave_b) {
#pragma unroll
        for (int k = 0; k < NU; ++k) {
          if (dec_b[k] != -1) {
            const int half = lane & 1, seg = dec_b[k] >> 28, hh = (dec_b[k] >> 16) & 0xfff, di = dec_b[k] & 0xffff;
            T* const Sb = reinterpret_cast<T*>(seg == 0 ? p.seg_out[0] : seg == 1 ? p.seg_out[1] : p.seg_out[2]);
            const int ldb = seg == 0 ? p.seg_ld[0] : seg == 1 ? p.seg_ld[1] : p.seg_ld[2];
            const char* src = stg + (half * 8) * PITCH + ((lane + 64 * k) >> 1) * 2;
            unsigned short w[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) w[r] = *reinterpret_cast<const unsigned short*>(src + r * PITCH);
            const u32x4 v4 = {(unsigned)w[0] | ((unsigned)w[1] << 16), (unsigned)w[2] | ((unsigned)w[3] << 16), (unsigned)w[4] | ((unsigned)w[5] << 16),
                              (unsigned)w[6] | ((unsigned)w[7] << 16)};
            *reinterpret_cast<u32x4*>(Sb + ((long long)(b * p.heads + hh) * p.head_dim + di) * ldb + tok0 + half * 8) = v4;
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// ---- GEGLU epilogue, "pack first" (round 6) ---------------------------------------------------------------------------------------------
// The feed-forward projections of levels 1-3 (N = 5120 / 10240: 8 / 4 tile rounds per launch, 120 of the 533 tile rounds of a DDIM step) ran the
// round-1 epilogue: every 16-row block was gated in f32, staged as f32 in two half-width passes, read back and packed, each step behind its own
// lgkmcnt(0).  Same scheme as epilogue_linear_packed: pass 1 gates ALL blocks (value block 2 jo, gate block 2 jo + 1 of the wave: LayerNorm fold
// as packed FMAs, the packed polynomial gate of fyc_common.h::geglu_pair) and packs them to 16-bit pairs in place - WTN / 2 x WTM x 2 registers;
// pass 2 stages 16 rows x the wave's WTN * 8 output columns as 16-bit words, reads the block back in one go and stores 16-byte row segments.
template <typename T, int BM, int BN, int WGM, int WGN, int STG_BYTES, int MODE>
__device__ __forceinline__ void epilogue_geglu_packed(const GemmP& p, f32x4 (&acc)[BM / WGM / 16][BN / WGN / 16], int tile_m, int tile_n,
                                                      long long bz, char* stg_stage, int wave, int lane, const char* pre) {
  static_assert(sizeof(T) == 2, "16-bit outputs only (bf16 / f16)");
  constexpr int WTM = BM / WGM / 16, WTN = BN / WGN / 16, OT = WTN / 2;
  constexpr bool LN = (MODE == FYC_GEMM_PLAIN);
  constexpr int PITCH = OT * 32 + 16;                // bytes per staged row: OT * 16 output columns of 2 bytes + 16
  constexpr int CPR = OT * 2;                        // 16-byte chunks per staged row
  constexpr int RPP = 64 / CPR;                      // rows per store instruction
  constexpr int NQ = (16 + RPP - 1) / RPP;
  constexpr int SLICE = 16 * PITCH;
  static_assert(WGM * WGN * SLICE + 2 * BN * 4 <= STG_BYTES, "staging + column constants must fit in one ring stage");
  const int wm = wave / WGN, wn = wave % WGN;
  const int g = lane >> 4, r16 = lane & 15;
  T* O = reinterpret_cast<T*>(p.out) + bz * p.stride_o;
  __builtin_amdgcn_s_barrier();                      // every wave is done reading the stage we reuse
  char* stg = stg_stage + wave * SLICE;
  float* colc = reinterpret_cast<float*>(stg_stage + WGM * WGN * SLICE);
  if (pre != nullptr) colc = reinterpret_cast<float*>(const_cast<char*>(pre) + BM * 8);      // pre-staged inputs (issue_consts)
  else stage_col_constants<BN, LN>(p, colc, tile_m * BM, tile_n, wave * 64 + lane);
  const bool has_ln = LN && p.ln_stats != nullptr;
  f32x2 rs2[WTM], nm2[WTM];
#pragma unroll
  for (int i = 0; i < WTM; ++i) {
    float mu = 0.f, rs = 1.f;
    if (has_ln) {
      if (pre != nullptr) pre_ln_row<BM>(pre, (wm * WTM + i) * 16 + r16, mu, rs);
      else if (tile_m * BM + (wm * WTM + i) * 16 + r16 < p.M) ln_row(p, tile_m * BM + (wm * WTM + i) * 16 + r16, mu, rs);
    }
    rs2[i] = (f32x2){rs, rs}; nm2[i] = (f32x2){-rs * mu, -rs * mu};
  }
  const int nl_w0 = wn * WTN * 16;                   // this wave's first packed column inside the tile
  // ---- pass 1: gate and pack -------------------------------------------------------------------------------------------------------
  u32x2 pk[WTM][OT];
  auto pass1 = [&](auto ln_c) __attribute__((always_inline)) {
    constexpr bool HAS_LN = decltype(ln_c)::value;
#pragma unroll
    for (int jo = 0; jo < OT; ++jo) {
      const int nl = nl_w0 + (2 * jo) * 16 + g * 4;  // packed value column inside the tile; its gate is 16 further
      const f32x4 bh = *reinterpret_cast<const f32x4*>(colc + nl), bg = *reinterpret_cast<const f32x4*>(colc + nl + 16);
      f32x4 sh = (f32x4){0.f, 0.f, 0.f, 0.f}, sg = sh;
      if (HAS_LN) { sh = *reinterpret_cast<const f32x4*>(colc + BN + nl); sg = *reinterpret_cast<const f32x4*>(colc + BN + nl + 16); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < WTM; ++i) {
        f32x2 h01 = {acc[i][2 * jo][0], acc[i][2 * jo][1]}, h23 = {acc[i][2 * jo][2], acc[i][2 * jo][3]};
        f32x2 g01 = {acc[i][2 * jo + 1][0], acc[i][2 * jo + 1][1]}, g23 = {acc[i][2 * jo + 1][2], acc[i][2 * jo + 1][3]};
        if (HAS_LN) {
          h01 = __builtin_elementwise_fma(h01, rs2[i], __builtin_elementwise_fma(nm2[i], (f32x2){sh[0], sh[1]}, (f32x2){bh[0], bh[1]}));
          h23 = __builtin_elementwise_fma(h23, rs2[i], __builtin_elementwise_fma(nm2[i], (f32x2){sh[2], sh[3]}, (f32x2){bh[2], bh[3]}));
          g01 = __builtin_elementwise_fma(g01, rs2[i], __builtin_elementwise_fma(nm2[i], (f32x2){sg[0], sg[1]}, (f32x2){bg[0], bg[1]}));
          g23 = __builtin_elementwise_fma(g23, rs2[i], __builtin_elementwise_fma(nm2[i], (f32x2){sg[2], sg[3]}, (f32x2){bg[2], bg[3]}));
        } else {
          h01 = h01 + (f32x2){bh[0], bh[1]}; h23 = h23 + (f32x2){bh[2], bh[3]};
          g01 = g01 + (f32x2){bg[0], bg[1]}; g23 = g23 + (f32x2){bg[2], bg[3]};
        }
        const f32x2 lo2 = geglu_pair(h01, g01), hi2 = geglu_pair(h23, g23);
        unsigned lo = Pair16<T>::pack(lo2.x, lo2.y), hi = Pair16<T>::pack(hi2.x, hi2.y);
        asm volatile("" : "+v"(lo), "+v"(hi));       // pin the conversion here (see epilogue_linear_packed)
        pk[i][jo] = (u32x2){lo, hi};
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (has_ln) pass1(std::true_type{}); else pass1(std::false_type{});
  // ---- pass 2: 16 rows x (OT * 16) output columns per step through the wave's staging slice -----------------------------------------------
  const int lrow = lane / CPR, lch = lane - lrow * CPR;
  const bool lact = lrow < RPP;
  const int n_out = p.N >> 1;
  const int n_lane = ((tile_n * BN + nl_w0) >> 1) + lch * 8;      // this lane's 8 output columns
#pragma unroll
  for (int i = 0; i < WTM; ++i) {
#pragma unroll
    for (int jo = 0; jo < OT; ++jo) *reinterpret_cast<u32x2*>(stg + r16 * PITCH + jo * 32 + g * 8) = pk[i][jo];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    u32x4 v4q[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int rowc = (q * RPP + lrow) < 15 ? (q * RPP + lrow) : 15;
      v4q[q] = *reinterpret_cast<const u32x4*>(stg + rowc * PITCH + lch * 16);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int row = q * RPP + lrow;
      const int m = tile_m * BM + (wm * WTM + i) * 16 + row;
      if (lact && row < 16 && m < p.M && n_lane < n_out) *reinterpret_cast<u32x4*>(O + (long long)m * p.ldo + n_lane) = v4q[q];
    }
  }
}

// the residual tile into the accumulators (MFMA layout: a lane holds 4 consecutive channels of a row), before the K loop
template <typename T, int BM, int BN, int WGM, int WGN>
__device__ __forceinline__ void load_residual_acc(const GemmP& p, f32x4 (&acc)[BM / WGM / 16][BN / WGN / 16], int tile_m, int tile_n, int wave, int lane) {
  // ALL loads of the wave are issued before the first one is converted (the accumulators are free: the raw words need no registers
  // of their own), and every load is issued - rows / columns outside the problem read the zero page.  History (profiles/
  // r04_gemm_phase_trace.txt, column `gap`): behind a per-element `if (inside)` hipcc waited for each load before issuing the next
  // (21 000 cycles per 128x320 tile); one batch per row block still meant four dependent round trips (14 600 / 24 800 cycles).
  // (Spreading the chunks over the first K-tile iterations instead - tried - makes the accumulator index a run-time value: hipcc
  // merges the per-chunk cases into one body with a computed address and moves the accumulator array to scratch.)
  constexpr int WTM = BM / WGM / 16, WTN = BN / WGN / 16;
  const int wm = wave / WGN, wn = wave % WGN;
  const int g = lane >> 4, r16 = lane & 15;
  const T* R = reinterpret_cast<const T*>(p.residual);
  const T* zero = reinterpret_cast<const T*>(p.zero);
  // Round 6 (profiles/r06_gemm_phase_trace.txt: the gap between an epilogue and the next K loop is 3-4 k ticks without a residual, 17-27 k with one on the
  // 256x320 tile): the loads above fetch what a lane owns - 8 bytes, and a load instruction 16 rows x 32 contiguous bytes.  Column blocks are now
  // fetched in PAIRS with 16 bytes per lane (lane g of a row: columns 32 jp + 8 g .. + 7, i.e. 16 rows x 64 contiguous bytes per instruction, half as
  // many instructions) and re-sorted in registers to the accumulator layout: (lane bit 5, lane bit 4, dword pair) = (block of the pair, half of the
  // block, quarter of the half) must become (half, quarter, block) - v_permlane16_swap then v_permlane32_swap on the dword pairs (0, 2) and (1, 3).
  // Same values, same conversion: bitwise the results of the 8-byte form (-DFYC_RES_LOAD8 rebuilds it).
#ifdef FYC_RES_LOAD8
  constexpr int NP = 0;
#else
  constexpr int NP = WTN / 2;                          // column-block pairs; an odd last block keeps the 8-byte form
#endif
  u32x4 raw16[WTM][NP > 0 ? NP : 1];
  u32x2 raw[WTM][WTN - 2 * NP > 0 ? WTN - 2 * NP : 1];
#pragma unroll
  for (int i = 0; i < WTM; ++i) {
    const int m = tile_m * BM + (wm * WTM + i) * 16 + r16;
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
      const int n = tile_n * BN + (wn * WTN + 2 * jp) * 16 + g * 8;
      const T* ptr = (m < p.M && n < p.N) ? R + ((long long)m * p.ldr + n) : zero;      // (host: N % 8 == 0, ldr % 8 == 0, 16-byte aligned base)
      raw16[i][jp] = *reinterpret_cast<const u32x4*>(ptr);
    }
#pragma unroll
    for (int j = 2 * NP; j < WTN; ++j) {
      const int n = tile_n * BN + (wn * WTN + j) * 16 + g * 4;
      const T* ptr = (m < p.M && n < p.N) ? R + ((long long)m * p.ldr + n) : zero;
      raw[i][j - 2 * NP] = *reinterpret_cast<const u32x2*>(ptr);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < WTM; ++i) {
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
      unsigned x0 = raw16[i][jp][0], x1 = raw16[i][jp][1], y0 = raw16[i][jp][2], y1 = raw16[i][jp][3];
      auto a = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
      auto b = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
      auto c = __builtin_amdgcn_permlane32_swap(a[0], a[1], false, false);
      auto d = __builtin_amdgcn_permlane32_swap(b[0], b[1], false, false);
      acc[i][2 * jp] = (f32x4){Pair16<T>::lo(c[0]), Pair16<T>::hi(c[0]), Pair16<T>::lo(d[0]), Pair16<T>::hi(d[0])};
      acc[i][2 * jp + 1] = (f32x4){Pair16<T>::lo(c[1]), Pair16<T>::hi(c[1]), Pair16<T>::lo(d[1]), Pair16<T>::hi(d[1])};
    }
#pragma unroll
    for (int j = 2 * NP; j < WTN; ++j)
      acc[i][j] = (f32x4){Pair16<T>::lo(raw[i][j - 2 * NP][0]), Pair16<T>::hi(raw[i][j - 2 * NP][0]), Pair16<T>::lo(raw[i][j - 2 * NP][1]), Pair16<T>::hi(raw[i][j - 2 * NP][1])};
  }
}

// ---- epilogue: lane holds out[m][n0 .. n0+3] per (i, j).
// `stg_stage`: an LDS region of STG_BYTES that no wave reads any more once all waves passed the barrier inside.
// WIDE (bf16 only, host-checked alignment: GemmP::wide) selects the LDS-staged 16-byte-access epilogues; the narrow per-lane
// epilogue is a separate instantiation so that the hot kernels carry neither its code nor its register pressure (with both
// in one kernel the narrow loops no longer unrolled and the whole accumulator array lived in scratch).
template <typename T, int BM, int BN, int WGM, int WGN, int EPI, int STG_BYTES, int MODE, bool WIDE>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, f32x4 (&acc)[BM / WGM / 16][BN / WGN / 16], int tile_m, int tile_n,
                                              long long bz, char* stg_stage, int wave, int lane, const char* pre, int& ecnt) {
  constexpr int WTM = BM / WGM / 16, WTN = BN / WGN / 16;
  constexpr bool LN = (MODE == FYC_GEMM_PLAIN);   // the folded LayerNorm only exists for plain GEMMs: keep it out of the conv kernels
  const int wm = wave / WGN, wn = wave % WGN;
  const int g = lane >> 4, r16 = lane & 15;
  // ---- epilogue: lane holds out[m][n0 .. n0+3] per (i, j) ---------------------------------
  T* O = reinterpret_cast<T*>(p.out) + bz * p.stride_o;
  const T* R = reinterpret_cast<const T*>(p.residual);
  if constexpr (WIDE && EPI == FYC_EPI_HEADS && sizeof(T) == 2) {
    if (p.fast1 && p.heads_pk) {
#if defined(FYC_ABL_EPI) && FYC_ABL_EPI == 3      // timing build: no epilogue (the accumulators stay alive)
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) asm volatile("" :: "v"(acc[i][j][0]), "v"(acc[i][j][1]), "v"(acc[i][j][2]), "v"(acc[i][j][3]));
      return;
#endif
      epilogue_heads_packed<T, BM, BN, WGM, WGN, STG_BYTES, MODE>(p, acc, tile_m, tile_n, stg_stage, wave, lane, pre);
      return;
    }
  }
  if constexpr (WIDE && EPI == FYC_EPI_HEADS) {
    // Wide head-split epilogue.  The plain path stores what a lane holds - 4 consecutive channels (8 B) for q / k and four
    // single bf16 values a whole row pitch apart for the transposed V^T.  Here each wave stages its f32 tile through LDS
    // (as the linear epilogue does) and then writes q / k as 16-byte runs along the head dimension and V^T as 16-byte runs
    // along the TOKEN axis (8 consecutive tokens of one V^T row).  Host guarantees: head_dim % 8 == 0, tokens % 16 == 0,
    // transposed row pitches % 8 == 0, 16-byte aligned segment bases.
    constexpr int JG = (WTN + 1) / 2;
    constexpr int PITCH = JG * 64 + 16;
    static_assert(WGM * WGN * 16 * PITCH + 2 * BN * 4 <= STG_BYTES, "staging + column constants must fit in one ring stage");
#if defined(FYC_ABL_EPI) && FYC_ABL_EPI == 3      // timing build: no epilogue (the accumulators stay alive)
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j) asm volatile("" :: "v"(acc[i][j][0]), "v"(acc[i][j][1]), "v"(acc[i][j][2]), "v"(acc[i][j][3]));
    return;
#endif
    __builtin_amdgcn_s_barrier();
    char* stg = stg_stage + wave * (16 * PITCH);
    float* colc = reinterpret_cast<float*>(stg_stage + WGM * WGN * 16 * PITCH);
    if (pre != nullptr) colc = reinterpret_cast<float*>(const_cast<char*>(pre) + BM * 8);      // pre-staged inputs (issue_consts)
    else stage_col_constants<BN, LN>(p, colc, tile_m * BM, tile_n, wave * 64 + lane);
    const int n_w0 = tile_n * BN + wn * WTN * 16, nl_w0 = wn * WTN * 16;
#pragma unroll
    for (int i = 0; i < WTM; ++i) {
      const int m0 = tile_m * BM + (wm * WTM + i) * 16;            // first of the 16 token rows of this block (same batch element)
      float ln_mu = 0.f, ln_rs = 1.f;
      if (LN && p.ln_stats) {
        if (pre != nullptr) pre_ln_row<BM>(pre, (wm * WTM + i) * 16 + r16, ln_mu, ln_rs);
        else if (m0 + r16 < p.M) ln_row(p, m0 + r16, ln_mu, ln_rs);
      }
      const int b = m0 / p.tokens, tok0 = m0 - b * p.tokens;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j0 = h * JG;
        const int nj = (WTN - j0 < JG) ? (WTN - j0) : JG;
        if (nj <= 0) continue;
#pragma unroll
        for (int jj = 0; jj < JG; ++jj) {
          const int jo = j0 + jj;
          if (jo >= WTN) continue;
          const int nl = nl_w0 + jo * 16 + g * 4;
          f32x4 v = acc[i][jo];
          const f32x4 s4 = *reinterpret_cast<const f32x4*>(colc + BN + nl), b4 = *reinterpret_cast<const f32x4*>(colc + nl);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = ((LN && p.ln_stats ? ln_rs * (v[r] - ln_mu * s4[r]) : v[r]) + b4[r]) * p.out_scale;
          *reinterpret_cast<f32x4*>(stg + r16 * PITCH + (jj * 16 + g * 4) * 4) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int n0 = n_w0 + j0 * 16;                               // first GEMM column of this pass
        if (m0 < p.M) {
          // (1) q / k style segments: 8 consecutive head-dim elements of one token
          const int cpr = nj * 2;
          for (int c = lane; c < 16 * cpr; c += 64) {
            const int row = c / cpr, ch = c - row * cpr;
            const int n = n0 + ch * 8;
            if (n >= p.N) continue;
            const int seg = fdiv_small(n, p.inv_seg_cols), cs = n - seg * p.seg_cols;
            if (p.seg_transposed[seg]) continue;
            const int hh = fdiv_small(cs, p.inv_head_dim), di = cs - hh * p.head_dim;
            float v8[8];
            *reinterpret_cast<f32x4*>(v8) = *reinterpret_cast<const f32x4*>(stg + row * PITCH + ch * 32);
            *reinterpret_cast<f32x4*>(v8 + 4) = *reinterpret_cast<const f32x4*>(stg + row * PITCH + ch * 32 + 16);
            T* S = reinterpret_cast<T*>(p.seg_out[seg]);
            store8<T>(S + ((long long)(b * p.heads + hh) * p.tokens + tok0 + row) * p.head_dim + di, v8);
          }
          // (2) transposed segments (V^T [b*H][d][ld]): 8 consecutive tokens of one head-dim row
          for (int u = lane; u < nj * 32; u += 64) {
            const int col = u >> 1, half = u & 1;
            const int n = n0 + col;
            if (n >= p.N) continue;
            const int seg = fdiv_small(n, p.inv_seg_cols), cs = n - seg * p.seg_cols;
            if (!p.seg_transposed[seg]) continue;
            const int hh = fdiv_small(cs, p.inv_head_dim), di = cs - hh * p.head_dim;
            float v8[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) v8[r] = *reinterpret_cast<const float*>(stg + (half * 8 + r) * PITCH + col * 4);
            T* S = reinterpret_cast<T*>(p.seg_out[seg]);
            store8<T>(S + ((long long)(b * p.heads + hh) * p.head_dim + di) * p.seg_ld[seg] + tok0 + half * 8, v8);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    return;
  }
  if constexpr (WIDE && EPI == FYC_EPI_LINEAR) {
    epilogue_linear_packed<T, BM, BN, WGM, WGN, STG_BYTES, MODE>(p, acc, tile_m, tile_n, bz, stg_stage, wave, lane, pre, ecnt);
    return;
  }
  if constexpr (WIDE && EPI == FYC_EPI_GEGLU && sizeof(T) == 2) {
    if (p.fast1) {
#if defined(FYC_ABL_EPI) && FYC_ABL_EPI == 3      // timing build: no epilogue (the accumulators stay alive)
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) asm volatile("" :: "v"(acc[i][j][0]), "v"(acc[i][j][1]), "v"(acc[i][j][2]), "v"(acc[i][j][3]));
      return;
#endif
      epilogue_geglu_packed<T, BM, BN, WGM, WGN, STG_BYTES, MODE>(p, acc, tile_m, tile_n, bz, stg_stage, wave, lane, pre);
      return;
    }
  }
  if constexpr (WIDE && EPI != FYC_EPI_HEADS && EPI != FYC_EPI_LINEAR) {
    // Wide epilogue (bf16 GEGLU, and LINEAR with an activation: the conditioning encoders): the MFMA layout gives a lane only 4 consecutive channels (8 B), i.e.
    // 32-B row segments per store/residual-load instruction - measured as half the time of the K<=640 layers
    // (profiles/r01_gemm_epilogue_ablation.txt).  Each wave therefore transposes its f32 results through a
    // private slice of the LDS stage that was consumed last and issues 16-B/lane accesses covering >=128-B
    // contiguous row segments.  Single rounding: the residual is added in f32 after staging.
    constexpr bool GLU = (EPI == FYC_EPI_GEGLU);
    constexpr int OT = GLU ? WTN / 2 : WTN;            // 16-column output tiles per wave
    constexpr int JG = (OT + 1) / 2;                   // output tiles per pass
    constexpr int PITCH = JG * 64 + 16;                // bytes per staged row (f32), +16 keeps ds_write_b128 conflict-free
    static_assert(WGM * WGN * 16 * PITCH + 2 * BN * 4 <= STG_BYTES, "staging + column constants must fit in one ring stage");
#if defined(FYC_ABL_EPI) && FYC_ABL_EPI == 3      // timing build: no epilogue (the accumulators stay alive)
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j) asm volatile("" :: "v"(acc[i][j][0]), "v"(acc[i][j][1]), "v"(acc[i][j][2]), "v"(acc[i][j][3]));
    return;
#endif
    __builtin_amdgcn_s_barrier();                      // every wave is done reading the stage we reuse
    char* stg = stg_stage + wave * (16 * PITCH);
    const int n_w0 = tile_n * BN + wn * WTN * 16;      // first GEMM column of this wave
    const int o_w0 = GLU ? (n_w0 >> 1) : n_w0;         // first output column of this wave
    const int n_out = GLU ? (p.N >> 1) : p.N;
    float* colc = reinterpret_cast<float*>(stg_stage + WGM * WGN * 16 * PITCH);   // [2][BN], see stage_col_constants
    // (no output statistics here: fyc_gemm() only takes chan_parts / row_parts with the plain LINEAR epilogue)
    if (pre != nullptr) colc = reinterpret_cast<float*>(const_cast<char*>(pre) + BM * 8);      // pre-staged inputs (issue_consts)
    else stage_col_constants<BN, LN>(p, colc, tile_m * BM, tile_n, wave * 64 + lane);
    float ln_mu[WTM], ln_rs[WTM];
#pragma unroll
    for (int i = 0; i < WTM; ++i) {
      const int m_lane = tile_m * BM + (wm * WTM + i) * 16 + r16;
      ln_mu[i] = 0.f; ln_rs[i] = 1.f;
      if (LN && p.ln_stats) {
        if (pre != nullptr) pre_ln_row<BM>(pre, (wm * WTM + i) * 16 + r16, ln_mu[i], ln_rs[i]);
        else if (m_lane < p.M) ln_row(p, m_lane, ln_mu[i], ln_rs[i]);
      }
    }
    const int nl_w0 = wn * WTN * 16;                   // this wave's first column inside the tile
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int j0 = h * JG;
      const int nj = (OT - j0 < JG) ? (OT - j0) : JG;
      if (nj <= 0) continue;
      const int cpr = nj * 2;                           // 8-element chunks per staged row
      // store pass: a lane keeps ONE 8-column chunk (lch) and walks rows lrow, lrow + rpp, ...: stores still cover whole
      // contiguous row segments, and the per-column statistics accumulate in registers over all rows of the pass
      const int rpp = 64 / cpr;                         // rows per store instruction
      const int lrow = lane / cpr, lch = lane - lrow * cpr;
      const bool lact = lrow < rpp;
      // residual rows of a 16-row block are fetched one block ahead (the kernel has one workgroup per CU: nothing else would
      // hide the HBM round trip, and 8 exposed round trips per tile were most of the time of the K = 320 layers)
      constexpr int NP = 3;                             // store instructions per 16-row block: ceil(16 / rpp) <= 3
      u32x4 rres[2][NP];
      auto fetch_res = [&](int i, u32x4 (&dst)[NP]) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          const int row = q * rpp + lrow;
          const int m = tile_m * BM + (wm * WTM + i) * 16 + row;
          const int n = o_w0 + j0 * 16 + lch * 8;
          dst[q] = (u32x4){0u, 0u, 0u, 0u};
          if (q * rpp < 16 && lact && row < 16 && m < p.M && n < n_out) dst[q] = *reinterpret_cast<const u32x4*>(R + (long long)m * p.ldr + n);
        }
      };
      if (!GLU && R) fetch_res(0, rres[0]);
#pragma unroll
      for (int i = 0; i < WTM; ++i) {
        const int m_lane = tile_m * BM + (wm * WTM + i) * 16 + r16;
        const float* rb = (!GLU && p.rowbias && !p.rb_tile && m_lane < p.M) ? p.rowbias + (long long)(m_lane / p.rows_per_batch) * p.ldrb : nullptr;
        if (!GLU && R && i + 1 < WTM) fetch_res(i + 1, rres[(i + 1) & 1]);
#pragma unroll
        for (int jj = 0; jj < JG; ++jj) {
          const int jo = j0 + jj;
          if (jo >= OT) continue;
          f32x4 v;
          if (GLU) {
            f32x4 hv = acc[i][2 * jo], gv = acc[i][2 * jo + 1];
            const int nl = nl_w0 + (2 * jo) * 16 + g * 4;  // packed value column inside the tile; its gate is 16 further
            if (LN && p.ln_stats) {
              const f32x4 sh = *reinterpret_cast<const f32x4*>(colc + BN + nl), sg = *reinterpret_cast<const f32x4*>(colc + BN + nl + 16);
#pragma unroll
              for (int r = 0; r < 4; ++r) { hv[r] = ln_rs[i] * (hv[r] - ln_mu[i] * sh[r]); gv[r] = ln_rs[i] * (gv[r] - ln_mu[i] * sg[r]); }
            }
            const f32x4 bh = *reinterpret_cast<const f32x4*>(colc + nl), bg = *reinterpret_cast<const f32x4*>(colc + nl + 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) { hv[r] += bh[r]; gv[r] += bg[r]; }
            if constexpr (sizeof(T) == 2) {              // bf16 outputs: polynomial gate (fyc_common.h::geglu_pair)
              const f32x2 lo = geglu_pair((f32x2){hv[0], hv[1]}, (f32x2){gv[0], gv[1]}), hi = geglu_pair((f32x2){hv[2], hv[3]}, (f32x2){gv[2], gv[3]});
              v[0] = lo.x; v[1] = lo.y; v[2] = hi.x; v[3] = hi.y;
            } else {
              v[0] = hv[0] * gelu_erf_f(gv[0]); v[1] = hv[1] * gelu_erf_f(gv[1]);
              v[2] = hv[2] * gelu_erf_f(gv[2]); v[3] = hv[3] * gelu_erf_f(gv[3]);
            }
          } else {
            const int n = n_w0 + jo * 16 + g * 4;
            v = acc[i][jo];
            const int nl = nl_w0 + jo * 16 + g * 4;
            if (LN && p.ln_stats) {
              const f32x4 s4 = *reinterpret_cast<const f32x4*>(colc + BN + nl);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = ln_rs[i] * (v[r] - ln_mu[i] * s4[r]);
            }
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(colc + nl);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += b4[r];
            if (rb && n < p.N) {
              const f32x4 r4 = *reinterpret_cast<const f32x4*>(rb + n);
              v[0] += r4[0]; v[1] += r4[1]; v[2] += r4[2]; v[3] += r4[3];
            }
            if (EPI == EPI_LINEAR_ACT) { v[0] = activate(v[0], p.act); v[1] = activate(v[1], p.act); v[2] = activate(v[2], p.act); v[3] = activate(v[3], p.act); }
          }
          *reinterpret_cast<f32x4*>(stg + r16 * PITCH + (jj * 16 + g * 4) * 4) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          const int r0 = q * rpp;
          if (r0 >= 16) continue;
          const int row = r0 + lrow, ch = lch;
          const int m = tile_m * BM + (wm * WTM + i) * 16 + row;
          const int n = o_w0 + j0 * 16 + ch * 8;
          if (lact && row < 16 && m < p.M && n < n_out) {
            float v[8];
            *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(stg + row * PITCH + ch * 32);
            *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(stg + row * PITCH + ch * 32 + 16);
            if (!GLU) {
              if (R) {
                const u32x4 t = rres[i & 1][q];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[2 * e] += Pair16<T>::lo(t[e]); v[2 * e + 1] += Pair16<T>::hi(t[e]); }
              }
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] *= p.out_scale;
            }
            // statistics of the values AS STORED: pack once (v_cvt_pk_bf16_f32), store, unpack with a shift / mask
            u32x4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = Pair16<T>::pack(v[2 * e], v[2 * e + 1]);
            *reinterpret_cast<u32x4*>(O + (long long)m * p.ldo + n) = pk;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    return;
  }
  if constexpr (!WIDE) {
  float* colc = reinterpret_cast<float*>(stg_stage);
  constexpr bool STATS_OK = (EPI == FYC_EPI_LINEAR || EPI == EPI_LINEAR_ACT) && 2 * BN * 4 + stat_bytes<BM, BN, WGM>() <= STG_BYTES;
  float* cacc = colc + 2 * BN;
  float* racc = cacc + stat_arrays<WGM>() * BN * 2;
  const bool do_cs = STATS_OK && p.chan_parts != nullptr, do_rp = STATS_OK && p.row_parts != nullptr;
  if (p.colc || do_cs || do_rp) {
    __builtin_amdgcn_s_barrier();                      // every wave is done reading the stage we reuse
    if (do_cs || do_rp) stats_zero<BM, BN, WGM, WGM * WGN * 64>(cacc, wave * 64 + lane);
    if (p.colc) stage_col_constants<BN, LN>(p, colc, tile_m * BM, tile_n, wave * 64 + lane);
    else { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
  }
  const int first_sample = do_cs ? (tile_m * BM) / p.cs_rows : 0;
#pragma unroll
  for (int i = 0; i < WTM; ++i) {
    const int m = tile_m * BM + (wm * WTM + i) * 16 + r16;
    // values as stored, for the column statistics: summed over the 16 rows of the block by a fixed xor tree AFTER the row-divergent
    // part (every lane takes part; rows / columns outside the problem stay 0), then added to the array this (wave row, slot) owns
    float xs[STATS_OK ? WTN : 1][4];
    if constexpr (STATS_OK) {
#pragma unroll
      for (int j = 0; j < WTN; ++j) xs[j][0] = xs[j][1] = xs[j][2] = xs[j][3] = 0.f;
    }
    const int st_slot = do_cs ? (tile_m * BM + (wm * WTM + i) * 16) / p.cs_rows - first_sample : 0;
    if (m < p.M) {
    float st_rs = 0.f, st_rq = 0.f;                    // row statistics of this lane's columns
    float ln_mu = 0.f, ln_rs = 1.f;
    if (LN && p.ln_stats) ln_row(p, m, ln_mu, ln_rs);
    if (EPI == FYC_EPI_GEGLU) {
      // packed columns: [32b, 32b+16) = value channels 16b.., [32b+16, 32b+32) = their gates
#pragma unroll
      for (int j = 0; j + 1 < WTN; j += 2) {
        const int n = tile_n * BN + (wn * WTN + j) * 16 + g * 4;  // value column (packed index)
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float h = acc[i][j][r], gt = acc[i][j + 1][r];
          if (LN && p.ln_stats) { h = ln_rs * (h - ln_mu * p.ln_colsum[n + r]); gt = ln_rs * (gt - ln_mu * p.ln_colsum[n + 16 + r]); }
          if (p.bias) { h += p.bias[n + r]; gt += p.bias[n + 16 + r]; }
          v[r] = h * gelu_erf_f(gt);
        }
        const int oc = (n >> 5) * 16 + (n & 15);
        ElemIO<T>::st4(O + (long long)m * p.ldo + oc, v);
      }
    } else {
      const float* rb = (p.rowbias && !(p.colc && p.rb_tile)) ? p.rowbias + (long long)(m / p.rows_per_batch) * p.ldrb : nullptr;
#pragma unroll
      for (int j = 0; j < WTN; ++j) {
        const int n = tile_n * BN + (wn * WTN + j) * 16 + g * 4;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
        const bool full = (n + 3 < p.N);
        if (p.colc) {                                    // staged constants (N % 4 == 0 here): colsum, then bias (+ tile rowbias)
          const int nl = (wn * WTN + j) * 16 + g * 4;
          const f32x4 s4 = *reinterpret_cast<const f32x4*>(colc + BN + nl), b4 = *reinterpret_cast<const f32x4*>(colc + nl);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (LN && p.ln_stats ? ln_rs * (v[r] - ln_mu * s4[r]) : v[r]) + b4[r];
        } else if (LN && p.ln_stats) {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (full || n + r < p.N) v[r] = ln_rs * (v[r] - ln_mu * p.ln_colsum[n + r]);
        }
        if (p.bias && !p.colc) {
          if (full) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + n);
            v[0] += b4[0]; v[1] += b4[1]; v[2] += b4[2]; v[3] += b4[3];
          } else {
            for (int r = 0; r < 4 && n + r < p.N; ++r) v[r] += p.bias[n + r];
          }
        }
        if (rb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (full || n + r < p.N) v[r] += rb[n + r];
        }
        if (EPI == FYC_EPI_LINEAR || EPI == EPI_LINEAR_ACT) {
          if (EPI == EPI_LINEAR_ACT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = activate(v[r], p.act);
          }
          if (R) {
            if (full) {
              float rr[4];
              ElemIO<T>::ld4(R + (long long)m * p.ldr + n, rr);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += rr[r];
            } else {
              for (int r = 0; r < 4 && n + r < p.N; ++r) v[r] += ElemIO<T>::ld(R + (long long)m * p.ldr + n + r);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= p.out_scale;
          if (full) {
            ElemIO<T>::st4(O + (long long)m * p.ldo + n, v);
          } else {
            for (int r = 0; r < 4 && n + r < p.N; ++r) ElemIO<T>::st(O + (long long)m * p.ldo + n + r, v[r]);
          }
          if (do_cs || do_rp) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (full || n + r < p.N) {
                const float x = round_to<T>(v[r]);
                st_rs += x; st_rq = __builtin_fmaf(x, x, st_rq);
                if constexpr (STATS_OK) { if (do_cs) xs[j][r] = x; }
              }
            }
          }
        } else {  // FYC_EPI_HEADS: split columns into segments (q|k|v) and heads
          const int seg = n / p.seg_cols, c = n - seg * p.seg_cols;
          const int h = c / p.head_dim, di = c - h * p.head_dim;
          const int b = m / p.tokens, tok = m - b * p.tokens;
          T* S = reinterpret_cast<T*>(p.seg_out[seg]);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= p.out_scale;
          if (!p.seg_transposed[seg]) {
            ElemIO<T>::st4(S + ((long long)(b * p.heads + h) * p.tokens + tok) * p.head_dim + di, v);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              ElemIO<T>::st(S + ((long long)(b * p.heads + h) * p.head_dim + di + r) * p.seg_ld[seg] + tok, v[r]);
          }
        }
      }
      if (do_rp) { float* dst = racc + ((wm * WTM + i) * 16 + r16) * 2; lds_add(dst, st_rs); lds_add(dst + 1, st_rq); }
    }
    }  // m < p.M
    if constexpr (STATS_OK) {
      if (do_cs) {
        float* own = cacc + (wm + st_slot) * (BN * 2) + ((wn * WTN) * 16 + g * 4) * 2;
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float sx = xs[j][r], sq = xs[j][r] * xs[j][r];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { sx += __shfl_xor(sx, o); sq += __shfl_xor(sq, o); }
            if (r16 == 0) { own[(j * 16 + r) * 2] += sx; own[(j * 16 + r) * 2 + 1] += sq; }
          }
        }
      }
    }
  }
  if (do_cs || do_rp) stats_flush<BM, BN, WGM, WGM * WGN * 64>(p, cacc, tile_m, tile_n, wave * 64 + lane);
  }  // !WIDE
}

template <typename T, int BM, int BN, int WGM, int WGN, int MODE, int EPI, int NS, int RB = 128, bool WIDE = false, int MI = 16>
__global__ void __launch_bounds__(WGM* WGN * 64) fyc_gemm_kernel(const GemmP p) {
  typedef Mma<T> Tr;
  typedef typename Tr::Frag Frag;
  constexpr int NT = WGM * WGN * 64;
  constexpr int CH = 16 / (int)sizeof(T);  // elements per 16-B chunk
  constexpr int CPR = RB / 16;             // 16-B chunks per LDS row (8 or 4)
  constexpr int BK = CPR * CH;             // elements per K tile (one RB-byte LDS row)
  constexpr int KSTEPS = RB / 64;          // MFMA k-steps (4 chunks each) per K tile
  constexpr int A_IT = BM * CPR / NT, B_IT = BN * CPR / NT;
  constexpr int LOADS = A_IT + B_IT;       // DMA instructions per thread per K tile
  constexpr int WTM = BM / WGM / 16, WTN = BN / WGN / 16;
  static_assert(EPI != FYC_EPI_GEGLU || WTN % 2 == 0, "GEGLU pairs value / gate column blocks inside a wave");
  constexpr bool M32 = (MI == 32);       // 32x32x16 matrix instruction in the K loop (16-bit operands, 128-byte K tiles, even block counts)
  static_assert(MI == 16 || (MI == 32 && sizeof(T) == 2 && RB == 128 && WTM % 2 == 0 && WTN % 2 == 0), "32x32x16 main loop: 16-bit operands, wave tile of whole 32x32 blocks");
  constexpr int A_BYTES = BM * RB, STAGE = (BM + BN) * RB;
  constexpr bool STAGGER = (WGM * WGN == 8) && KSTEPS >= 2 && NS == 2;
  static_assert(A_IT * NT == BM * CPR && B_IT * NT == BN * CPR, "tile/threads mismatch");
  static_assert(NS >= 2 && NS <= 4 && (NS - 2) * LOADS < 64, "ring depth");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  // Persistent tile stream: a block walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... and treats their
  // K tiles as ONE continuous DMA stream, so the first K tiles of the next output tile are already in
  // flight while the current tile runs its last MFMAs and its epilogue (hides launch + first-fill latency,
  // which is ~1/4 of the time of the K=320 layers).
  // XCD-aware tile order: block b runs on XCD b%8; give each XCD a contiguous run of tiles
  // (n fastest) so the A panel of a row-block is fetched into one L2 only.  Bijective for any count.
  const int ntiles = p.tiles_m * p.tiles_n;
  auto remap = [&](int t) {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = t & 7, idx = t >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  };
  const long long bz = blockIdx.z;
  const T* __restrict__ A = reinterpret_cast<const T*>(p.a) + bz * p.stride_a;
  const T* __restrict__ W = reinterpret_cast<const T*>(p.w) + bz * p.stride_w;
  const T* __restrict__ A2 = reinterpret_cast<const T*>(p.a2);   // optional second K source (PLAIN): A = [a | a2]
  const T* zero = reinterpret_cast<const T*>(p.zero);

  // ---- per-thread loader state of the tile being ISSUED --------------------------------------------
  // Kept small on purpose (the 256x320 tile runs at the 256-VGPR cap and used to spill its 45 descriptor registers around the
  // epilogue): a thread's LDS row inside a tile is lrow + it*(NT/CPR) and, because NT/CPR is a multiple of 16, its swizzled
  // K offset is the same for every `it` and for both operands.  Addresses are rebuilt from (tile, it) at issue time (a few
  // VALU ops per 1 KiB DMA); only the conv gather keeps per-row state (pixel base and packed top-left tap position).
  const int lrow = tid / CPR;
  const int koff = ((tid % CPR) ^ swz_key<RB, MI>(lrow)) * CH;
  constexpr int ROWS_IT = NT / CPR;
  static_assert(ROWS_IT % 16 == 0, "row stride per DMA instruction must keep the swizzle key");
  int i_tm = 0, i_tn = 0;                 // tile coordinates of the tile being issued (wave-uniform)
  // element offsets of this thread's first row of the tile (row `it` adds it * ROWS_IT * ld, a wave-uniform term)
  // (32-bit element offsets: fyc_gemm() refuses operands of 2^32 elements or more - the largest tensor of the path, the 768x768
  // VAE activations of 32 frames, has 2.4e9 - and a base is one register instead of two in kernels that run at the 256-VGPR cap)
  unsigned a_base = 0, a2_base = 0, b_base = 0;
  // conv: per row, pixel position of the top-left tap (pos0 = frame base + iy0*Win + ix0, may point outside the image) and a
  // 9-bit mask of the taps that fall inside the image (0 for rows beyond M) - built once per tile (KT = 45..360 K tiles follow);
  // per K tile a gather address is pos0 + tap offset (scalar) -> one 64-bit multiply-add
  constexpr bool C3 = (MODE == FYC_GEMM_CONV3X3);
  int a_pos[C3 ? A_IT : 1], a_msk[C3 ? A_IT : 1];
  int a_pix[MODE == FYC_GEMM_CONV3X3_UP2 ? A_IT : 1], a_yx[MODE == FYC_GEMM_CONV3X3_UP2 ? A_IT : 1];   // upsampled conv: frame pixel base (-1: row outside M), (iy0 << 16) | (ix0 & 0xffff)
  int tap = 0, c0 = 0;  // conv: filter tap and channel offset of the K tile being issued
  // split-K: work item w = (output tile w / S, K slice w % S); K slice s covers K tiles [s*KT/S, (s+1)*KT/S)
  const int KT = (p.K + BK - 1) / BK;
  const int S = p.splitk > 1 ? p.splitk : 1;
  auto kt_begin = [&](int work) { return (int)(((long long)(work % S) * KT) / S); };
  auto kt_end = [&](int work) { return (int)(((long long)(work % S + 1) * KT) / S); };
  auto setup_issue = [&](int work) {
    const int t = remap(work / S);
    tile_coords(p, t, i_tm, i_tn);
    const int kt0 = kt_begin(work);
    tap = kt0 % 9; c0 = (kt0 / 9) * BK;      // conv K order (slab, tap, channel): K tile kt = tap kt%9 of slab kt/9 (RB = 128; host keeps split-K off the 64-byte tiles)
    if (MODE == FYC_GEMM_PLAIN || S == 1) { tap = 0; c0 = 0; }
    b_base = (unsigned)(i_tn * BN + lrow) * (unsigned)p.ldw + koff;
    if (MODE == FYC_GEMM_PLAIN) {
      a_base = (unsigned)(i_tm * BM + lrow) * (unsigned)p.lda + koff;
      a2_base = (unsigned)(i_tm * BM + lrow) * (unsigned)p.lda2 + koff - p.k_split;     // (+ k0 >= k_split at use)
    } else {
#pragma unroll
      for (int it = 0; it < A_IT; ++it) {
        const int m = i_tm * BM + lrow + it * ROWS_IT;
        const int hw = p.Hout * p.Wout;
        const int fr = m / hw, rem = m - fr * hw, oy = rem / p.Wout, ox = rem - oy * p.Wout;
        const int iy0 = oy * p.conv_stride - p.conv_pad, ix0 = ox * p.conv_stride - p.conv_pad;
        if (C3) {
          a_pos[it] = fr * p.Hin * p.Win + iy0 * p.Win + ix0;
          int msk = 0;
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) {
            const int iy = iy0 + tp / 3, ix = ix0 + tp % 3;
            if ((unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win) msk |= 1 << tp;
          }
          a_msk[it] = (m < p.M) ? msk : 0;
        } else {
          a_pix[it] = (m < p.M) ? fr * p.Hin * p.Win : -1;
          a_yx[it] = (iy0 << 16) | (ix0 & 0xffff);
        }
      }
    }
  };

  auto src_a = [&](int it, int k0) -> const T* {
    if (MODE == FYC_GEMM_PLAIN) {
      const int m = i_tm * BM + lrow + it * ROWS_IT;
      const int k = k0 + koff;
      if (m >= p.M || k >= p.K) return zero;
      if (A2 != nullptr && k >= p.k_split) return A2 + (size_t)(a2_base + (unsigned)(it * ROWS_IT) * (unsigned)p.lda2 + k0);
      return A + (size_t)(a_base + (unsigned)(it * ROWS_IT) * (unsigned)p.lda + k0);
    } else if (C3) {
      const int ky = tap / 3, kx = tap - 3 * ky;                      // wave-uniform
      const int pos = a_pos[it] + ky * p.Win + kx;
      return ((a_msk[it] >> tap) & 1) ? A + (size_t)((unsigned)pos * (unsigned)p.Cin + (c0 + koff)) : zero;
    } else {  // nearest-upsampled input of virtual size (Hout, Wout): F.interpolate(mode="nearest") folded into the gather
      const int ky = tap / 3, kx = tap - 3 * ky;
      const int iy = (a_yx[it] >> 16) + ky, ix = (int)(short)(a_yx[it] & 0xffff) + kx;
      const bool ok = a_pix[it] >= 0 && (unsigned)iy < (unsigned)p.Hout && (unsigned)ix < (unsigned)p.Wout;
      int sy, sx;
      if (p.up_exact2) { sy = iy >> 1; sx = ix >> 1; }
      else {  // torch: src = min(floor(dst * (in / out)), in - 1), scale in f32
        sy = min((int)floorf((float)iy * p.up_sh), p.Hin - 1);
        sx = min((int)floorf((float)ix * p.up_sw), p.Win - 1);
      }
      return ok ? A + (size_t)((unsigned)(a_pix[it] + sy * p.Win + sx) * (unsigned)p.Cin + c0 + koff) : zero;
    }
  };
  auto src_b = [&](int it, int k0) -> const T* {
    const int n = i_tn * BN + lrow + it * ROWS_IT;
    return (n < p.N && k0 + koff < p.K) ? W + (size_t)(b_base + (unsigned)(it * ROWS_IT) * (unsigned)p.ldw + k0) : zero;
  };

  f32x4 acc[WTM][WTN];
  f32x16 acc32[M32 ? WTM / 2 : 1][M32 ? WTN / 2 : 1];      // (M32: the K loop accumulates here; acc is filled from it behind the loop)
  FYC_STAMP_DECL;
  int fyc_trace_e = 0;      // (timing builds: stamps inside the packed epilogue)

  const int g = lane >> 4, r16 = lane & 15;
  const int sw = swz_key<RB>(r16);   // all fragment rows are r16 + multiples of 16: same key
  // FRAG_ALL: tiles whose accumulators leave room read ALL fragments of a k-step, then issue its MFMAs from registers.  Left to
  // itself hipcc sinks the A reads between the MFMAs (6 reads - wait - 5 MFMA - [1 read - wait - 5 MFMA] x 3 on the 128x320 tile):
  // three exposed LDS round trips per 20 MFMAs, on both waves of a SIMD at once.
  constexpr bool FRAG_ALL = sizeof(T) == 2 && (WTM * WTN + WTM + WTN) * 4 <= 150;
  // FRAG_ROW (round 6): the big tiles (256x320: 4 x 10 blocks per wave, 256x256: 8 x 4) have no room for all fragments of a k-step, and hipcc's
  // own order was: the column fragments + A[0] - wait - one block row of MFMAs - [read A[i] - lgkmcnt(0) - one block row] x (WTM - 1): a FULL LDS round
  // trip exposed in front of every block row but the first, 8 per K tile and wave on the 256x320 tile (profiles/r06_gemm_row_prefetch_ab.txt).  Now the
  // row fragments run PF block rows ahead in a small register ring (PF x WTN x 16 matrix cycles >= ~250), pinned by scheduling fences: the wait in
  // front of block row i is lgkmcnt(PF) and only the first read batch of a k-step is exposed.
#ifdef FYC_NO_FRAG_ROW
  constexpr bool FRAG_ROW = false;
#else
  constexpr bool FRAG_ROW = sizeof(T) == 2 && !FRAG_ALL && !M32;
#endif
  constexpr int PF = (WTN >= 8) ? 2 : (WTN >= 4 ? 4 : 6);      // block rows of read-ahead
  // FYC_LATE_POS (A/B, round 6): where the late half of the waves issues its DMAs - 0 (default): between the two k-steps; r > 0: inside the
  // SECOND k-step, behind block row r - 1 (FRAG_ROW tiles only; the callback `mid` of compute())
#ifndef FYC_LATE_POS
#define FYC_LATE_POS 0
#endif
  constexpr int LATE_ROW = (FRAG_ROW && FYC_LATE_POS > 0 && FYC_LATE_POS < WTM) ? FYC_LATE_POS : 0;
  auto no_mid = []() {};
  auto compute = [&](int stage, int s0, int s1, auto&& mid) {     // MFMA k-steps [s0, s1) of one staged K tile
    const char* sA = smem + stage * STAGE + (wm * WTM * 16 + r16) * RB;
    const char* sB = smem + stage * STAGE + A_BYTES + (wn * WTN * 16 + r16) * RB;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      if (s < s0 || s >= s1) continue;
      const int coff = ((4 * s + g) ^ sw) * 16;
      if constexpr (FRAG_ROW) {
        constexpr int RING = PF + 1;
        Frag bfr[WTN], ar[RING];
#pragma unroll
        for (int j = 0; j < WTN; ++j) bfr[j] = *reinterpret_cast<const Frag*>(sB + j * 16 * RB + coff);
#pragma unroll
        for (int i = 0; i < PF && i < WTM; ++i) ar[i % RING] = *reinterpret_cast<const Frag*>(sA + i * 16 * RB + coff);
#pragma unroll
        for (int i = 0; i < WTM; ++i) {
          if (i + PF < WTM) ar[(i + PF) % RING] = *reinterpret_cast<const Frag*>(sA + (i + PF) * 16 * RB + coff);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < WTN; ++j) acc[i][j] = Tr::mma(bfr[j], ar[i % RING], acc[i][j]);
          __builtin_amdgcn_sched_barrier(0);
          if (LATE_ROW > 0 && s == KSTEPS - 1 && i == LATE_ROW - 1) { mid(); __builtin_amdgcn_sched_barrier(0); }
        }
        continue;
      }
      Frag af[WTM], bf[WTN];
#pragma unroll
      for (int i = 0; i < WTM; ++i) af[i] = *reinterpret_cast<const Frag*>(sA + i * 16 * RB + coff);
#pragma unroll
      for (int j = 0; j < WTN; ++j) bf[j] = *reinterpret_cast<const Frag*>(sB + j * 16 * RB + coff);
      if (FRAG_ALL) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) acc[i][j] = Tr::mma(bf[j], af[i], acc[i][j]);
      if (FRAG_ALL) __builtin_amdgcn_sched_barrier(0);
    }
  };
  // XSTEP (round 6): the small tiles (FRAG_ALL) have room for the fragments of BOTH k-steps of a K tile: the second step's reads are issued in
  // front of the first step's matrix instructions, so one LDS round trip per K tile is exposed instead of two.
  // (A/B, profiles/r06_gemm_row_prefetch_ab.txt: -DFYC_NO_XSTEP / -DFYC_NO_FRAG_ROW rebuild the round-5 orders)
#ifdef FYC_NO_XSTEP
  constexpr bool XSTEP = false;
#else
  constexpr bool XSTEP = FRAG_ALL && KSTEPS == 2 && !M32 && (WTM * WTN + 2 * (WTM + WTN)) * 4 <= 190;
#endif
  auto compute_xstep = [&](int stage, auto&& between) {
    const char* sA = smem + stage * STAGE + (wm * WTM * 16 + r16) * RB;
    const char* sB = smem + stage * STAGE + A_BYTES + (wn * WTN * 16 + r16) * RB;
    const int coff0 = (g ^ sw) * 16, coff1 = ((4 + g) ^ sw) * 16;
    Frag af0[WTM], bf0[WTN], af1[WTM], bf1[WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i) af0[i] = *reinterpret_cast<const Frag*>(sA + i * 16 * RB + coff0);
#pragma unroll
    for (int j = 0; j < WTN; ++j) bf0[j] = *reinterpret_cast<const Frag*>(sB + j * 16 * RB + coff0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < WTM; ++i) af1[i] = *reinterpret_cast<const Frag*>(sA + i * 16 * RB + coff1);
#pragma unroll
    for (int j = 0; j < WTN; ++j) bf1[j] = *reinterpret_cast<const Frag*>(sB + j * 16 * RB + coff1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j) acc[i][j] = Tr::mma(bf0[j], af0[i], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);
    between();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j) acc[i][j] = Tr::mma(bf1[j], af1[i], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);
  };
  // 32x32x16 form: k32-step s = two k16-steps; lane (h = lane / 32, r32 = lane % 32) reads row r32 of a 32-row block, chunk 2 s' + h
  const int r32 = lane & 31, h32 = lane >> 5;
  const int sw32 = swz_key<RB, 32>(r32);
  auto compute32 = [&](int stage, int s0, int s1) {
    constexpr int BI = M32 ? WTM / 2 : 1, BJ = M32 ? WTN / 2 : 1;
    const char* sA = smem + stage * STAGE + (wm * WTM * 16 + r32) * RB;
    const char* sB = smem + stage * STAGE + A_BYTES + (wn * WTN * 16 + r32) * RB;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {            // one k32 step = the fragments of its two k16 steps read together, then 2 x BI x BJ instructions
      if (s < s0 || s >= s1) continue;
      const int coff0 = ((4 * s + h32) ^ sw32) * 16, coff1 = ((4 * s + 2 + h32) ^ sw32) * 16;
      Frag af[2][BI], bf[2][BJ];
#pragma unroll
      for (int i = 0; i < BI; ++i) { af[0][i] = *reinterpret_cast<const Frag*>(sA + i * 32 * RB + coff0); af[1][i] = *reinterpret_cast<const Frag*>(sA + i * 32 * RB + coff1); }
#pragma unroll
      for (int j = 0; j < BJ; ++j) { bf[0][j] = *reinterpret_cast<const Frag*>(sB + j * 32 * RB + coff0); bf[1][j] = *reinterpret_cast<const Frag*>(sB + j * 32 * RB + coff1); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < BI; ++i)
#pragma unroll
          for (int j = 0; j < BJ; ++j) acc32[i][j] = Mma32<T>::mma(bf[u][j], af[u][i], acc32[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto issue = [&](int kt, int stage) {
    char* sA = smem + stage * STAGE;
    char* sB = sA + A_BYTES;
    const int k0 = kt * BK;
#ifdef FYC_ABLATE_CONV_A
    // TIMING BUILD (wrong results): the A gather of a 3x3 convolution is issued for `FYC_ABLATE_CONV_A` of the 9 taps of a slab only -
    // the upper bound of what a halo tile of the input in LDS (each pixel fetched once per slab, read by all nine taps) could save
    if (MODE == FYC_GEMM_PLAIN || tap < FYC_ABLATE_CONV_A)
#endif
#pragma unroll
    for (int it = 0; it < A_IT; ++it) glds16(src_a(it, k0), sA + (it * NT + wave * 64) * 16);
#pragma unroll
    for (int it = 0; it < B_IT; ++it) glds16(src_b(it, k0), sB + (it * NT + wave * 64) * 16);
    if (MODE != FYC_GEMM_PLAIN) {  // K order = (128-B channel slab, ky, kx, channel): the 9 taps of a slab are adjacent
      if (RB == 128) {
        if (++tap == 9) { tap = 0; c0 += BK; }
      } else {                       // 64-B tiles: two tiles per (slab, tap) unit
        if (c0 & BK) { c0 -= BK; if (++tap == 9) { tap = 0; c0 += 2 * BK; } }
        else c0 += BK;
      }
    }
  };

  // Phase shift (round 6).  Every block of a launch runs identical tiles, so all 256 CUs reach their epilogue TOGETHER: ~42 MB of stores
  // leave in one burst at ~2.2 TB/s while the K-loop phases in between leave the HBM write path idle (profiles/r04_gemm_phase_trace.txt),
  // and - loads and stores retire in order on one counter - a wave cannot start the next tile's K loop before its stores are acknowledged.
  // Starting every other block of an XCD half a tile period late puts one half of the chip's epilogues beside the other half's K loops.
  if (p.phase_delay > 0 && ((blockIdx.x >> 3) & 1)) {
    for (int i = 0; i < p.phase_delay; ++i) __builtin_amdgcn_s_sleep(16);
  }
  // ---- epilogue inputs by DMA into the region behind the ring (see pre_bytes) ---------------------------------------------------------
  constexpr bool PRE_BUILT = WIDE && NS == 2;
  char* const pre_lds = smem + NS * STAGE;
  auto issue_consts = [&](int c_tm, int c_tn) {
    constexpr int SP = (BM * 8 + 1023) / 1024, CP = (BN * 4 + 1023) / 1024;       // 1-KiB DMA pieces of the row statistics / of one column array
    const int nrb = p.rowbias != nullptr ? (p.rb_tile ? 1 : p.rb_slots) : 0;
    const int npieces = SP + CP * (2 + nrb);
    const int b0 = (c_tm * BM) / p.rows_per_batch, nb = (p.M + p.rows_per_batch - 1) / p.rows_per_batch;
    for (int q = wave; q < npieces; q += WGM * WGN) {                             // wave-uniform
      const char* src = p.zero;
      int dst, bytes;
      if (q < SP) {                                                                // {mean, rstd} of two rows per lane (host: M even)
        bytes = BM * 8 - q * 1024;
        dst = q * 1024;
        const int row = c_tm * BM + (q * 1024 + lane * 16) / 8;
        if (p.ln_stats != nullptr && row < p.M) src = reinterpret_cast<const char*>(p.ln_stats + 2ll * row);
      } else {
        const int a = (q - SP) / CP, piece = (q - SP) - a * CP;                   // array: 0 bias, 1 column sums, 2 + slot: row-bias rows
        bytes = BN * 4 - piece * 1024;
        dst = BM * 8 + a * (BN * 4) + piece * 1024;
        const int n = c_tn * BN + (piece * 1024 + lane * 16) / 4;
        const float* base = a == 0 ? p.bias : a == 1 ? (p.ln_stats != nullptr ? p.ln_colsum : nullptr)
                                   : (b0 + a - 2 < nb ? p.rowbias + (long long)(b0 + a - 2) * p.ldrb : nullptr);
        if (base != nullptr && n < p.N) src = reinterpret_cast<const char*>(base + n);
      }
      if (lane * 16 < bytes) glds16(src, pre_lds + dst);
    }
  };
  // ---- main loop: NS-deep ring over the continuous K-tile stream, counted waits -------------------
  const int nwork = ntiles * S;
  int i_tile = blockIdx.x, i_kt = 0, i_kt_end = 0;   // issue side of the stream (i_tile: work item being issued)
  int st_c = 0, st_i = 0;                 // stage being computed / issued
  int n_ahead = 0;                        // stream elements issued and not yet consumed
  auto begin_issue = [&]() { setup_issue(i_tile); i_kt = kt_begin(i_tile); i_kt_end = kt_end(i_tile); };
  auto issue_next = [&]() {
    issue(i_kt, st_i);
    st_i = (st_i + 1 == NS) ? 0 : st_i + 1;
    ++n_ahead;
    if (++i_kt == i_kt_end) {
      i_tile += gridDim.x;
      if (i_tile < nwork) begin_issue();
    }
  };
  if (i_tile < nwork) begin_issue();
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (i_tile < nwork) issue_next();

  for (int tile = blockIdx.x; tile < nwork; tile += gridDim.x) {
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (M32) {
#pragma unroll
      for (int i = 0; i < WTM / 2; ++i)
#pragma unroll
        for (int j = 0; j < WTN / 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;
    }
    if constexpr (PRE_BUILT) {
      // the epilogue's inputs of THIS tile, by DMA, in front of its K loop (not inside it: 200 more instructions in the loop body cost the
      // convolutions 2-4 %): behind a barrier - nobody reads the previous tile's inputs any more -, retired by the vmcnt(0) + barrier of the
      // first K-loop iteration
      if (p.pre) {
        __builtin_amdgcn_s_barrier();
        int c_tm, c_tn;
        tile_coords(p, remap(tile / S), c_tm, c_tn);
        issue_consts(c_tm, c_tn);
      }
    }
    if constexpr (WIDE && EPI == FYC_EPI_LINEAR) {
      if (p.res_acc) {                                 // the residual rides in the accumulators (epilogue_linear_packed)
        int rm, rn;
        tile_coords(p, remap(tile / S), rm, rn);
        load_residual_acc<T, BM, BN, WGM, WGN>(p, acc, rm, rn, wave, lane);
        if constexpr (M32) {
#pragma unroll
          for (int i = 0; i < WTM / 2; ++i)
#pragma unroll
            for (int j = 0; j < WTN / 2; ++j) acc16_to_acc32(acc32[i][j], acc[2 * i][2 * j], acc[2 * i][2 * j + 1], acc[2 * i + 1][2 * j], acc[2 * i + 1][2 * j + 1]);
        }
      }
    }
    FYC_STAMP(p, wave, lane);
    const int kt_hi = kt_end(tile), kt_lo = kt_begin(tile);
    for (int kt = kt_lo; kt < kt_hi; ++kt) {
      // the oldest in-flight element must have landed; up to NS-2 younger ones may stay in flight
      const int ahead = min(NS - 2, n_ahead - 1);
      if (NS >= 4 && ahead == 2) wait_vmcnt<2 * LOADS>();
      else if (NS >= 3 && ahead == 1) wait_vmcnt<LOADS>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      --n_ahead;
      // Role stagger (8-wave workgroups, waves w and w+4 share a SIMD): the ~130-instruction address + DMA issue block of a K
      // tile keeps the matrix pipe idle when both waves of a SIMD run it at the same time right behind the barrier.  The
      // second half of the waves therefore runs its first MFMA k-step FIRST and issues in the middle: on every SIMD one wave's
      // issue block sits beside its partner's MFMAs.  (Safe with the 2-deep ring: the stage being refilled was consumed by
      // every wave before this barrier, and the late DMAs still have a k-step of both waves to land.)
      // (one copy of the MFMA code, two of the issue block: the accumulators never cross a divergent join)
      const bool late = STAGGER && p.stagger && wave >= (WGM * WGN) / 2;
      if (!late && i_tile < nwork) issue_next();
      if constexpr (XSTEP) {
        compute_xstep(st_c, [&]() { if (late && i_tile < nwork) issue_next(); });
      } else {
        if constexpr (M32) compute32(st_c, 0, 1); else compute(st_c, 0, 1, no_mid);
        if (LATE_ROW == 0 && late && i_tile < nwork) issue_next();
        if constexpr (M32) compute32(st_c, 1, KSTEPS); else compute(st_c, 1, KSTEPS, [&]() { if (late && i_tile < nwork) issue_next(); });
      }
      st_c = (st_c + 1 == NS) ? 0 : st_c + 1;
    }
    if constexpr (M32) {      // 32x32 blocks -> the 16x16 layout every epilogue (and the split-K partial store) is written for
#pragma unroll
      for (int i = 0; i < WTM / 2; ++i)
#pragma unroll
        for (int j = 0; j < WTN / 2; ++j) acc32_to_acc16(acc32[i][j], acc[2 * i][2 * j], acc[2 * i][2 * j + 1], acc[2 * i + 1][2 * j], acc[2 * i + 1][2 * j + 1]);
    }
    FYC_STAMP(p, wave, lane);
    const int t = remap(tile / S);
    int tile_m, tile_n;
    tile_coords(p, t, tile_m, tile_n);
    if (S > 1) {      // split-K: raw f32 partial sums; fyc_gemm's finish kernel adds the slices and applies the epilogue
      float* Wp = p.ws + (long long)(tile % S) * p.M * p.N;
#pragma unroll
      for (int i = 0; i < WTM; ++i) {
        const int m = tile_m * BM + (wm * WTM + i) * 16 + r16;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
          const int n = tile_n * BN + (wn * WTN + j) * 16 + g * 4;
          if (n < p.N) *reinterpret_cast<f32x4*>(Wp + (long long)m * p.N + n) = acc[i][j];
        }
      }
      continue;
    }

    gemm_epilogue<T, BM, BN, WGM, WGN, EPI, STAGE, MODE, WIDE>(p, acc, tile_m, tile_n, bz, smem + ((st_c == 0) ? NS - 1 : st_c - 1) * STAGE, wave, lane,
                                                               (PRE_BUILT && p.pre) ? pre_lds : nullptr, fyc_trace_e);
    FYC_STAMP(p, wave, lane);
  }  // tile stream
}

// rowbias rows (batch elements) a row tile of bm rows can touch; 0 = more than the packed LINEAR epilogue stages (fyc_gemm() then takes the narrow epilogue)
inline int rowbias_slots(int bm, int rpb) {
  if (rpb <= 0 || rpb % 16 != 0) return 0;
  const int n = (bm % rpb == 0) ? bm / rpb : (bm - 1) / rpb + 2;
  return n <= RB_SLOTS ? n : 0;
}

template <typename T, int BM, int BN, int WGM, int WGN, int MODE, int EPI, int NS, int RB = 128, bool WIDE = false, int MI = 16>
int launch(const GemmP& p, int batch, hipStream_t st) {
  constexpr bool PRE_BUILT = WIDE && NS == 2;
  constexpr int smem = NS * (BM + BN) * RB + (PRE_BUILT ? pre_bytes<BM, BN>() : 0);
  static_assert(smem <= 160 * 1024, "LDS budget");
  auto kern = fyc_gemm_kernel<T, BM, BN, WGM, WGN, MODE, EPI, NS, RB, WIDE, MI>;
  int dev = 0;
  (void)hipGetDevice(&dev);
  // per-device one-time setup (attribute + CU count), guarded: one process may drive several GPUs from several threads
  static std::mutex mu;
  static bool attr_done[FYC_MAX_DEVICES] = {};
  static int n_cu_dev[FYC_MAX_DEVICES] = {};
  int n_cu = 256;
  if (dev >= 0 && dev < FYC_MAX_DEVICES) {
    std::lock_guard<std::mutex> lk(mu);
    if (!attr_done[dev]) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      hipDeviceProp_t pr;
      n_cu_dev[dev] = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
      attr_done[dev] = true;
    }
    n_cu = n_cu_dev[dev];
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  }
  GemmP q = p;
  q.tiles_m = (p.M + BM - 1) / BM;
  q.tiles_n = (p.N + BN - 1) / BN;
  q.rb_tile = (p.colc && p.rowbias != nullptr && p.rows_per_batch % BM == 0) ? 1 : 0;
  q.rb_slots = (WIDE && EPI == FYC_EPI_LINEAR && p.rowbias != nullptr && !q.rb_tile) ? rowbias_slots(BM, p.rows_per_batch) : 0;
  // epilogue inputs by DMA under the K loop: 2-deep-ring wide epilogues with at least two K tiles per output tile; the row statistics as
  // stored {mean, rstd} pairs, two rows per 16-byte lane (M even, 16-byte aligned); row-bias rows only where the packed LINEAR epilogue
  // stages them (fyc_set_tuning key 12 = 1 restores the loads in the epilogue: A/B)
  {
    constexpr int BK = (RB / 16) * (16 / (int)sizeof(T));
    const int kt = (p.K + BK - 1) / BK;
    const bool rb_ok = p.rowbias == nullptr || (EPI == FYC_EPI_LINEAR && (q.rb_tile || q.rb_slots > 0));
    q.pre = (PRE_BUILT && g_fyc_tuning[12] != 1 && kt >= 2 && !(q.splitk > 1) && p.ln_nparts == 0 && rb_ok && p.wide &&
             (p.ln_stats == nullptr || (p.M % 2 == 0 && ((uintptr_t)p.ln_stats % 16) == 0))) ? 1 : 0;
  }
  q.fast1 = g_fyc_tuning[13] == 1 ? 0 : 1;
  // pack-first head-split epilogue: in the pipeline it wins at the 16x16 level only (8192x3840x1280 0.588 -> 0.562 ms per DDIM step, 8192x1280x1280
  // 0.235 -> 0.212) and LOSES at the 64x64 / 32x32 levels (131072x960x320 0.970 -> 1.051, 32768x1920x640 0.666 -> 0.756: profiles/r06_gemm_epilogue_ab.txt
  // part 4) - key 15 = 1 forces it on, 2 off
  q.heads_pk = g_fyc_tuning[15] == 1 ? 1 : g_fyc_tuning[15] == 2 ? 0 : (p.M <= 8192 ? 1 : 0);
  q.stagger = g_fyc_tuning[5] == 1 ? 0 : 1;
  q.phase_delay = g_fyc_tuning[11] > 0 ? g_fyc_tuning[11] : 0;
  q.res_acc = (WIDE && EPI == FYC_EPI_LINEAR && p.residual != nullptr && p.ln_stats == nullptr && !(q.splitk > 1)) ? 1 : 0;
#ifdef FYC_TRACE
  q.trace = g_fyc_trace;
#endif
  q.strip = (q.tiles_n > 4 && g_fyc_tuning[4] >= 0) ? (g_fyc_tuning[4] > 0 ? g_fyc_tuning[4] : (q.tiles_n >= 16 ? 8 : 4)) : 0;   // measured: profiles/r01_gemm_strip_order.txt
  // persistent grid: as many blocks as stay resident (LDS-limited), each walks a strided tile list
  int occ = (160 * 1024) / smem;
  const int wave_cap = 32 / (WGM * WGN);
  if (occ > wave_cap) occ = wave_cap;
  if (occ < 1) occ = 1;
  long long resident = (long long)n_cu * occ / (batch > 0 ? batch : 1);
  if (resident < n_cu) resident = n_cu;
  const long long ntiles = (long long)q.tiles_m * q.tiles_n * (q.splitk > 1 ? q.splitk : 1);
  dim3 grid((unsigned)(ntiles < resident ? ntiles : resident), 1, batch);
  hipLaunchKernelGGL(kern, grid, dim3(WGM * WGN * 64), smem, st, q);
  FYC_CHECK_LAUNCH("fyc_gemm");
  return 0;
}

// tile configurations: id -> (BM, BN, WGM, WGN)
//   1: 128x128, 4 waves   2: 128x64, 4 waves   3: 256x128, 8 waves   4: 256x64, 4 waves
//   5: 256x320, 8 waves   6: 128x320, 8 waves  7: 256x256, 8 waves   8 / 10: 128x320 / 128x128 with 64-byte K tiles
// All use the 2-deep ring (deeper rings measured no gain, profiles/r01_gemm_tile_sweep*.txt); config 1 is also built 3-deep so
// that the counted-wait ring logic stays exercised (tests).  The narrow epilogue (WIDE = false: f32 parity mode and bf16
// problems whose shapes / alignment rule out 16-byte accesses) only exists for configs 1 and 2.
template <typename T, int MODE, int EPI, bool WIDE>
int dispatch_cfg(int cfg, int ns, const GemmP& p, int batch, hipStream_t st) {
  if constexpr (!WIDE) {
    if (cfg != 1 && cfg != 2) cfg = (p.N % 128 == 0 || p.N > 512) ? 1 : 2;
    if (cfg == 1) return launch<T, 128, 128, 2, 2, MODE, EPI, 2, 128, false>(p, batch, st);
    return launch<T, 128, 64, 2, 2, MODE, EPI, 2, 128, false>(p, batch, st);
  } else {
    switch (cfg) {
      case 1: if (ns == 3) return launch<T, 128, 128, 2, 2, MODE, EPI, 3, 128, true>(p, batch, st);
              return launch<T, 128, 128, 2, 2, MODE, EPI, 2, 128, true>(p, batch, st);
      case 2: if constexpr (MODE == FYC_GEMM_PLAIN) {   // deeper rings for the latency-bound small-M linears (8x8 latent level): see gemm.hip::choose
                if (ns == 4) return launch<T, 128, 64, 2, 2, MODE, EPI, 4, 128, true>(p, batch, st);
                if (ns == 3) return launch<T, 128, 64, 2, 2, MODE, EPI, 3, 128, true>(p, batch, st);
              }
              return launch<T, 128, 64, 2, 2, MODE, EPI, 2, 128, true>(p, batch, st);
      case 3: return launch<T, 256, 128, 4, 2, MODE, EPI, 2, 128, true>(p, batch, st);
      case 4: return launch<T, 256, 64, 4, 1, MODE, EPI, 2, 128, true>(p, batch, st);
      // N = 320*k (every layer width of SD-1.5): 320-wide tiles read the A panel once per 320 columns
      case 5: return launch<T, 256, 320, 4, 2, MODE, EPI, 2, 128, true>(p, batch, st);
      case 6: if constexpr (EPI == FYC_EPI_GEGLU) FYC_FAIL(-2, "fyc_gemm: tile config 6 gives a wave an odd number of column blocks: not built for GEGLU");
              else return launch<T, 128, 320, 2, 4, MODE, EPI, 2, 128, true>(p, batch, st);
      case 7: return launch<T, 256, 256, 2, 4, MODE, EPI, 2, 128, true>(p, batch, st);
      // 64-byte K tiles: half the LDS per stage -> two independent 4-wave blocks per CU with 64x160 wave tiles (8)
      case 8: return launch<T, 128, 320, 2, 2, MODE, EPI, 2, 64, true>(p, batch, st);
      case 10: return launch<T, 128, 128, 2, 2, MODE, EPI, 2, 64, true>(p, batch, st);
      // round 6: 128x160 over 2x2 waves (64x80 per wave, the wave tile of config 6) with 72 KB of ring: TWO independent workgroups per CU,
      // so one's epilogue / first fill runs under the other's K loop - for the short-K problems whose 128x320 tiles spend as long in the
      // epilogue as in the K loop (profiles/r06_gemm_two_workgroups_ab.txt)
      // round 6: the 32x32x16 matrix instruction in the K loop (same tiles and wave grids as 5 / 7 / 3).  Parity-green, 8-12 % SLOWER than the
      // 16x16x32 loop on every shape (profiles/r06_gemm_mi32_ab.txt): only in builds with FYC_BUILD_EXTRA=-DFYC_GEMM_MI32 (tests: FYC_TEST_MI32=1)
#ifdef FYC_GEMM_MI32
      case 12: return launch<T, 256, 320, 4, 2, MODE, EPI, 2, 128, true, 32>(p, batch, st);
      case 13: return launch<T, 256, 256, 2, 4, MODE, EPI, 2, 128, true, 32>(p, batch, st);
      case 14: return launch<T, 256, 128, 4, 2, MODE, EPI, 2, 128, true, 32>(p, batch, st);
#endif
      case 11: if constexpr (EPI == FYC_EPI_GEGLU) FYC_FAIL(-2, "fyc_gemm: tile config 11 gives a wave an odd number of column blocks: not built for GEGLU");
               else return launch<T, 128, 160, 2, 2, MODE, EPI, 2, 128, true>(p, batch, st);
    }
    FYC_FAIL(-2, "fyc_gemm: tile config %d not built", cfg);
  }
}

template <typename T, int MODE, int EPI>
int dispatch_ns(int ns, int cfg, const GemmP& p, int batch, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    if (p.wide) return dispatch_cfg<T, MODE, EPI, true>(cfg, ns, p, batch, st);
  }
  return dispatch_cfg<T, MODE, EPI, false>(cfg, ns, p, batch, st);
}

// one translation unit per (dtype, family) keeps the build parallel
int run_bf16_plain(const GemmP& p, int batch, int cfg, int ns, hipStream_t st);
int run_bf16_conv(const GemmP& p, int batch, int cfg, int ns, hipStream_t st);
int run_bf16_act(const GemmP& p, int batch, int cfg, hipStream_t st);   // LINEAR + activation: tile configs 1, 2, 6
int run_f32(const GemmP& p, int batch, int cfg, hipStream_t st);
// f16 storage (FYC_F16): the same kernels on v_mfma_f32_16x16x32_f16
int run_f16_plain(const GemmP& p, int batch, int cfg, int ns, hipStream_t st);
int run_f16_conv(const GemmP& p, int batch, int cfg, int ns, hipStream_t st);
int run_f16_act(const GemmP& p, int batch, int cfg, hipStream_t st);

}  // namespace fycg
