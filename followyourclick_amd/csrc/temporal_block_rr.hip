// fyc_temporal_block, register-resident form (the path taken when the caller passes the packed `wstream`): one temporal
// self-attention sub-block of the motion module (reference motion_module.py:270-283, 371-464) with NO activation traffic through
// LDS at all.  Same design as ff_block.hip:
//
//   * a workgroup = 4 wave64 (one per SIMD, the whole 512-register file each) owns 8 pixels x 16 frames = 128 token rows; a
//     wave owns TWO PIXELS, row = frame: a 16-row MFMA block is one pixel's frame axis, so the whole attention of a pixel
//     (q k^T over 16 frames, softmax, P v) happens inside one wave's registers;
//   * the wave's 32 x C tokens are normalised in registers ((x - mean) rstd, bf16; gamma / beta are folded into the weights /
//     bias tables) and stay there as the MFMA operand of all 24 projections (8 heads x q, k, v);
//   * q^T, k^T (features x tokens) come out of W x^T products, v (tokens x features) out of x W^T with the SAME operand registers
//     (A and B fragments of v_mfma_f32_16x16x32_bf16 have the same lane layout), and each result already is the operand of its
//     consumer: q^T, k^T -> B / A of S^T = k q^T;  S^T (keys x queries) -> after the softmax, B of O^T = v^T P^T;  v -> A of the
//     same product;  O^T (d x tokens) -> B of the output projection.  A lane holds 4 consecutive rows of a 16-row result where
//     an operand wants 8 consecutive k: the k-slot <-> feature map that follows from this is baked into the packed Wo' (and is
//     the same for q and k, whose product only needs it to be consistent): slot 8 g + e of k-step 0 = feature 4 g + e of block 0
//     (e < 4) / of block 1 (e >= 4); of k-step 1 = block 2 (e < 4) / zero.  No shuffle, no LDS round trip;
//   * the weights arrive as one pre-packed stream (engine/weights.py::pack_temporal_block): per head two stages of 1-KiB MFMA
//     fragments by asm-issued global_load_lds into a 2-deep LDS ring, dealt out between the MFMAs of the previous stage:
//       stage A: Wq', Wk' fragments (60) + the f32 bias table [frame][q 48 | k 48] (bias + positional encoding, 6 pieces)
//       stage B: Wv' fragments (30) + Wo' fragments of the head (40) + the f32 v bias table [48][frame] (3 pieces)
//   * epilogue: residual tile by DMA into the idle ring (row pitch padded against bank conflicts), + bias + accumulators, one
//     rounding, copy-out in 640-B rows.
//
// Built for C = 320, 8 heads of 40, 16 frames, bf16 (the 64x64 level of the 512^2 workload: 10 launches per UNet call).
// Compiled WITHOUT -amdgpu-mfma-vgpr-form (see _build.py): the accumulators must live in AGPRs for the 512-register budget.
#include <mutex>
#include <type_traits>

#include "fyc_common.h"

namespace {

constexpr int C_ = 320, H_ = 8, D_ = 40, F_ = 16, PIX = 8, ROWS = PIX * F_, NT = 256;
constexpr int KS = C_ / 32;                    // 10 MFMA k-steps over C
constexpr int NB = C_ / 16;                    // 20 column blocks of the output
constexpr int PIECE = 1024;
constexpr int A_TAB = 60, A_PIECES = 66;       // stage A: pieces s * 6 + b (b < 3 q blocks, b >= 3 k blocks), table at 60
constexpr int B_WO = 30, B_TAB = 70, B_PIECES = 73;   // stage B: pieces s * 3 + b (v blocks), 30 + t * 20 + j (Wo'), table at 70
constexpr int STAGE_BYTES = B_PIECES * PIECE;  // 74752 (stage A is padded to the same stride in the stream)
constexpr int NSTAGE = 2 * H_;
constexpr int LDS_BYTES = 2 * STAGE_BYTES;     // 149504
// residual / output tile of the epilogue (overlays the ring from 0), row = pixel * 16 + frame.  Row pitch 672 B: consecutive rows are
// 168 dwords = 40 (mod 64 banks) apart, so the 16 rows x 4 quads x 8 B of one in-place add spread over all banks (2 passes); at
// the natural 640 B - and with the frames of a pixel 8 rows apart - all 16 rows of an access met in one bank and the epilogue
// took 23 k of the tile's 111 k cycles (profiles/r03_temporal_block_rr_phases.txt)
constexpr int TP = C_ * 2 + 32;
constexpr int TILE_BYTES = ROWS * TP;          // 86016 = 84 pieces
static_assert(TILE_BYTES % PIECE == 0 && TILE_BYTES <= STAGE_BYTES + B_WO * PIECE, "the residual tile may only cover slot 0 and the v fragments of slot 1");
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

struct TRP {
  const bf16_t* x; bf16_t* out;
  const char* ws;
  const float* b_out;
  int pixels;
  float scale_log2e, eps;
};

// 1 KiB global -> LDS by DMA from inline asm (see ff_block.hip::dma16 for why not the builtin)
__device__ __forceinline__ void dma16(const char* gbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma16v(const void* gsrc, unsigned lds_dst) {                        // per-lane source address
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma_landed_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}
#ifdef TB_NO_MFMA                                              // ablation build: DMA + LDS reads + barriers only, wrong results
template <typename Frag> __device__ __forceinline__ f32x4 mfma(Frag a, Frag b, f32x4 c) { c[0] += __builtin_bit_cast(f32x4, a)[0] + __builtin_bit_cast(f32x4, b)[0]; return c; }
#else
__device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
#endif
template <typename Frag> __device__ __forceinline__ Frag frag(const char* sl, int piece) { return *reinterpret_cast<const Frag*>(sl + piece * PIECE); }

// reductions over the four 16-lane rows of a wave by v_permlane16_swap / v_permlane32_swap: plain VALU.  No LDS-queue instruction
// (ds_bpermute = __shfl_xor) may sit between asm-issued DMAs: profiles/r03_ff_block_race.txt
__device__ __forceinline__ float rows_sum(float v) { return swap32_sum(swap16_sum(v)); }   // (fyc_common.h: opaque swap results, as for the maxima)
__device__ __forceinline__ float rows_max(float v) { return swap32_max(swap16_max(v)); }   // (fyc_common.h: the plain fmaxf form is miscompiled)
template <typename T> __device__ __forceinline__ typename Pair16<T>::Vec8 op8(const f32x4& a, const f32x4& b) {          // two 4-row results -> one 8-slot operand
  return __builtin_bit_cast(typename Pair16<T>::Vec8, (u32x4){Pair16<T>::pack(a[0], a[1]), Pair16<T>::pack(a[2], a[3]), Pair16<T>::pack(b[0], b[1]), Pair16<T>::pack(b[2], b[3])});
}
template <typename T> __device__ __forceinline__ typename Pair16<T>::Vec8 op4(const f32x4& a) {                          // one 4-row result, upper slots zero
  return __builtin_bit_cast(typename Pair16<T>::Vec8, (u32x4){Pair16<T>::pack(a[0], a[1]), Pair16<T>::pack(a[2], a[3]), 0u, 0u});
}

#ifdef TB_TIMING                                                // phase timestamps of wave 0 of every workgroup (tools/tblock_probe.py --timing)
__device__ unsigned long long g_tb_time[1024 * 24];
#define TB_MARK(k) do { if (tid == 0) g_tb_time[blockIdx.x * 24 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define TB_MARK(k) do {} while (0)
#endif

// T: the 16-bit element type of x / out / the weight stream (bf16_t or f16_t; TRP's pointers are typed bf16_t for both)
template <typename T>
__global__ void __launch_bounds__(NT) tblock_rr_kernel(const TRP p) {
  typedef typename Pair16<T>::Vec8 Frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int g = lane >> 4, r16 = lane & 15;
  const unsigned lane16 = (unsigned)lane * 16u;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int tiles_per_clip = p.pixels / PIX;
  const int clip = blockIdx.x / tiles_per_clip, p0 = (blockIdx.x - clip * tiles_per_clip) * PIX;
  const long long fstride = (long long)p.pixels * C_;       // elements between consecutive frames of a pixel
  const bf16_t* xb = p.x + ((long long)clip * F_ * p.pixels + p0) * C_;
  bf16_t* ob = p.out + ((long long)clip * F_ * p.pixels + p0) * C_;

  // ---- the wave's 2 pixels x 16 frames as MFMA operands: lane (frame r16, quad g) holds x[pixel 2 wave + i][frame][32 s + 8 g .. +8]
  Frag xa[2][KS];
  {
    const bf16_t* xr = xb + r16 * fstride + (2 * wave) * C_ + g * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s = 0; s < KS; ++s) xa[i][s] = *reinterpret_cast<const Frag*>(xr + i * C_ + s * 32);
  }
  TB_MARK(0);
  {                                                           // stage 0 (head 0, A) into slot 0
    const char* src = p.ws;
#pragma unroll 1
    for (int q = wave; q < A_PIECES; q += 4) dma16(src + q * PIECE, lane16, lds0 + q * PIECE);
  }

  // LayerNorm in registers: statistics two-pass, then the tokens are replaced by (x - mean) rstd rounded to bf16
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const u32x4 t = __builtin_bit_cast(u32x4, xa[i][k]);
#pragma unroll
      for (int e = 0; e < 4; ++e) s += Pair16<T>::lo(t[e]) + Pair16<T>::hi(t[e]);
    }
    const float m = rows_sum(s) * (1.0f / C_);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const u32x4 t = __builtin_bit_cast(u32x4, xa[i][k]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = Pair16<T>::lo(t[e]) - m, b = Pair16<T>::hi(t[e]) - m;
        q = __builtin_fmaf(a, a, q);
        q = __builtin_fmaf(b, b, q);
      }
    }
    const float rs = rsqrtf(rows_sum(q) * (1.0f / C_) + p.eps);
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      u32x4 t = __builtin_bit_cast(u32x4, xa[i][k]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        t[e] = Pair16<T>::pack((Pair16<T>::lo(t[e]) - m) * rs, (Pair16<T>::hi(t[e]) - m) * rs);
      xa[i][k] = __builtin_bit_cast(Frag, t);
    }
  }

  f32x4 oacc[2][NB];                                          // out^T: rows (features) 16 j + 4 g .. +4, column (token) r16 of pixel i
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) oacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // n-th piece of this wave of stage tnext (np pieces; the last round wraps around and fetches the first pieces again: no branch)
  auto dma_piece = [&](int tnext, int n, int np) {
#ifdef TB_NO_DMA
    return;                                                   // ablation build (tools/tblock_probe.py): compute only, wrong results
#endif
    int q = wave + 4 * n;
    q = q >= np ? q - np : q;
    dma16(p.ws + (long long)tnext * STAGE_BYTES + q * PIECE, lane16, lds0 + (tnext & 1) * STAGE_BYTES + q * PIECE);
  };

  Frag q_op[2][2], k_op[2][2];

  // ---- stage A of a head (slot 0): q^T, k^T = W' xn^T (+ bias + positional encoding), packed as operands of S^T ------------
  auto stage_a = [&](int t) {
    const char* base = smem;
    const char* sl = base + lane16;
    f32x4 qk[2][6];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int b = 0; b < 6; ++b) qk[i][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Frag w[6], w1[6];                                       // fragments two k-steps ahead of their MFMAs: one k-step (192 cycles of
#pragma unroll                                                 // MFMA) does not cover an LDS round trip while DMA pieces are landing
    for (int b = 0; b < 6; ++b) { w[b] = frag<Frag>(sl, b); w1[b] = frag<Frag>(sl, 6 + b); }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      Frag n[6];
#pragma unroll
      for (int b = 0; b < 6; ++b) n[b] = (s + 2 < KS) ? frag<Frag>(sl, (s + 2) * 6 + b) : w1[b];
#pragma unroll
      for (int b = 0; b < 6; ++b)
#pragma unroll
        for (int i = 0; i < 2; ++i) qk[i][b] = mfma(w[b], xa[i][s], qk[i][b]);
      dma_piece(t + 1, 2 * s, B_PIECES);                      // 19 pieces per wave of stage B (73): two per k-step ...
      if (s < KS - 1) dma_piece(t + 1, 2 * s + 1, B_PIECES);  // ... 19 = 2 * 9 + 1
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < 6; ++b) { w[b] = w1[b]; w1[b] = n[b]; }
    }
    const float* tab = reinterpret_cast<const float*>(base + A_TAB * PIECE) + r16 * 96 + g * 4;     // [frame][q 48 | k 48]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int b = 0; b < 6; ++b) qk[i][b] += *reinterpret_cast<const f32x4*>(tab + b * 16);
      q_op[i][0] = op8<T>(qk[i][0], qk[i][1]);
      q_op[i][1] = op4<T>(qk[i][2]);
      k_op[i][0] = op8<T>(qk[i][3], qk[i][4]);
      k_op[i][1] = op4<T>(qk[i][5]);
    }
  };

  // ---- stage B of a head (slot 1): S^T = k q^T and its softmax beside v = xn Wv'^T; O^T = v^T P^T; out^T += Wo' O^T ----------
  auto stage_b = [&](int t, auto last_head) {
    constexpr bool LAST = decltype(last_head)::value;
    const char* base = smem + STAGE_BYTES;
    const char* sl = base + lane16;
    f32x4 st[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      st[i] = mfma(k_op[i][0], q_op[i][0], (f32x4){0.f, 0.f, 0.f, 0.f});
      st[i] = mfma(k_op[i][1], q_op[i][1], st[i]);
    }
    f32x4 v[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int b = 0; b < 3; ++b) v[i][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Frag w[3], w1[3], w2[3];                                // three k-steps ahead: a k-step is only 6 MFMAs here
#pragma unroll
    for (int b = 0; b < 3; ++b) { w[b] = frag<Frag>(sl, b); w1[b] = frag<Frag>(sl, 3 + b); w2[b] = frag<Frag>(sl, 6 + b); }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      Frag n[3];
#pragma unroll
      for (int b = 0; b < 3; ++b) n[b] = (s + 3 < KS) ? frag<Frag>(sl, (s + 3) * 3 + b) : w2[b];
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int i = 0; i < 2; ++i) v[i][b] = mfma(xa[i][s], w[b], v[i][b]);
      if constexpr (!LAST) {                                   // 17 pieces per wave of the next head's stage A (66): 2 * 8 + 1
        if (s < 9) dma_piece(t + 1, s < 8 ? 2 * s : 16, A_PIECES);
        if (s < 8) dma_piece(t + 1, 2 * s + 1, A_PIECES);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < 3; ++b) { w[b] = w1[b]; w1[b] = w2[b]; w2[b] = n[b]; }
    }
    if constexpr (LAST) {                                     // every wave is done with the v fragments: the residual tile may land over them
      __syncthreads();
#pragma unroll 1
      for (int q = wave; q < TILE_BYTES / PIECE; q += 4) {      // the LDS image is linear; every lane fetches the 16 B that belong at its place
        const int off = q * PIECE + (int)lane16, row = off / TP;
        const int col = min(off - row * TP, C_ * 2 - 16);         // (the 32 pad bytes of a row re-fetch its last chunk)
        dma16v(reinterpret_cast<const char*>(xb + (row & 15) * fstride + (row >> 4) * C_) + col, lds0 + q * PIECE);
      }
    }
    // softmax over the 16 keys of a query: a lane holds keys 4 g .. 4 g + 3 of query r16
    Frag p_op[2];
    float inv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x4 sv = st[i] * p.scale_log2e;
      const float m = rows_max(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])));
#pragma unroll
      for (int e = 0; e < 4; ++e) sv[e] = __builtin_amdgcn_exp2f(sv[e] - m);
      inv[i] = __builtin_amdgcn_rcpf(rows_sum((sv[0] + sv[1]) + (sv[2] + sv[3])));
      p_op[i] = op4<T>(sv);
    }
    Frag wo[4], wo1[4];                                     // the first Wo' fragments are requested before the attention chain
#pragma unroll
    for (int q = 0; q < 4; ++q) { wo[q] = frag<Frag>(sl, B_WO + q); wo1[q] = frag<Frag>(sl, B_WO + 4 + q); }
    const float* tab = reinterpret_cast<const float*>(base + B_TAB * PIECE) + r16 * 16 + g * 4;      // [feature 48][frame]
    Frag o_op[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x4 ot[3];
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        v[i][b] += *reinterpret_cast<const f32x4*>(tab + b * 256);
        ot[b] = mfma(op4<T>(v[i][b]), p_op[i], (f32x4){0.f, 0.f, 0.f, 0.f});
      }
#pragma unroll
      for (int b = 0; b < 3; ++b) ot[b] *= inv[i];
      o_op[i][0] = op8<T>(ot[0], ot[1]);
      o_op[i][1] = op4<T>(ot[2]);
    }
    // output projection of the head: 20 column blocks x 2 k-steps, four fragments in flight
#pragma unroll
    for (int u = 0; u < 2 * NB / 4; ++u) {                     // u = t2 * 5 + jb: pieces B_WO + 4 u .. + 4, two groups in flight
      const int t2 = u / (NB / 4), jb = u - t2 * (NB / 4);
      Frag n[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) n[q] = (u + 2 < 2 * NB / 4) ? frag<Frag>(sl, B_WO + (u + 2) * 4 + q) : wo1[q];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i) oacc[i][jb * 4 + q] = mfma(wo[q], o_op[i][t2], oacc[i][jb * 4 + q]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) { wo[q] = wo1[q]; wo1[q] = n[q]; }
    }
  };

  TB_MARK(1);
  for (int h = 0; h + 1 < H_; ++h) {
    dma_landed_barrier();                                     // stage A of head h landed; slot 1 is free (stage B of head h - 1 is done)
    TB_MARK(2 + 3 * (h & 1));
    stage_a(2 * h);
    TB_MARK(3 + 3 * (h & 1));
    dma_landed_barrier();
    TB_MARK(4 + 3 * (h & 1));
    stage_b(2 * h + 1, std::false_type{});
    TB_MARK(8 + (h & 1));
  }
  dma_landed_barrier();
  TB_MARK(10);
  stage_a(2 * H_ - 2);
  dma_landed_barrier();
  TB_MARK(11);
  stage_b(2 * H_ - 1, std::true_type{});
  TB_MARK(12);

  // ---- epilogue: LDS tile [pixel][frame][C + pad] ------------------------------------------------------------------------------
  dma_landed_barrier();                                       // residual tile landed; every wave is done with the ring
  TB_MARK(13);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      T* a = reinterpret_cast<T*>(smem + ((2 * wave + i) * 16 + r16) * TP) + j * 16 + g * 4;
      const f32x4 bo = *reinterpret_cast<const f32x4*>(p.b_out + j * 16 + g * 4);
      float rr[4], vv[4];
      ElemIO<T>::ld4(a, rr);
#pragma unroll
      for (int r = 0; r < 4; ++r) vv[r] = oacc[i][j][r] + bo[r] + rr[r];
      ElemIO<T>::st4(a, vv);
    }
  __syncthreads();
  {
    const int cg = tid % 40, rsl = tid / 40;                  // column group of 8 channels, row slice: rows rsl, rsl + 6, ...
    if (rsl < 6) {
#pragma unroll 2
      for (int row = rsl; row < ROWS; row += 6) {
        const u32x4 vv = *reinterpret_cast<const u32x4*>(smem + row * TP + cg * 16);
        *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(ob + (row & 15) * fstride + (row >> 4) * C_) + cg * 16) = vv;
      }
    }
  }
  TB_MARK(14);
}

}  // namespace

#ifdef TB_TIMING
extern "C" int fyc_tb_timing(unsigned long long* host_out, int n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_tb_time), sizeof(unsigned long long) * n);
}
#endif

int64_t fyc_temporal_block_rr_wstream_bytes() { return (int64_t)NSTAGE * STAGE_BYTES; }
int64_t fyc_temporal_block_rr_lds_bytes() { return LDS_BYTES; }

// called by fyc_temporal_block (temporal_block.hip) after its argument checks when a->wstream is set
int fyc_temporal_block_rr_launch(const fyc_temporal_block_args* a, void* stream) {
  FYC_REQUIRE(((uintptr_t)a->x % 16) == 0 && ((uintptr_t)a->out % 16) == 0 && ((uintptr_t)a->wstream % 16) == 0 && ((uintptr_t)a->b_out % 16) == 0,
              "fyc_temporal_block: operands must be 16-byte aligned");
  TRP p;
  p.x = (const bf16_t*)a->x; p.out = (bf16_t*)a->out; p.ws = (const char*)a->wstream; p.b_out = a->b_out; p.pixels = a->pixels;
  p.scale_log2e = a->scale * 1.44269504088896340736f; p.eps = a->eps;
  {
    constexpr int kMaxDev = 64;
    static std::mutex mu;
    static bool attr_done[kMaxDev] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    if (dev < 0 || dev >= kMaxDev || !attr_done[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tblock_rr_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(tblock_rr_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      if (e != hipSuccess) FYC_FAIL(-3, "fyc_temporal_block: %d bytes of dynamic LDS refused: %s", LDS_BYTES, hipGetErrorString(e));
      if (dev >= 0 && dev < kMaxDev) attr_done[dev] = true;
    }
  }
  if (a->dtype == FYC_F16) hipLaunchKernelGGL(tblock_rr_kernel<f16_t>, dim3((unsigned)(a->clips * (a->pixels / PIX))), dim3(NT), LDS_BYTES, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(tblock_rr_kernel<bf16_t>, dim3((unsigned)(a->clips * (a->pixels / PIX))), dim3(NT), LDS_BYTES, (hipStream_t)stream, p);
  FYC_CHECK_LAUNCH("fyc_temporal_block (register-resident)");
  return 0;
}
