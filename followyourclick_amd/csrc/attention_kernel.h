// Flash-style attention for gfx950: bf16 (or f16: template parameter T) MFMA QK^T / PV with wave-level online softmax (kernel template).
//
// Layouts (written by the GEMM "HEADS" epilogue): q,k [B*H][N][d] ; vt [B*H][d][ldvt] (V transposed,
// keys contiguous) ; o token-major [B*N][ldo] (channel = h*d + i) for the to_out GEMM.
//
// Work split: workgroup = 4 wave64 = 64*QT queries of one (batch, head); each wave owns QT tiles of 16 queries for the
// whole key loop.  K and V^T tiles of 64 keys go global->LDS by DMA (global_load_lds, 16 B/lane) into a 3-deep ring:
// the wait that retires tile j is counted (vmcnt(LOADS): tile j+1 stays in flight), the barrier is the raw s_barrier
// (cdna_hip_programming.md "Pipelining across barriers") and tile j+2 is issued right behind it.
//
// MFMA orientation (v_mfma_f32_16x16x32_bf16, D[row][col]: col = lane&15, row = 4*(lane>>4)+reg):
//   S^T = K Q^T  : A = K rows (keys), B = Q rows (queries)  -> a lane holds, for ONE query (col), scores of 4 keys per
//                  tile; two tiles with interleaved key rows give it 8 consecutive keys 8g..8g+7 of a 32-key block.
//   O^T = V^T P^T: B = P^T is exactly those 8 scores (exponentiated, bf16) - no cross-lane traffic,
//                  A = V^T rows (dv) x 8 consecutive keys = one 16-B LDS read.
// The head dim is covered by d/32 full k-steps plus one 16-wide step (v_mfma_f32_16x16x16_bf16) when d%32 is 8 or 16:
// d = 40 issues 32+16 instead of 64.
//
// Softmax on a VALU diet (the loop is VALU-bound before it is MFMA-bound at d = 40):
//   * Q is pre-multiplied by scale*log2(e): a score needs no multiply;
//   * the row sum l is a row of O^T: V^T gets a row of ones at index d, so the PV MFMAs accumulate sum_k p_k (of the
//     bf16-rounded p that also multiply V) - no adds, and l is rescaled together with O^T;
//   * when index d is a padding slot of the head dim (d % 16 == 8, e.g. 40), K gets a column of ones there and Q the
//     value -m (running max, bf16): the QK^T MFMA delivers s - m directly, so p = exp2(s') with no subtract;
//   * the running max is only moved when a score exceeds it by more than RESCALE_THR (p stays <= 2^THR), checked with
//     a wave-uniform ballot: the O^T rescale runs in the first tiles only.
#pragma once
#include <type_traits>

#include "fyc_common.h"

namespace fyca {

struct AttnP {
  const bf16_t* q; const bf16_t* k; const bf16_t* vt; bf16_t* o;   // 16-bit elements of the kernel's T (bf16 or f16: same pointer arithmetic)
  int batch, heads, n_q, n_k, d, ldo, ldvt, kv_batch_div, o_accumulate;
  float sl2e, o_scale;   // scale * log2(e)
  int nqb;               // query blocks per (b,h)
  const char* zero;
};

// 64 x bf16 1.0 (a V^T row of ones), then {1.0, 0 x 7} (the K column of ones): sources of the direct-to-LDS loads
#define FYC_1 0x3F80
static __device__ __attribute__((aligned(16))) const unsigned short fyc_ones[72] = {
    FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1,
    FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1,
    FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1,
    FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1,
    FYC_1, 0, 0, 0, 0, 0, 0, 0};
#undef FYC_1
#define FYC_1 0x3C00
static __device__ __attribute__((aligned(16))) const unsigned short fyc_ones_f16[72] = {
    FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1,
    FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1,
    FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1,
    FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1, FYC_1,
    FYC_1, 0, 0, 0, 0, 0, 0, 0};
#undef FYC_1
template <typename T> __device__ __forceinline__ const unsigned short* ones_table();
template <> __device__ __forceinline__ const unsigned short* ones_table<bf16_t>() { return fyc_ones; }
template <> __device__ __forceinline__ const unsigned short* ones_table<f16_t>() { return fyc_ones_f16; }

typedef __attribute__((ext_vector_type(4))) short s16x4;

// the two MFMA shapes of the loop per element type: 16x16x32 on 8-element fragments, 16x16x16 on 4-element ones (raw 16-bit words)
template <typename T> struct AttnMma;
template <> struct AttnMma<bf16_t> {
  __device__ static __forceinline__ f32x4 k32(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  __device__ static __forceinline__ f32x4 k16(s16x4 a, s16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
};
template <> struct AttnMma<f16_t> {
  __device__ static __forceinline__ f32x4 k32(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  __device__ static __forceinline__ f32x4 k16(s16x4 a, s16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
  }
};

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#ifndef FYC_ATTN_BUILTIN_ISSUE
// The (up to) four exec-masked DMAs of a full K / V^T tile as ONE statement: the lane masks are computed once per kernel (SGPR pairs), a DMA
// whose mask is empty in this wave is branched over (so that the counted vmcnt waits stay exact), M0 and EXEC leave as they came.  The
// compiler's form - v_cmp + s_and_saveexec + s_cbranch + 64-bit address add + s_mov m0 + s_nop + DMA + s_or exec per DMA - is ~50 instructions
// per tile, ~35 of them SALU, in a loop that is bound by instruction issue (profiles/r05_attention_pmc.txt); this is 24.  Measured
// (profiles/r05_attention_instruction_diet_ab.txt, d = 40, N = 4096): 665 -> 693 TFLOP/s without the s_setprio pairs and the sNaN-quieting
// maxima, -> 708-717 with this statement.  -DFYC_ATTN_BUILTIN_ISSUE rebuilds the compiler's form.
__device__ __forceinline__ void dma4_masked(unsigned long long m0_, unsigned long long m1_, unsigned long long m2_, unsigned long long m3_,
                                            unsigned o0, unsigned o1, unsigned o2, unsigned o3,
                                            const char* b0, const char* b1, const char* b2, const char* b3,
                                            unsigned l0, unsigned l1, unsigned l2, unsigned l3) {
  unsigned keep_m0;
  unsigned long long keep_exec;
  asm volatile("s_mov_b32 %[km0], m0\n\ts_mov_b64 %[kex], exec\n\t"
               "s_mov_b64 exec, %[ma]\n\ts_cbranch_execz 1f\n\ts_mov_b32 m0, %[la]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[oa], %[ba]\n"
               "1:\n\ts_mov_b64 exec, %[mb]\n\ts_cbranch_execz 2f\n\ts_mov_b32 m0, %[lb]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[ob], %[bb]\n"
               "2:\n\ts_mov_b64 exec, %[mc]\n\ts_cbranch_execz 3f\n\ts_mov_b32 m0, %[lc]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[oc], %[bc]\n"
               "3:\n\ts_mov_b64 exec, %[md]\n\ts_cbranch_execz 4f\n\ts_mov_b32 m0, %[ld]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[od], %[bd]\n"
               "4:\n\ts_mov_b64 exec, %[kex]\n\ts_mov_b32 m0, %[km0]"
               : [km0] "=&s"(keep_m0), [kex] "=&s"(keep_exec)
               : [ma] "s"(m0_), [mb] "s"(m1_), [mc] "s"(m2_), [md] "s"(m3_), [oa] "v"(o0), [ob] "v"(o1), [oc] "v"(o2), [od] "v"(o3),
                 [ba] "s"(b0), [bb] "s"(b1), [bc] "s"(b2), [bd] "s"(b3), [la] "s"(l0), [lb] "s"(l1), [lc] "s"(l2), [ld] "s"(l3)
               : "memory");
}
#endif

// Round 5: the loop is bound by instruction ISSUE (profiles/r05_attention_pmc.txt: 150 instructions per 21 MFMAs, three waves per SIMD whose
// issue-active time adds up to the kernel's duration), so instructions that only re-order work between the waves of a SIMD are pure cost:
// the s_setprio pairs around the MFMA groups (8 per 64-key tile) are compiled in only with -DFYC_ATTN_SETPRIO.
#ifdef FYC_ATTN_SETPRIO
#define FYC_ATTN_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define FYC_ATTN_PRIO(x) do { } while (0)
#endif

constexpr float RESCALE_THR = 6.0f;   // log2 units: probabilities stay <= 64 between moves of the running max

// max over the 4 lane quads that share a query column: xor 16 inside each 32-lane half (ds_swizzle bit mode, no LDS
// traffic), then across the halves (v_permlane32_swap)
__device__ __forceinline__ float quad_max(float mx) {
  mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, mx), 0x401F)));
  return swap32_max(mx);      // (fyc_common.h: NOT fmaxf over the builtin's two results - hipcc folds that to the first one)
}

// DP16: padded head dim / 16 (K-dim of QK^T).  DVT: 16-row blocks of O^T = d/16 + 1 (the extra row at index d is l).
// DVT == DP16 <=> d % 16 == 8: index d is a free padding slot of the head dim, used for the in-MFMA max subtraction.
template <typename T, int DP16, int DVT, int QT>
__global__ void __launch_bounds__(256, (DP16 == 3 && DVT == 3 && QT == 3) ? 3 : 1) fyc_attn_kernel(const AttnP p) {   // (d = 40, the 5-ms shape: held to the 168 registers of three waves per SIMD)
  typedef typename Pair16<T>::Vec8 Frag;
  const unsigned short* const fyc_ones = ones_table<T>();
  constexpr int KS = DP16 / 2;           // full 32-wide k-steps
  constexpr bool TAIL = (DP16 & 1) != 0; // plus one 16-wide k-step
  constexpr bool MSUB = (DVT == DP16);
  constexpr bool PIPE = DP16 <= 3;       // d <= 48: both 32-key blocks of a tile are software-pipelined (see the tile loop)
  constexpr int DC = DP16 * 2;           // 16-B chunks per K row
  constexpr bool KXOR = (DC == 8);       // 128-B rows: XOR swizzle; otherwise one pad chunk per row (odd pitch)
  constexpr int PC = KXOR ? 8 : DC + 1;  // K row pitch in chunks
  constexpr int KB = 64;                 // keys per LDS tile
  constexpr int K_IT = (KB * PC + 255) / 256, K_BYTES = K_IT * 256 * 16;
  constexpr int V_ROWS = DVT * 16, V_IT = (V_ROWS * 8 + 255) / 256, V_BYTES = V_IT * 256 * 16;
  constexpr int STAGE = K_BYTES + V_BYTES, LOADS = K_IT + V_IT, NS = 3;
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
  const T* Q = reinterpret_cast<const T*>(p.q) + (long long)bh * p.n_q * p.d;
  const T* K = reinterpret_cast<const T*>(p.k) + (long long)kvb * p.n_k * p.d;
  const T* VT = reinterpret_cast<const T*>(p.vt) + (long long)kvb * p.d * p.ldvt;
  const char* zero = p.zero;
  const int ntiles = (p.n_k + KB - 1) / KB;
  const int dchunks = p.d >> 3;

  // ---- K / V^T tile loaders.  What a lane fetches is the same in every tile except for a wave-uniform base: lanes that carry
  // data keep a 32-bit byte offset (global_load_lds with an SGPR base), lanes that carry constants (head-dim padding, the ones
  // column / row, rows beyond the tile) are written ONCE per ring stage by `issue_const` and masked off afterwards - the
  // per-tile issue block is 4 DMAs + their exec masks instead of ~100 VALU compares / selects / 64-bit adds.  Only a ragged
  // last tile takes the general, fully predicated path (`issue_general`).
  // Per-thread byte offsets of the data lanes (0xffffffff = this lane carries a constant).  With <= 4 DMA instructions per
  // tile they live in LDS behind the ring (one ds_read_b128 per tile) instead of in 4 VGPRs + masks: at 64 queries per wave
  // and d = 40 the kernel sits exactly at the 168-register limit of 3 waves per SIMD.
  constexpr bool OFFS_IN_LDS = (K_IT + V_IT) <= 4;
  constexpr unsigned NODATA = 0xffffffffu;
  unsigned k_off[K_IT], v_off[V_IT];
  int my_loads = 0;            // DMA instructions this WAVE issues per full tile (instructions with no data lane are skipped)
#pragma unroll
  for (int it = 0; it < K_IT; ++it) {
    const int L = it * 256 + tid, row = L / PC, cc = L - row * PC;
    const int c = KXOR ? (cc ^ (row & 7)) : cc;
    k_off[it] = (row < KB && c < dchunks) ? (unsigned)(row * p.d + c * 8) * 2u : NODATA;
    my_loads += __builtin_amdgcn_ballot_w64(k_off[it] != NODATA) != 0 ? 1 : 0;
  }
#pragma unroll
  for (int it = 0; it < V_IT; ++it) {
    const int L = it * 256 + tid, row = L >> 3, cc = L & 7;
    const int c = cc ^ (row & 7);
    v_off[it] = (row < p.d) ? (unsigned)(row * p.ldvt + c * 8) * 2u : NODATA;
    my_loads += __builtin_amdgcn_ballot_w64(v_off[it] != NODATA) != 0 ? 1 : 0;
  }
  my_loads = __builtin_amdgcn_readfirstlane(my_loads);
#ifndef FYC_ATTN_BUILTIN_ISSUE
  unsigned long long dmask[4] = {0ull, 0ull, 0ull, 0ull};      // data lanes of the (K_IT + V_IT <= 4) DMAs of a full tile, this wave
  if (K_IT + V_IT <= 4) {
#pragma unroll
    for (int it = 0; it < K_IT; ++it) dmask[it] = __builtin_amdgcn_ballot_w64(k_off[it] != NODATA);
#pragma unroll
    for (int it = 0; it < V_IT; ++it) dmask[K_IT + it] = __builtin_amdgcn_ballot_w64(v_off[it] != NODATA);
  }
  const unsigned lds_ring = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
#endif
  char* const offs_lds = smem + NS * STAGE + tid * 16;
  if (OFFS_IN_LDS) {
    u32x4 o4 = {NODATA, NODATA, NODATA, NODATA};
#pragma unroll
    for (int it = 0; it < K_IT; ++it) o4[it] = k_off[it];
#pragma unroll
    for (int it = 0; it < V_IT; ++it) o4[K_IT + it] = v_off[it];
    *reinterpret_cast<u32x4*>(offs_lds) = o4;      // only this thread reads it back: no barrier needed
  }
  auto issue_const = [&](int stage) {
    char* sK = smem + stage * STAGE;
    char* sV = sK + K_BYTES;
#pragma unroll
    for (int it = 0; it < K_IT; ++it) {
      const int L = it * 256 + tid, row = L / PC, cc = L - row * PC;
      const int c = KXOR ? (cc ^ (row & 7)) : cc;
      if (k_off[it] == NODATA) glds16((MSUB && row < KB && c == dchunks) ? (const void*)(fyc_ones + 64) : (const void*)zero, sK + (it * 256 + wave * 64) * 16);
    }
#pragma unroll
    for (int it = 0; it < V_IT; ++it) {
      const int row = (it * 256 + tid) >> 3;
      if (v_off[it] == NODATA) glds16(row == p.d ? (const void*)fyc_ones : (const void*)zero, sV + (it * 256 + wave * 64) * 16);
    }
  };
  auto issue_full = [&](int tile, int stage) {          // all KB keys of the tile exist
    char* sK = smem + stage * STAGE;
    char* sV = sK + K_BYTES;
    const char* kb = reinterpret_cast<const char*>(K) + (long long)tile * (KB * 2) * p.d;
    const char* vb = reinterpret_cast<const char*>(VT) + (long long)tile * (KB * 2);
#ifndef FYC_ATTN_BUILTIN_ISSUE
    if constexpr (OFFS_IN_LDS) {
      const u32x4 o4 = *reinterpret_cast<const u32x4*>(offs_lds);
      const unsigned lk = lds_ring + stage * STAGE + wave * 1024, lv = lk + K_BYTES;
      auto base = [&](int i) { return i < K_IT ? kb : vb; };
      auto ldst = [&](int i) { return i < K_IT ? lk + i * 4096 : lv + (i - K_IT) * 4096; };
      dma4_masked(dmask[0], dmask[1], dmask[2], dmask[3], o4[0], o4[1], o4[2], o4[3], base(0), base(1), base(2), base(3), ldst(0), ldst(1), ldst(2), ldst(3));
      return;
    }
#endif
    unsigned ko[K_IT], vo[V_IT];
    if (OFFS_IN_LDS) {
      const u32x4 o4 = *reinterpret_cast<const u32x4*>(offs_lds);
#pragma unroll
      for (int it = 0; it < K_IT; ++it) ko[it] = o4[it];
#pragma unroll
      for (int it = 0; it < V_IT; ++it) vo[it] = o4[K_IT + it];
    } else {
#pragma unroll
      for (int it = 0; it < K_IT; ++it) ko[it] = k_off[it];
#pragma unroll
      for (int it = 0; it < V_IT; ++it) vo[it] = v_off[it];
    }
#pragma unroll
    for (int it = 0; it < K_IT; ++it)
      if (ko[it] != NODATA) glds16(kb + ko[it], sK + (it * 256 + wave * 64) * 16);
#pragma unroll
    for (int it = 0; it < V_IT; ++it)
      if (vo[it] != NODATA) glds16(vb + vo[it], sV + (it * 256 + wave * 64) * 16);
  };
  auto issue_general = [&](int tile, int stage) {       // ragged last tile: every lane predicated, every lane issues
    char* sK = smem + stage * STAGE;
    char* sV = sK + K_BYTES;
    const int key0 = tile * KB;
#pragma unroll
    for (int it = 0; it < K_IT; ++it) {
      const int L = it * 256 + tid, row = L / PC, cc = L - row * PC;
      const int c = KXOR ? (cc ^ (row & 7)) : cc;
      const int key = key0 + row;
      const bool live = row < KB && key < p.n_k;
      const void* src = zero;
      if (live && c < dchunks) src = K + (long long)key * p.d + c * 8;
      else if (MSUB && live && c == dchunks) src = fyc_ones + 64;    // {1, 0 x 7}: the column that adds -m to every score
      glds16(src, sK + (it * 256 + wave * 64) * 16);
    }
#pragma unroll
    for (int it = 0; it < V_IT; ++it) {
      const int L = it * 256 + tid, row = L >> 3, cc = L & 7;
      const int c = cc ^ (row & 7);
      const int key = key0 + c * 8;
      const void* src = zero;
      if (row < p.d) { if (key < p.ldvt) src = VT + (long long)row * p.ldvt + key; }
      else if (row == p.d) src = fyc_ones;                            // the row of ones that makes O^T[d] = sum_k p_k
      glds16(src, sV + (it * 256 + wave * 64) * 16);
    }
  };
  const int nfull = p.n_k / KB;                          // tiles 0 .. nfull-1 are full
  auto issue = [&](int tile, int stage) { if (tile < nfull) issue_full(tile, stage); else issue_general(tile, stage); };
  // DMA instructions of tile t still allowed in flight when tile t-1 is awaited
  auto loads_of = [&](int tile) { return tile < nfull ? my_loads : LOADS; };
  auto wait_dyn = [&](int n) {
    switch (n) {
      case 0: wait_vmcnt<0>(); break;   case 1: wait_vmcnt<1>(); break;   case 2: wait_vmcnt<2>(); break;   case 3: wait_vmcnt<3>(); break;
      case 4: wait_vmcnt<4>(); break;   case 5: wait_vmcnt<5>(); break;   case 6: wait_vmcnt<6>(); break;   case 7: wait_vmcnt<7>(); break;
      case 8: wait_vmcnt<8>(); break;   case 9: wait_vmcnt<9>(); break;   case 10: wait_vmcnt<10>(); break; case 11: wait_vmcnt<11>(); break;
      default: wait_vmcnt<12>(); break;
    }
  };
  static_assert(LOADS <= 12, "wait_dyn covers up to 12 DMA instructions per tile");
#pragma unroll
  for (int st = 0; st < NS; ++st) issue_const(st);
  issue(0, 0);
  if (ntiles > 1) issue(1, 1);

  // ---- Q fragments (B operand), pre-scaled by scale*log2(e): lane (query = r16, quad g) holds Q[query][32ks + 8g .. +8]
  //      and, for the 16-wide step, Q[query][32*KS + 4g .. +4]
  const int qbase = qb * (64 * QT) + wave * (16 * QT);
  Frag qf[QT][KS > 0 ? KS : 1];
  s16x4 qt4[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int query = qbase + qt * 16 + r16;
    const bool qok = query < p.n_q;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int dd = 32 * ks + 8 * g;
      float v[8];
      load8<T>((qok && dd < p.d) ? Q + (long long)query * p.d + dd : reinterpret_cast<const T*>(zero), v);
      u32x4 pk;
#pragma unroll
      for (int i = 0; i < 4; ++i) pk[i] = Pair16<T>::pack(v[2 * i] * p.sl2e, v[2 * i + 1] * p.sl2e);
      qf[qt][ks] = __builtin_bit_cast(Frag, pk);
    }
    if (TAIL) {
      const int dd = 32 * KS + 4 * g;
      float v[4];
      ElemIO<T>::ld4((qok && dd < p.d) ? Q + (long long)query * p.d + dd : reinterpret_cast<const T*>(zero), v);
      u32x2 pk = {Pair16<T>::pack(v[0] * p.sl2e, v[1] * p.sl2e), Pair16<T>::pack(v[2] * p.sl2e, v[3] * p.sl2e)};
      qt4[qt] = __builtin_bit_cast(s16x4, pk);
    }
  }
  // MSUB: the slot of head-dim index d inside this lane's fragments (element 0 of quad MG of the last k-step)
  constexpr int MOFF = MSUB ? ((DP16 * 16 - 8) - (TAIL ? 32 * KS : 32 * (KS - 1))) : 0;   // d - first index of the last step
  constexpr int MG = TAIL ? MOFF / 4 : MOFF / 8;
  auto set_neg_max = [&](int qt, float m_bf16_exact) {       // writes -m into Q'[query][d]
    const unsigned short bits = Pair16<T>::to_bits(-m_bf16_exact);
    if (g == MG) {
      if (TAIL) qt4[qt][0] = (short)bits;
      else {
        u32x4 t = __builtin_bit_cast(u32x4, qf[qt][KS > 0 ? KS - 1 : 0]);
        t[0] = (t[0] & 0xffff0000u) | bits;
        qf[qt][KS > 0 ? KS - 1 : 0] = __builtin_bit_cast(Frag, t);
      }
    }
  };

  f32x4 o[QT][DVT];
  float m_run[QT];        // running max in log2 units (MSUB: exactly representable in bf16, mirrored in Q')
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m_run[qt] = 0.f;
#pragma unroll
    for (int dv = 0; dv < DVT; ++dv) o[qt][dv] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  bool started = false;   // the first key block always fixes the running max (scores may sit anywhere)

  int st_c = 0, st_i = (ntiles > 1) ? 2 : 1;
  // One tile of the key loop.  STEADY (round 5, the issue-bound loop's instruction diet): tile + 2 is a FULL tile - so is tile + 1, nothing is
  // ragged, both 32-key blocks hold keys - and every per-tile question the general body asks (is there a tile to issue, which issue path, is
  // this the ragged tile, one block or two) is answered at compile time: the steady-state loop below carries none of those compares / branches.
  auto tile_body = [&](const int tile, auto steady_c) __attribute__((always_inline)) {
    constexpr bool STEADY = decltype(steady_c)::value;
    // tile landed; the next one may stay in flight.  (Steady state: most waves issue all LOADS instructions of a tile - one compare instead of
    // the 13-way switch of wait_dyn, whose compare tree costs ~8 scalar instructions per tile)
    if (STEADY && my_loads == LOADS) wait_vmcnt<LOADS>();
    else if (STEADY) wait_dyn(my_loads);
    else wait_dyn(tile + 1 < ntiles ? loads_of(tile + 1) : 0);
    __builtin_amdgcn_s_barrier();
    if (STEADY) { issue_full(tile + 2, st_i); st_i = (st_i + 1 == NS) ? 0 : st_i + 1; }
    else if (tile + 2 < ntiles) { issue(tile + 2, st_i); st_i = (st_i + 1 == NS) ? 0 : st_i + 1; }
    const char* sK = smem + st_c * STAGE;
    const char* sV = sK + K_BYTES;
    st_c = (st_c + 1 == NS) ? 0 : st_c + 1;
    const bool tail = STEADY ? false : (tile * KB + KB > p.n_k);
    // Two 32-key blocks per tile, software-pipelined inside the wave: both blocks' QK^T MFMAs are issued before the first
    // block's softmax, so the matrix pipe works on block 1's scores while the VALU does block 0's exponentials, and block 0's
    // PV MFMAs run under block 1's softmax (a wave issues in order: without this its MFMA and VALU phases simply add up -
    // measured: matrix pipe 43 % busy, VALU 21 %, the rest dependency stalls).
    auto qk = [&](int kb, f32x4 (&s)[2][QT]) {
      // S^T tiles: tile t row i <-> key kb*32 + 8*(i>>2) + 4t + (i&3).
      // The 16-wide steps of all 2*QT score tiles first (onto zero), then the 32-wide chains on top, with the two groups pinned:
      // hipcc 7.2 puts no wait state between a v_mfma_f32_16x16x32_bf16 and a v_mfma_f32_16x16x16_bf16 that accumulates onto its
      // result (or vice versa), and the consumer then reads a stale accumulator (measured: head dims with full steps + a tail
      // came out wrong whenever the two were adjacent).  This order puts 2*QT-1 independent MFMAs (>= 24 cycles) behind each
      // 4-pass producer before its accumulator is read again; the 32-wide chains are same-opcode back-to-back (interlocked).
      const char* kr[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) kr[t] = sK + (kb * 32 + 8 * (r16 >> 2) + 4 * t + (r16 & 3)) * (PC * 16);
      s16x4 kt4[2];
      if (TAIL) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int krow = kb * 32 + 8 * (r16 >> 2) + 4 * t + (r16 & 3);
          const int chunk = 4 * KS + (g >> 1);
          kt4[t] = *reinterpret_cast<const s16x4*>(kr[t] + (KXOR ? (chunk ^ (krow & 7)) : chunk) * 16 + (g & 1) * 8);
        }
      }
      FYC_ATTN_PRIO(1);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          s[t][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (TAIL) s[t][qt] = AttnMma<T>::k16(kt4[t], qt4[qt], s[t][qt]);
        }
      if (TAIL && KS > 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int krow = kb * 32 + 8 * (r16 >> 2) + 4 * t + (r16 & 3);
        Frag kf[KS > 0 ? KS : 1];        // one key tile's fragments at a time: large head dims would not fit both
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int chunk = 4 * ks + g;
          kf[ks] = *reinterpret_cast<const Frag*>(kr[t] + (KXOR ? (chunk ^ (krow & 7)) : chunk) * 16);
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) s[t][qt] = AttnMma<T>::k32(kf[ks], qf[qt][ks], s[t][qt]);
      }
      FYC_ATTN_PRIO(0);
    };
    auto softmax_pv = [&](int kb, f32x4 (&s)[2][QT], f32x4 (&sn)[2][QT], bool has_next) {
      // ---- online softmax; lane holds keys kb*32 + 8g + 4t + r of query r16.  MSUB: s already is score - m_run.
      float mx[QT];        // this LANE's maximum over its 8 keys (the other three quads of the query column hold the other 24)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        if (tail) {
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (tile * KB + kb * 32 + 8 * g + 4 * t + r >= p.n_k) s[t][qt][r] = -INFINITY;
        }
        const float m0 = fmaxf(fmaxf(s[0][qt][0], s[0][qt][1]), s[0][qt][2]);
        const float m1 = fmaxf(fmaxf(s[0][qt][3], s[1][qt][0]), s[1][qt][1]);
        mx[qt] = fmaxf(fmaxf(m0, m1), fmaxf(s[1][qt][2], s[1][qt][3]));
      }
      // does ANY lane of the wave see a score above its running max + THR?  (the per-lane maxima suffice for the decision;
      // the cross-quad reduction that makes the four quads of a query column agree on the new max runs on the rare path only)
      bool move = !started;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) move = move || (MSUB ? mx[qt] > RESCALE_THR : mx[qt] > m_run[qt] + RESCALE_THR);
      float shift[QT];     // what still has to be subtracted from the scores of THIS block
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) shift[qt] = MSUB ? 0.f : m_run[qt];
      if (__builtin_amdgcn_ballot_w64(move) != 0) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) mx[qt] = quad_max(mx[qt]);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          const bool mv = !started || (MSUB ? mx[qt] > RESCALE_THR : mx[qt] > m_run[qt] + RESCALE_THR);
          if (MSUB) {
            // new max = bf16(m_run + mx) so that it fits Q' exactly; this block's scores are still relative to the old one
            const float m_new = mv ? round_through<T>(m_run[qt] + mx[qt]) : m_run[qt];
            const float delta = m_new - m_run[qt];
            const float alpha = started ? __builtin_amdgcn_exp2f(-delta) : 1.0f;    // first block: O^T is 0, and 2^-delta may overflow
            m_run[qt] = m_new;
            shift[qt] = delta;
            set_neg_max(qt, m_new);
            if (has_next) {      // the next block's scores were computed against the old max as well
#pragma unroll
              for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) sn[t][qt][r] -= delta;
            }
#pragma unroll
            for (int dv = 0; dv < DVT; ++dv) { o[qt][dv][0] *= alpha; o[qt][dv][1] *= alpha; o[qt][dv][2] *= alpha; o[qt][dv][3] *= alpha; }
          } else {
            const float m_new = mv ? mx[qt] : m_run[qt];
            const float alpha = started ? __builtin_amdgcn_exp2f(m_run[qt] - m_new) : 1.0f;
            m_run[qt] = m_new;
            shift[qt] = m_new;
#pragma unroll
            for (int dv = 0; dv < DVT; ++dv) { o[qt][dv][0] *= alpha; o[qt][dv][1] *= alpha; o[qt][dv][2] *= alpha; o[qt][dv][3] *= alpha; }
          }
        }
        started = true;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[t][qt][r] -= shift[qt];
      } else if (!MSUB) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[t][qt][r] -= shift[qt];
      }
      Frag pf[QT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        u32x4 pk;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          pk[2 * t] = Pair16<T>::pack(__builtin_amdgcn_exp2f(s[t][qt][0]), __builtin_amdgcn_exp2f(s[t][qt][1]));
          pk[2 * t + 1] = Pair16<T>::pack(__builtin_amdgcn_exp2f(s[t][qt][2]), __builtin_amdgcn_exp2f(s[t][qt][3]));
        }
        pf[qt] = __builtin_bit_cast(Frag, pk);
      }
      // ---- O^T += V^T P^T  (row d of V^T is all ones: O^T[d] accumulates the row sums)
      FYC_ATTN_PRIO(1);
#pragma unroll
      for (int dv = 0; dv < DVT; ++dv) {
        const int vrow = dv * 16 + r16;
        const int chunk = kb * 4 + g;
        const Frag vf = *reinterpret_cast<const Frag*>(sV + vrow * 128 + ((chunk ^ (vrow & 7)) * 16));
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) o[qt][dv] = AttnMma<T>::k32(vf, pf[qt], o[qt][dv]);
      }
      FYC_ATTN_PRIO(0);
    };
    const int nb = STEADY ? 2 : ((tile * KB + 32 < p.n_k) ? 2 : 1);      // 32-key blocks of this tile that hold keys (uniform)
    f32x4 s0[2][QT], s1[2][QT];
    if constexpr (PIPE) {
      qk(0, s0);
      if (nb > 1) { qk(1, s1); __builtin_amdgcn_sched_barrier(0); }
      softmax_pv(0, s0, s1, nb > 1);
      if (nb > 1) softmax_pv(1, s1, s0, false);
    } else {            // larger head dims: two blocks of scores in flight do not fit the register file
      qk(0, s0);
      softmax_pv(0, s0, s1, false);
      if (nb > 1) { qk(1, s0); softmax_pv(1, s0, s1, false); }
    }
  };
  const int n_steady = nfull > 2 ? nfull - 2 : 0;          // tiles t with t + 2 < nfull
  int tile = 0;
#ifndef FYC_ATTN_ONE_LOOP
  for (; tile < n_steady; ++tile) tile_body(tile, std::true_type());
#endif
  for (; tile < ntiles; ++tile) tile_body(tile, std::false_type());

  // ---- epilogue: lane holds O[query r16][dv*16 + 4g + r]; l = O^T[d] sits in quad (d%16)/4, register 0 of the last block
  const int lsrc = ((p.d & 15) >> 2) * 16 + r16;
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const float l = __shfl(o[qt][DVT - 1][0], lsrc);
    const float inv = 1.0f / l;
    const int query = qbase + qt * 16 + r16;
    if (query >= p.n_q) continue;
    T* orow = reinterpret_cast<T*>(p.o) + ((long long)b * p.n_q + query) * p.ldo + h * p.d;
#pragma unroll
    for (int dv = 0; dv < DVT; ++dv) {
      const int dd = dv * 16 + 4 * g;
      if (dd >= p.d) continue;
      float v[4] = {o[qt][dv][0] * inv, o[qt][dv][1] * inv, o[qt][dv][2] * inv, o[qt][dv][3] * inv};
      if (p.o_accumulate) {
        float prev[4];
        ElemIO<T>::ld4(orow + dd, prev);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = prev[r] + p.o_scale * v[r];
      }
      ElemIO<T>::st4(orow + dd, v);
    }
  }
}

template <typename T, int DP16, int DVT, int QT>
int launch_attn(const AttnP& p0, hipStream_t st) {
  constexpr int DC = DP16 * 2;
  constexpr int PC = (DC == 8) ? 8 : DC + 1;
  constexpr int K_BYTES = ((64 * PC + 255) / 256) * 256 * 16;
  constexpr int V_BYTES = ((DVT * 16 * 8 + 255) / 256) * 256 * 16;
  constexpr int smem = 3 * (K_BYTES + V_BYTES) + 256 * 16;      // ring + per-thread loader offsets
  static_assert(smem <= 160 * 1024, "LDS budget");
  auto kern = fyc_attn_kernel<T, DP16, DVT, QT>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);   // per call: cheap, and correct on every device
  AttnP p = p0;
  p.nqb = (p.n_q + 64 * QT - 1) / (64 * QT);
  dim3 grid(p.batch * p.heads * p.nqb);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, p);
  FYC_CHECK_LAUNCH("fyc_attention");
  return 0;
}

// one translation unit per (element type, group of head dims) keeps the build parallel: attention_{small,medium,large}[_f16].hip
// instantiate the templates of attention_groups.h
template <typename T> int run_small(const AttnP& p, int qt, hipStream_t st);     // d <= 48; qt = 16-query tiles per wave (2, 3, 4)
template <typename T> int run_medium(const AttnP& p, int qt, hipStream_t st);    // 48 < d <= 96
template <typename T> int run_large(const AttnP& p, hipStream_t st);              // 96 < d <= 160

// d -> (DP16, DVT) = (ceil(d/16), d/16 + 1), d a multiple of 8
#define FYC_ATTN_CASE(D, QT) case D: return launch_attn<T, (D + 15) / 16, D / 16 + 1, QT>(p, st)

}  // namespace fyca
