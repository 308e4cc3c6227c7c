// Ping-pong main loop for the 8-wave bf16 GEMM / implicit-GEMM convolution tiles (gfx950).  Same operands, LDS image, tile
// stream and epilogues as gemm_kernel.h (fyc_gemm_kernel); what changes is WHO does WHAT WHEN inside a K tile.
//
// What the compiler made of the one-phase loop (256x320 tile, 64x160 per wave, 160 accumulator registers): per MFMA k-step
// `11 ds_read_b128 - lgkmcnt(0) - 10 MFMA - [1 ds_read_b128 - lgkmcnt(0) - 10 MFMA] x 3`: with one A fragment buffer left
// by the register budget, every group of 10 MFMAs waits for a full LDS round trip, and both waves of a SIMD run the same phase
// at the same time.  Here the two waves of a SIMD alternate roles, separated by s_barrier (the guide's 8-phase idea with our
// tile): while the waves of group X (waves 0-3, one per SIMD) issue the 40 MFMAs of a k-step from registers, the waves of group Y
// (waves 4-7, their SIMD partners) read the 14 fragments of THEIR next k-step and issue DMA, and vice versa:
//
//   interval      4T          4T+1        4T+2        4T+3        4T+4
//   group X    R(T,0)+dmaA   M(T,0)      R(T,1)      M(T,1)|vm   R(T+1,0)+dmaA ...
//   group Y    M(T-1,1)      R(T,0)+dmaB M(T,0)      R(T,1)|vm   M(T,1)        ...
//
// R(T,s) = fragment reads of MFMA k-step s of K tile T, M = its MFMAs, | = s_barrier between intervals (every wave passes
// every barrier; Y runs one barrier behind X), `vm` = s_waitcnt vmcnt(0) of the wave's own DMA.  The DMA of K tile T+1 is
// split evenly: each group fetches ITS OWN half of the A rows (X: rows [0, BM/2), Y: the rest) and half of the weight rows, in
// its R(T,0) phase, and waits for it with vmcnt(0) in interval 4T+3 (X: 3 intervals of flight, Y: 2).
// Safety of the 2-deep ring, stage (T+1)%2 = the stage of K tile T-1: weight rows are only read in R phases, last by X in
// interval 4T-2 and by Y in 4T-1; a group's own A rows are read by that group only - the 256x320 tile keeps three A fragment
// buffers and streams row block 3 in behind the MFMAs of block 0 (ASTREAM below), so its last A reads are in M(T-1,1): X
// in 4T-1, Y in 4T.  Every read is retired (lgkmcnt(0)) before the barrier that ends its interval, so X may refill from 4T on
// and Y from 4T+1 on - exactly where their R(T,0) phases sit.  The new data is waited for in interval 4T+3 by the issuing
// waves and first read in 4T+4 (X) / 4T+5 (Y), one barrier later.
// Per output tile the groups are re-aligned (X waits one barrier at the end, Y one at the start) so that the epilogue's own
// barriers see all eight waves in the same place.
#pragma once
#include "../../../followyourclick_amd/csrc/gemm_kernel.h"

namespace fycg {

template <int BM, int BN, int WGM, int WGN, int MODE, int EPI>
__global__ void __launch_bounds__(512) fyc_gemm_pp_kernel(const GemmP p) {
  typedef bf16_t T;
  typedef Mma<T> Tr;
  typedef typename Tr::Frag Frag;
  constexpr int RB = 128, CH = 8, BK = 64, KSTEPS = 2;
  constexpr int WTM = BM / WGM / 16, WTN = BN / WGN / 16;
  constexpr int A_BYTES = BM * RB, STAGE = (BM + BN) * RB;
  constexpr int A_PIECES = BM / 32, B_PIECES = BN / 32;    // DMA instructions per loader thread per K tile (a group = 256 threads = 32 rows x 8 chunks)
  static_assert(WGM * WGN == 8 && BM % 32 == 0 && BN % 32 == 0, "8 waves, 32-row loader pieces");
  static_assert(EPI != FYC_EPI_GEGLU || WTN % 2 == 0, "GEGLU pairs value / gate column blocks inside a wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const bool grp_y = wave >= 4;            // wave-uniform: SIMD partner group (waves w and w + 4 share a SIMD)
  const int gw = wave & 3;                 // wave inside its group

  const int ntiles = p.tiles_m * p.tiles_n;
  auto remap = [&](int t) {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = t & 7, idx = t >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  };
  const T* __restrict__ A = reinterpret_cast<const T*>(p.a);
  const T* __restrict__ W = reinterpret_cast<const T*>(p.w);
  const T* __restrict__ A2 = reinterpret_cast<const T*>(p.a2);
  const T* zero = reinterpret_cast<const T*>(p.zero);

  // ---- loader state of the work item being ISSUED -----------------------------------------------------------------------------
  // a group = 256 threads = 32 rows x 8 chunks per DMA piece; group X issues A pieces [0, AH) (its own rows) and B pieces
  // [0, BX), group Y A pieces [AH, 2 AH) and B pieces [BX, B_PIECES)
  constexpr bool C3 = (MODE == FYC_GEMM_CONV3X3), UP = (MODE == FYC_GEMM_CONV3X3_UP2);
  static_assert(A_PIECES % 2 == 0, "each group loads its own half of the A rows");
  constexpr int AH = A_PIECES / 2;
  constexpr int BX = B_PIECES / 2, BP = B_PIECES - BX;     // BP: most B pieces a thread handles
  const int lt = tid & 255;                // thread inside its group: LDS row lt / 8 (+ 32 per piece), chunk lt % 8
  const int lrow = lt >> 3;
  const int koff = ((lt & 7) ^ (lrow & 7)) * CH;     // swizzled source chunk (the piece stride of 32 rows keeps the key)
  const int a_lo = grp_y ? AH : 0;
  const int b_lo = grp_y ? BX : 0, b_n = grp_y ? B_PIECES - BX : BX;
  int i_tm = 0, i_tn = 0;
  // 32-bit element offsets (host: every operand of a ping-pong launch has fewer than 2^32 elements): one register per base, and
  // the compiler's loop-invariant products of the conv gather stay 32-bit too (as 64-bit values they went to scratch)
  unsigned a_base = 0, a2_base = 0, b_base = 0;    // this thread's first row in a / a2 / w (plain GEMM: a, a2)
  unsigned amask = 0, bmask = 0;                   // piece `it` of the thread's share fetches a real row (inside M resp. N)
  int a_pos[C3 ? AH : 1], a_msk[C3 ? AH : 1];
  int a_pix[UP ? AH : 1], a_yx[UP ? AH : 1];
  int tap = 0, c0 = 0;
  const int KT = p.K / BK;                 // host: K % 64 == 0, K >= 128
  const int S = p.splitk > 1 ? p.splitk : 1;
  auto kt_begin = [&](int work) { return (int)(((long long)(work % S) * KT) / S); };
  auto kt_end = [&](int work) { return (int)(((long long)(work % S + 1) * KT) / S); };
  auto setup_issue = [&](int work) __attribute__((always_inline)) {
    const int t = remap(work / S);
    tile_coords(p, t, i_tm, i_tn);
    const int kt0 = kt_begin(work);
    tap = kt0 % 9; c0 = (kt0 / 9) * BK;
    if (MODE == FYC_GEMM_PLAIN || S == 1) { tap = 0; c0 = 0; }
    b_base = (unsigned)(i_tn * BN + lrow + b_lo * 32) * (unsigned)p.ldw + koff;
    bmask = 0;
#pragma unroll
    for (int it = 0; it < BP; ++it) bmask |= (it < b_n && i_tn * BN + lrow + (b_lo + it) * 32 < p.N) ? (1u << it) : 0u;
    const int m0 = i_tm * BM + lrow + a_lo * 32;
    if (MODE == FYC_GEMM_PLAIN) {
      a_base = (unsigned)m0 * (unsigned)p.lda + koff;
      a2_base = (unsigned)m0 * (unsigned)p.lda2 + koff - p.k_split;     // (+ k0 >= k_split at use: never negative)
      amask = 0;
#pragma unroll
      for (int it = 0; it < AH; ++it) amask |= (m0 + it * 32 < p.M) ? (1u << it) : 0u;
    } else {
#pragma unroll
      for (int it = 0; it < AH; ++it) {
        const int m = m0 + it * 32;
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
  auto src_a = [&](int it, int k0) __attribute__((always_inline)) -> const T* {      // `it`: index into the thread's own pieces
    if (MODE == FYC_GEMM_PLAIN) {
      if (!((amask >> it) & 1u)) return zero;
      if (A2 != nullptr && k0 >= p.k_split) return A2 + (size_t)(a2_base + (unsigned)(it * 32) * (unsigned)p.lda2 + k0);   // wave-uniform: k_split % 64 == 0
      return A + (size_t)(a_base + (unsigned)(it * 32) * (unsigned)p.lda + k0);
    } else if (C3) {
      const int ky = tap / 3, kx = tap - 3 * ky;
      const int pos = a_pos[it] + ky * p.Win + kx;
      return ((a_msk[it] >> tap) & 1) ? A + (size_t)((unsigned)pos * (unsigned)p.Cin + (c0 + koff)) : zero;
    } else {
      const int ky = tap / 3, kx = tap - 3 * ky;
      const int iy = (a_yx[it] >> 16) + ky, ix = (int)(short)(a_yx[it] & 0xffff) + kx;
      const bool ok = a_pix[it] >= 0 && (unsigned)iy < (unsigned)p.Hout && (unsigned)ix < (unsigned)p.Wout;
      int sy, sx;
      if (p.up_exact2) { sy = iy >> 1; sx = ix >> 1; }
      else {
        sy = min((int)floorf((float)iy * p.up_sh), p.Hin - 1);
        sx = min((int)floorf((float)ix * p.up_sw), p.Win - 1);
      }
      return ok ? A + (size_t)((unsigned)(a_pix[it] + sy * p.Win + sx) * (unsigned)p.Cin + c0 + koff) : zero;
    }
  };
  auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
    char* sS = smem + stage * STAGE;
    const int k0 = kt * BK;
    char* dstA = sS + gw * 1024 + a_lo * 4096;
#pragma unroll
    for (int it = 0; it < AH; ++it) glds16(src_a(it, k0), dstA + it * 4096);
    char* dstB = sS + A_BYTES + gw * 1024 + b_lo * 4096;
#pragma unroll
    for (int it = 0; it < BP; ++it)
      if (it < b_n) glds16(((bmask >> it) & 1u) ? W + (size_t)(b_base + (unsigned)(it * 32) * (unsigned)p.ldw + k0) : zero, dstB + it * 4096);
    if (MODE != FYC_GEMM_PLAIN) {
      if (++tap == 9) { tap = 0; c0 += BK; }
    }
  };

  f32x4 acc[WTM][WTN];
  FYC_STAMP_DECL;
  // Register budget of the 256x320 tile: 160 accumulators + 14 fragments (56) leave 40 registers for everything else, and the
  // gather state of the convolutions then spills INTO the issue block (a scratch reload + vmcnt(0) between two DMAs).  ASTREAM:
  // three A fragment buffers instead of four; the read phase fills them with row blocks 0..2, and the MFMA phase re-fills buffer
  // 0 with row block 3 right behind the 10 MFMAs of row block 0 - the read returns under the 20 MFMAs of blocks 1 and 2.  (With
  // two buffers the second refill sits right in front of its lgkmcnt(0) - hipcc does not count LDS waits while a DMA is in flight.)
  constexpr bool ASTREAM = (WTM * WTN * 4 + (WTM + WTN) * 4 > 200) && WTM == 4;
  constexpr int ABUF = ASTREAM ? 3 : WTM;
  Frag af[ABUF], bf[WTN];
  const int g = lane >> 4, r16 = lane & 15;
  const int sw = r16 & 7;
  const int frag_a = (wm * WTM * 16 + r16) * RB, frag_b = A_BYTES + (wn * WTN * 16 + r16) * RB;
  auto read_frags = [&](int stage, int s) __attribute__((always_inline)) {
    const char* sA = smem + stage * STAGE + frag_a;
    const char* sB = smem + stage * STAGE + frag_b;
    const int coff = ((4 * s + g) ^ sw) * 16;
#pragma unroll
    for (int j = 0; j < WTN; ++j) bf[j] = *reinterpret_cast<const Frag*>(sB + j * 16 * RB + coff);
#pragma unroll
    for (int i = 0; i < ABUF; ++i) af[i] = *reinterpret_cast<const Frag*>(sA + i * 16 * RB + coff);
  };
  auto mma_all = [&](int stage, int s) __attribute__((always_inline)) {
    if constexpr (!ASTREAM) {
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) acc[i][j] = Tr::mma(bf[j], af[i], acc[i][j]);
    } else {
      const char* sA = smem + stage * STAGE + frag_a + ((4 * s + g) ^ sw) * 16;
#pragma unroll
      for (int j = 0; j < WTN; ++j) acc[0][j] = Tr::mma(bf[j], af[0], acc[0][j]);
      __builtin_amdgcn_sched_barrier(0);
      af[0] = *reinterpret_cast<const Frag*>(sA + 3 * 16 * RB);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 1; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) acc[i][j] = Tr::mma(bf[j], af[i], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < WTN; ++j) acc[3][j] = Tr::mma(bf[j], af[0], acc[3][j]);
    }
  };
  auto phase_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- the stream -----------------------------------------------------------------------------------------------------------
  const int nwork = ntiles * S;
  int i_tile = blockIdx.x, i_kt = 0, i_kt_end = 0;
  int st_c = 0, st_i = 0;
  auto begin_issue = [&]() __attribute__((always_inline)) { setup_issue(i_tile); i_kt = kt_begin(i_tile); i_kt_end = kt_end(i_tile); };
  auto issue_next = [&]() __attribute__((always_inline)) {
    issue(i_kt, st_i);
    st_i ^= 1;
    if (++i_kt == i_kt_end) {
      i_tile += gridDim.x;
      if (i_tile < nwork) begin_issue();
    }
  };
  if (i_tile < nwork) { begin_issue(); issue_next(); }
  wait_vmcnt<0>();

  for (int tile = blockIdx.x; tile < nwork; tile += gridDim.x) {
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI == FYC_EPI_LINEAR) {
      if (p.res_acc) {                                 // the residual rides in the accumulators (epilogue_linear_packed)
        int rm, rn;
        tile_coords(p, remap(tile / S), rm, rn);
        load_residual_acc<T, BM, BN, WGM, WGN>(p, acc, rm, rn, wave, lane);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    phase_barrier();                                 // the previous epilogue is done with its LDS staging; K tile 0 of this tile is visible
    if (grp_y) phase_barrier();                      // Y runs one barrier behind X
    FYC_STAMP(p, wave, lane);
    const int kt_hi = kt_end(tile);
    for (int kt = kt_begin(tile); kt < kt_hi; ++kt) {
      // ---- R(kt, 0): fragments of the first k-step; DMA of the next stream element into the other stage
      read_frags(st_c, 0);
      if (i_tile < nwork) issue_next();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      phase_barrier();
      // ---- M(kt, 0)
      if (p.stagger) __builtin_amdgcn_s_setprio(1);
      mma_all(st_c, 0);
      if (p.stagger) __builtin_amdgcn_s_setprio(0);
      phase_barrier();
      // ---- R(kt, 1)
      read_frags(st_c, 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (grp_y) wait_vmcnt<0>();                    // Y's share of the next stream element
      phase_barrier();
      // ---- M(kt, 1)
      if (p.stagger) __builtin_amdgcn_s_setprio(1);
      mma_all(st_c, 1);
      if (p.stagger) __builtin_amdgcn_s_setprio(0);
      if (!grp_y) wait_vmcnt<0>();                   // X's share of the next stream element
      phase_barrier();
      st_c ^= 1;
    }
    if (!grp_y) phase_barrier();                     // re-align: X waits for Y's last MFMA phase
    FYC_STAMP(p, wave, lane);

    const int t = remap(tile / S);
    int tile_m, tile_n;
    tile_coords(p, t, tile_m, tile_n);
    if (S > 1) {
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
    gemm_epilogue<T, BM, BN, WGM, WGN, EPI, STAGE, MODE, true>(p, acc, tile_m, tile_n, 0, smem + (st_c ^ 1) * STAGE, wave, lane);
    FYC_STAMP(p, wave, lane);
  }
}

template <int BM, int BN, int WGM, int WGN, int MODE, int EPI>
int launch_pp(const GemmP& p, hipStream_t st) {
  constexpr int smem = 2 * (BM + BN) * 128;
  static_assert(smem <= 160 * 1024, "LDS budget");
  auto kern = fyc_gemm_pp_kernel<BM, BN, WGM, WGN, MODE, EPI>;
  int dev = 0;
  (void)hipGetDevice(&dev);
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
  q.rb_slots = (EPI == FYC_EPI_LINEAR && p.rowbias != nullptr && !q.rb_tile) ? rowbias_slots(BM, p.rows_per_batch) : 0;
#ifdef FYC_TRACE
  q.trace = g_fyc_trace;
#endif
  q.res_acc = (EPI == FYC_EPI_LINEAR && p.residual != nullptr && p.ln_stats == nullptr && !(q.splitk > 1)) ? 1 : 0;
  q.stagger = g_fyc_tuning[8] == 1 ? 0 : 1;          // here: s_setprio(1) around the MFMA phases (fyc_set_tuning key 8 = 1: off)
  q.strip = (q.tiles_n > 4 && g_fyc_tuning[4] >= 0) ? (g_fyc_tuning[4] > 0 ? g_fyc_tuning[4] : (q.tiles_n >= 16 ? 8 : 4)) : 0;
  const long long ntiles = (long long)q.tiles_m * q.tiles_n * (q.splitk > 1 ? q.splitk : 1);
  dim3 grid((unsigned)(ntiles < n_cu ? ntiles : n_cu), 1, 1);
  hipLaunchKernelGGL(kern, grid, dim3(512), smem, st, q);
  FYC_CHECK_LAUNCH("fyc_gemm (ping-pong)");
  return 0;
}

// ping-pong tile configurations: id -> (BM, BN, WGM, WGN)
//   21: 256x320, 4x2 waves   22: 128x320, 2x4 waves   23: 256x256, 2x4 waves
constexpr bool pp_cfg(int cfg) { return cfg == 21 || cfg == 22 || cfg == 23; }
template <int MODE, int EPI>
int dispatch_pp(int cfg, const GemmP& p, hipStream_t st) {
  switch (cfg) {
    case 21: return launch_pp<256, 320, 4, 2, MODE, EPI>(p, st);
    case 22: if constexpr (EPI == FYC_EPI_GEGLU) FYC_FAIL(-2, "fyc_gemm: tile config 22 gives a wave an odd number of column blocks: not built for GEGLU");
             else return launch_pp<128, 320, 2, 4, MODE, EPI>(p, st);
    case 23: return launch_pp<256, 256, 2, 4, MODE, EPI>(p, st);
  }
  FYC_FAIL(-2, "fyc_gemm: ping-pong tile config %d not built", cfg);
}

int run_pp_plain(const GemmP& p, int cfg, hipStream_t st);
int run_pp_conv(const GemmP& p, int cfg, hipStream_t st);

}  // namespace fycg
