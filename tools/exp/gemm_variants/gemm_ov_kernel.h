// bf16 GEMM / implicit-GEMM convolution with the EPILOGUE UNDER THE NEXT TILE'S K LOOP (gfx950), tile 128x320, 8 waves.
//
// Why (profiles/r04_gemm_phase_trace.txt): the kernels of gemm_kernel.h run one workgroup per CU and every CU reaches its epilogue
// at about the same time.  An epilogue moves the output tile (and reads nothing else any more) at 4-5 bytes per cycle and CU - all 256
// CUs bursting into HBM together - while the matrix pipes idle; the K loops in between leave HBM almost idle.  With the pack-first
// epilogue a 128x320 tile still spends 12 000-18 000 cycles there, against 12 000 (K = 320) ... 108 000 (K = 2880) in its K loop.
// This kernel removes the epilogue as a phase:
//
//   * when a tile's K loop ends, its 80 accumulators are turned into final values and PACKED to bf16 pairs (40 registers, `pk`) at
//     the top of the next tile's first K-tile iteration - bias, LayerNorm fold, row bias and row statistics come out of small LDS
//     arrays that were filled by DMA during the tile's LAST K tile (no load instruction with a register destination, so no
//     compiler-inserted vmcnt(0) anywhere near the DMA stream);
//   * the packed tile then leaves through the wave's private staging slice ONE 16-row block per K-tile iteration of the next tile
//     (5 ds_write_b64, 3 ds_read_b128 + 3 global stores of 16 bytes per lane, placed between the fragment reads of the iteration
//     and its first MFMAs, where the wave would otherwise wait for LDS); the output statistics of the consuming GroupNorm accumulate
//     in registers over the four blocks, are reduced through the staging slice into a per-tile LDS accumulator and are published in
//     the fifth iteration - every hand-off between waves rides on the barrier the K loop has anyway;
//   * the residual tile is loaded into the fresh accumulators right behind the pack (20 loads in flight together), so the sum stays
//     one f32 accumulation with a single rounding;
//   * stores are issued BEFORE the iteration's DMA batch: vmcnt counts stores too, and with nothing but older operations in
//     front of it `s_waitcnt vmcnt(0)` at the top of the next iteration means what it meant before.
//   Output traffic is thereby spread over the K loop of the following tile instead of bursting between two loops.
//
// The K loop reads all 9 fragments of a k-step, then issues its 20 MFMAs from registers (fenced: hipcc would sink the A reads between
// the MFMAs); the store step and the DMA issue of an iteration sit between the first k-step's reads and its MFMAs, i.e. in the
// shadow of one of the two LDS round trips per K tile.  (A second fragment set for k-step 1 - 36 more registers next to 80
// accumulators, 40 packed words and 16 statistics registers - spilled inside the loop.)  Loader, LDS image, tile order and the
// wave-role stagger of the DMA issue are those of gemm_kernel.h.
//
// Built for: bf16, LINEAR epilogue without activation, K a multiple of 64 with at least 5 K tiles (4 store steps + the statistics
// step must fit under a K loop), statistics per whole tile (cs_rows % 128 == 0), no split-K / row statistics / batch.  fyc_gemm()
// falls back to the one-phase kernels for everything else.
#pragma once
#include "../../../followyourclick_amd/csrc/gemm_kernel.h"

namespace fycg {

template <int BM, int BN, int WGM, int WGN, int MODE>
struct OvLayout {
  static constexpr int RB = 128;
  static constexpr int WTM = BM / WGM / 16, WTN = BN / WGN / 16;
  static constexpr int STAGE = (BM + BN) * RB;
  static constexpr int PITCH = WTN * 32 + 16;                          // bf16 staging row of a wave
  static constexpr int SLICE = 16 * PITCH > 2048 ? 16 * PITCH : 2048;  // per wave: 16 rows, or the 64 x 8 floats of the statistics reduction
  static constexpr int CP = ((BN * 4 + 1023) / 1024) * 1024;           // one per-column constants array, whole 1-KiB DMA pieces
  static constexpr int LNP = ((BM * 8 + 1023) / 1024) * 1024;          // {mean, rstd} per row
  static constexpr int OFF_STG = 2 * STAGE;
  static constexpr int OFF_BIAS = OFF_STG + 8 * SLICE;
  static constexpr int OFF_CSUM = OFF_BIAS + CP;
  static constexpr int OFF_RB = OFF_CSUM + CP;
  static constexpr int OFF_LN = OFF_RB + RB_SLOTS * CP;
  static constexpr int OFF_CACC = OFF_LN + LNP;
  static constexpr int TOTAL = OFF_CACC + WGM * BN * 2 * 4;     // one accumulator array per wave row: plain stores, added in order
};

template <int BM, int BN, int WGM, int WGN, int MODE>
__global__ void __launch_bounds__(512) fyc_gemm_ov_kernel(const GemmP p) {
  typedef bf16_t T;
  typedef Mma<T> Tr;
  typedef typename Tr::Frag Frag;
  typedef OvLayout<BM, BN, WGM, WGN, MODE> L;
  constexpr int NT = 512, RB = 128, CH = 8, CPR = 8, BK = 64;
  constexpr int A_IT = BM * CPR / NT, B_IT = BN * CPR / NT;
  constexpr int WTM = L::WTM, WTN = L::WTN;
  constexpr int A_BYTES = BM * RB, STAGE = L::STAGE;
  static_assert(WGM * WGN == 8 && A_IT * NT == BM * CPR && B_IT * NT == BN * CPR, "8 waves; whole DMA pieces per thread");
  static_assert(WTM * WTN * 4 <= 96, "the packed copy of the previous tile must fit beside the accumulators");
  static_assert(L::TOTAL <= 160 * 1024, "LDS budget");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int ntiles = p.tiles_m * p.tiles_n;
  auto remap = [&](int t) {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = t & 7, idx = t >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  };
  const T* __restrict__ A = reinterpret_cast<const T*>(p.a);
  const T* __restrict__ W = reinterpret_cast<const T*>(p.w);
  const T* __restrict__ A2 = reinterpret_cast<const T*>(p.a2);
  const T* zero = reinterpret_cast<const T*>(p.zero);

  // ---- loader (as gemm_kernel.h) --------------------------------------------------------------------------------------------
  const int lrow = tid / CPR;
  const int koff = ((tid % CPR) ^ (lrow & 7)) * CH;
  constexpr int ROWS_IT = NT / CPR;
  int i_tm = 0, i_tn = 0;
  unsigned a_base = 0, a2_base = 0, b_base = 0;
  constexpr bool C3 = (MODE == FYC_GEMM_CONV3X3), UP = (MODE == FYC_GEMM_CONV3X3_UP2);
  int a_pos[C3 ? A_IT : 1], a_msk[C3 ? A_IT : 1];
  int a_pix[UP ? A_IT : 1], a_yx[UP ? A_IT : 1];
  int tap = 0, c0 = 0;
  const int KT = p.K / BK;
  auto setup_issue = [&](int work) __attribute__((always_inline)) {
    tile_coords(p, remap(work), i_tm, i_tn);
    tap = 0; c0 = 0;
    b_base = (unsigned)(i_tn * BN + lrow) * (unsigned)p.ldw + koff;
    if (MODE == FYC_GEMM_PLAIN) {
      a_base = (unsigned)(i_tm * BM + lrow) * (unsigned)p.lda + koff;
      a2_base = (unsigned)(i_tm * BM + lrow) * (unsigned)p.lda2 + koff - p.k_split;
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
  auto src_a = [&](int it, int k0) __attribute__((always_inline)) -> const T* {
    if (MODE == FYC_GEMM_PLAIN) {
      const int m = i_tm * BM + lrow + it * ROWS_IT;
      if (m >= p.M) return zero;
      if (A2 != nullptr && k0 >= p.k_split) return A2 + (size_t)(a2_base + (unsigned)(it * ROWS_IT) * (unsigned)p.lda2 + k0);
      return A + (size_t)(a_base + (unsigned)(it * ROWS_IT) * (unsigned)p.lda + k0);
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
    char* sA = smem + stage * STAGE;
    char* sB = sA + A_BYTES;
    const int k0 = kt * BK;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) glds16(src_a(it, k0), sA + (it * NT + wave * 64) * 16);
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int n = i_tn * BN + lrow + it * ROWS_IT;
      glds16(n < p.N ? W + (size_t)(b_base + (unsigned)(it * ROWS_IT) * (unsigned)p.ldw + k0) : zero, sB + (it * NT + wave * 64) * 16);
    }
    if (MODE != FYC_GEMM_PLAIN) {
      if (++tap == 9) { tap = 0; c0 += BK; }
    }
  };

  // ---- per-column / per-row constants of a tile: DMA pieces into the LDS arrays, issued in the tile's last K-tile iteration -------
  // piece q = 64 lanes x 16 bytes of one array; wave (q % 8) issues it.  Lanes past the array (or past N / M) read the zero page.
  const int n_rb = p.rowbias == nullptr ? 0 : (p.rb_tile ? 1 : p.rb_slots);
  auto issue_consts = [&](int tm, int tn) __attribute__((always_inline)) {
    constexpr int PC = L::CP / 1024;                       // pieces per column array
    const int f = lane * 4;                                // this lane's first float inside a piece
    int q = 0;
    auto col_piece = [&](const float* src, int off, int piece) __attribute__((always_inline)) {
      if ((q++ & 7) != wave) return;
      const int c = piece * 256 + f, n = tn * BN + c;
      const float* ptr = (src != nullptr && c < BN && n < p.N) ? src + n : reinterpret_cast<const float*>(p.zero);
      glds16(ptr, smem + off + piece * 1024);
    };
#pragma unroll
    for (int pc = 0; pc < PC; ++pc) col_piece(p.bias, L::OFF_BIAS, pc);
    if (p.ln_stats != nullptr) {
#pragma unroll
      for (int pc = 0; pc < PC; ++pc) col_piece(p.ln_colsum, L::OFF_CSUM, pc);
      constexpr int PL = L::LNP / 1024;
#pragma unroll
      for (int pc = 0; pc < PL; ++pc) {
        if ((q++ & 7) != wave) continue;
        const int row = pc * 128 + lane * 2, m = tm * BM + row;       // two rows of {mean, rstd} per lane (M is even: host)
        glds16((row < BM && m < p.M) ? p.ln_stats + 2ll * m : reinterpret_cast<const float*>(p.zero), smem + L::OFF_LN + pc * 1024);
      }
    }
    const int b0 = (tm * BM) / p.rows_per_batch, nb = (p.M + p.rows_per_batch - 1) / p.rows_per_batch;
    for (int sl = 0; sl < n_rb; ++sl) {
      const float* row = (b0 + sl < nb) ? p.rowbias + (long long)(b0 + sl) * p.ldrb : nullptr;
#pragma unroll
      for (int pc = 0; pc < PC; ++pc) col_piece(row, L::OFF_RB + sl * L::CP, pc);
    }
  };

  f32x4 acc[WTM][WTN];
  u32x2 pk[WTM][WTN];                      // the previous tile, final values as bf16 pairs, waiting for its store steps
  float cs8[8], cq8[8];                    // column statistics of the previous tile (this lane's 8 output columns)
  FYC_STAMP_DECL;
  const int g = lane >> 4, r16 = lane & 15;
  const int sw = r16 & 7;
  const int frag_a = (wm * WTM * 16 + r16) * RB, frag_b = A_BYTES + (wn * WTN * 16 + r16) * RB;
  Frag fa0[WTM], fb0[WTN];
  char* stg = smem + L::OFF_STG + wave * L::SLICE;
  float* cacc = reinterpret_cast<float*>(smem + L::OFF_CACC);
  const int nl_w0 = wn * WTN * 16;
  constexpr int PITCH = L::PITCH, SCPR = WTN * 2, RPP = 64 / SCPR, NQ = (16 + RPP - 1) / RPP;
  const int s_lrow = lane / SCPR, s_lch = lane - s_lrow * SCPR;
  const bool s_lact = s_lrow < RPP;
  const bool do_cs = p.chan_parts != nullptr;

  // final values of the finished tile (tm, tn), packed; see epilogue_linear_packed (gemm_kernel.h) for the arithmetic
  auto pack_tile = [&](int tm) __attribute__((always_inline)) {
    const float* bias = reinterpret_cast<const float*>(smem + L::OFF_BIAS);
    const float* csum = reinterpret_cast<const float*>(smem + L::OFF_CSUM);
    const float* lnst = reinterpret_cast<const float*>(smem + L::OFF_LN);
    const bool ln = p.ln_stats != nullptr;
    float mu[WTM], rs[WTM];
    int rbo[WTM];
#pragma unroll
    for (int i = 0; i < WTM; ++i) {
      const int row = (wm * WTM + i) * 16;
      mu[i] = 0.f; rs[i] = 1.f;
      if (ln) { const f32x2 ms = *reinterpret_cast<const f32x2*>(lnst + 2 * (row + r16)); mu[i] = ms.x; rs[i] = ms.y; }
      rbo[i] = (n_rb > 1) ? ((tm * BM + row) / p.rows_per_batch - (tm * BM) / p.rows_per_batch) * (L::CP / 4) : 0;
    }
    const float* rbc = reinterpret_cast<const float*>(smem + L::OFF_RB);
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
      const int nl = nl_w0 + j * 16 + g * 4;
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + nl);
      f32x4 s4 = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (ln) s4 = *reinterpret_cast<const f32x4*>(csum + nl);
#pragma unroll
      for (int i = 0; i < WTM; ++i) {
        f32x4 v = acc[i][j];
        if (ln) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = rs[i] * (v[r] - mu[i] * s4[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += b4[r];
        if (n_rb > 0) {
          const f32x4 r4 = *reinterpret_cast<const f32x4*>(rbc + rbo[i] + nl);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += r4[r];
        }
        unsigned lo = pack_bf16x2(v[0] * p.out_scale, v[1] * p.out_scale), hi = pack_bf16x2(v[2] * p.out_scale, v[3] * p.out_scale);
        asm volatile("" : "+v"(lo), "+v"(hi));
        pk[i][j] = (u32x2){lo, hi};
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) cs8[e] = cq8[e] = 0.f;
  };
  // rows [16 i, 16 i + 16) of the packed tile (tm, tn): staging slice -> 16-byte row segments -> global, statistics on the way
  auto store_block = [&](int i, int tm, int tn) __attribute__((always_inline)) {
    T* O = reinterpret_cast<T*>(p.out);
    const int n_lane = tn * BN + nl_w0 + s_lch * 8;
#pragma unroll
    for (int ii = 0; ii < WTM; ++ii) {
      if (ii != i) continue;                              // `i` is a runtime step counter; pk must be indexed by constants
#pragma unroll
      for (int j = 0; j < WTN; ++j) *reinterpret_cast<u32x2*>(stg + r16 * PITCH + j * 32 + g * 8) = pk[ii][j];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int row = q * RPP + s_lrow;
      const int m = tm * BM + (wm * WTM + i) * 16 + row;
      if (s_lact && row < 16 && m < p.M && n_lane < p.N) {
        const u32x4 v4 = *reinterpret_cast<const u32x4*>(stg + row * PITCH + s_lch * 16);
        *reinterpret_cast<u32x4*>(O + (long long)m * p.ldo + n_lane) = v4;
        if (do_cs) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x0 = __uint_as_float(v4[e] << 16), x1 = __uint_as_float(v4[e] & 0xffff0000u);
            cs8[2 * e] += x0; cq8[2 * e] = __builtin_fmaf(x0, x0, cq8[2 * e]);
            cs8[2 * e + 1] += x1; cq8[2 * e + 1] = __builtin_fmaf(x1, x1, cq8[2 * e + 1]);
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  // the wave's column sums -> the tile's LDS accumulator (lanes park their 8 sums in the idle staging slice, one lane per column adds
  // the RPP rows of its column and stores the total in the array of its wave row: no atomics, see gemm_kernel.h::stats_flush)
  auto reduce_stats = [&]() __attribute__((always_inline)) {
    float* red = reinterpret_cast<float*>(stg);
    constexpr int NCOL = SCPR * 8, ROWS_LIVE = RPP < 16 ? RPP : 16;
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      if (s_lact) {
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
        cacc[(wm * BN + nl_w0 + c) * 2 + ph] = t;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  };
  // every wave has added its sums (a barrier ago): plain stores of the tile's {sum, sum sq} per column, accumulator cleared for the next tile
  auto publish_stats = [&](int tm, int tn) __attribute__((always_inline)) {
    for (int c = tid; c < BN; c += NT) {
      const int n = tn * BN + c;
      float2 v = *reinterpret_cast<const float2*>(cacc + 2 * c);
#pragma unroll
      for (int w = 1; w < WGM; ++w) { const float2 u = *reinterpret_cast<const float2*>(cacc + (w * BN + c) * 2); v.x += u.x; v.y += u.y; }
      if (n < p.N) *reinterpret_cast<float2*>(p.chan_parts + ((long long)tm * p.N + n) * 2) = v;
    }
  };

  auto read_frags = [&](int stage, int s, Frag (&fa)[WTM], Frag (&fb)[WTN]) __attribute__((always_inline)) {
    const char* sA = smem + stage * STAGE + frag_a;
    const char* sB = smem + stage * STAGE + frag_b;
    const int coff = ((4 * s + g) ^ sw) * 16;
#pragma unroll
    for (int j = 0; j < WTN; ++j) fb[j] = *reinterpret_cast<const Frag*>(sB + j * 16 * RB + coff);
#pragma unroll
    for (int i = 0; i < WTM; ++i) fa[i] = *reinterpret_cast<const Frag*>(sA + i * 16 * RB + coff);
  };

  // ---- the stream -----------------------------------------------------------------------------------------------------------
  for (int c = tid; c < WGM * BN * 2; c += NT) cacc[c] = 0.f;
  int i_tile = blockIdx.x, i_kt = 0;
  int st_c = 0, st_i = 0;
  auto issue_next = [&]() __attribute__((always_inline)) {
    issue(i_kt, st_i);
    st_i ^= 1;
    if (++i_kt == KT) {
      i_kt = 0;
      i_tile += gridDim.x;
      if (i_tile < ntiles) setup_issue(i_tile);
    }
  };
  if (i_tile < ntiles) { setup_issue(i_tile); issue_next(); }
  int pend = 0, pend_tm = 0, pend_tn = 0;  // 0: nothing pending; 1: pack; 2 .. WTM + 1: store row block pend - 2; WTM + 2: publish statistics
  const bool late = wave >= 4;             // the SIMD partner group issues its DMA in the middle of the iteration (wave-role stagger)

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int cm, cn;
    tile_coords(p, remap(tile), cm, cn);
    for (int kt = 0; kt < KT; ++kt) {
      wait_vmcnt<0>();                               // this wave's share of the stage about to be read (and its stores of the last step)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt == 0) {
        if (pend == 1) { pack_tile(pend_tm); pend = 2; }
        if (p.res_acc) load_residual_acc<T, BM, BN, WGM, WGN>(p, acc, cm, cn, wave, lane);
        else {
#pragma unroll
          for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        FYC_STAMP(p, wave, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
      read_frags(st_c, 0, fa0, fb0);
      __builtin_amdgcn_sched_barrier(0);
      // one step of the previous tile's epilogue, in the shadow of the fragment reads; its stores precede this iteration's DMA batch
      if (pend >= 2) {
        if (pend < 2 + WTM) {
          store_block(pend - 2, pend_tm, pend_tn);
          ++pend;
          if (pend == 2 + WTM) { if (do_cs) reduce_stats(); else pend = 0; }
        } else {
          publish_stats(pend_tm, pend_tn);
          pend = 0;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!late) {
        if (i_tile < ntiles) issue_next();
        if (kt == KT - 1) issue_consts(cm, cn);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) acc[i][j] = Tr::mma(fb0[j], fa0[i], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
      read_frags(st_c, 1, fa0, fb0);
      __builtin_amdgcn_sched_barrier(0);
      if (late) {
        if (i_tile < ntiles) issue_next();
        if (kt == KT - 1) issue_consts(cm, cn);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) acc[i][j] = Tr::mma(fb0[j], fa0[i], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
      st_c ^= 1;
    }
    FYC_STAMP(p, wave, lane);
    FYC_STAMP(p, wave, lane);                        // (no epilogue phase: the trace format wants three stamps per tile)
    pend = 1; pend_tm = cm; pend_tn = cn;
  }
  // ---- drain: the last tile of the stream has no next K loop to hide under ---------------------------------------------------
  wait_vmcnt<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (pend == 1) {
    pack_tile(pend_tm);
#pragma unroll
    for (int i = 0; i < WTM; ++i) store_block(i, pend_tm, pend_tn);
    if (do_cs) {
      reduce_stats();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      publish_stats(pend_tm, pend_tn);
    }
  }
}

template <int BM, int BN, int WGM, int WGN, int MODE>
int launch_ov(const GemmP& p, hipStream_t st) {
  typedef OvLayout<BM, BN, WGM, WGN, MODE> L;
  constexpr int smem = L::TOTAL;
  auto kern = fyc_gemm_ov_kernel<BM, BN, WGM, WGN, MODE>;
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
  q.rb_tile = (p.rowbias != nullptr && p.rows_per_batch % BM == 0) ? 1 : 0;
  q.rb_slots = (p.rowbias != nullptr && !q.rb_tile) ? rowbias_slots(BM, p.rows_per_batch) : 0;
  q.res_acc = p.residual != nullptr ? 1 : 0;
#ifdef FYC_TRACE
  q.trace = g_fyc_trace;
#endif
  q.strip = (q.tiles_n > 4 && g_fyc_tuning[4] >= 0) ? (g_fyc_tuning[4] > 0 ? g_fyc_tuning[4] : (q.tiles_n >= 16 ? 8 : 4)) : 0;
  const long long ntiles = (long long)q.tiles_m * q.tiles_n;
  dim3 grid((unsigned)(ntiles < n_cu ? ntiles : n_cu), 1, 1);
  hipLaunchKernelGGL(kern, grid, dim3(512), smem, st, q);
  FYC_CHECK_LAUNCH("fyc_gemm (overlapped epilogue)");
  return 0;
}

// overlapped-epilogue tile configuration: 31 = 128x320, 2x4 waves
constexpr bool ov_cfg(int cfg) { return cfg == 31; }
int run_ov(const GemmP& p, int cfg, hipStream_t st);

}  // namespace fycg
