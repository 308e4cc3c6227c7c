// Register-staged variant of the MFMA GEMM / implicit-GEMM conv kernel (same operands, LDS image, MFMA orientation and
// epilogue as gemm_kernel.h).
//
// Why: the DMA ring of gemm_kernel.h keeps at most (NS-1) K tiles in flight and NS*(BM+BN)*128 B must fit the 160 KB LDS,
// i.e. one tile (57-74 KB) for the 320-wide configurations.  A K tile needs every one of its ~4600 16-byte loads, so its
// latency is the L2-MISS latency even at 90 % hit rate; with one tile in flight the measured fill rate is the
// latency-bound 45-60 GB/s per CU (profiles/r01_lds_fill_microbench.txt: 40-53 GB/s/CU beyond L2, 140 GB/s/CU L2-resident
// with twice the bytes in flight).  The register file (512 KB / CU) is the only larger on-chip store: here K tiles travel
// global -> VGPR (two tiles in flight, issued two iterations before they are needed) -> LDS (ds_write_b128, one tile ahead
// of the MFMAs) -> MFMA, which doubles the latency the pipeline tolerates at the same LDS footprint.
//
// Stream element e (a K tile of some output tile) uses LDS stage e&1 and register set e&1, so the loop is unrolled by two
// and every register-array index is a compile-time constant (no scratch).
#pragma once
#include <type_traits>

#include "gemm_kernel.h"

namespace fycg {

// The staging loads are inline asm on purpose: the compiler's automatic s_waitcnt insertion cannot count through this loop
// (the epilogue's global stores share vmcnt with loads on gfx9-family parts and complete out of order with respect to them,
// so every compiler-tracked load wait in the loop degrades to vmcnt(0) = "wait for the tile issued last as well").  Asm
// loads are invisible to that pass; the kernel issues exactly LOADS of them per step and waits with explicit counts.
// Contract: a value produced here may only be consumed after a wait_vmcnt<> that covers it.
__device__ __forceinline__ f32x4 gload16(const void* ptr) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
  return v;
}

template <typename T, int BM, int BN, int WGM, int WGN, int MODE, int EPI>
__global__ void __launch_bounds__(WGM* WGN * 64) fyc_gemm_staged_kernel(const GemmP p) {
  typedef Mma<T> Tr;
  typedef typename Tr::Frag Frag;
  constexpr int RB = 128;
  constexpr int NT = WGM * WGN * 64;
  constexpr int CH = 16 / (int)sizeof(T);
  constexpr int CPR = RB / 16;
  constexpr int BK = CPR * CH;
  constexpr int KSTEPS = RB / 64;
  constexpr int A_IT = BM * CPR / NT, B_IT = BN * CPR / NT;
  constexpr int LOADS = A_IT + B_IT;
  constexpr int WTM = BM / WGM / 16, WTN = BN / WGN / 16;
  constexpr int A_BYTES = BM * RB, STAGE = (BM + BN) * RB;
  static_assert(A_IT * NT == BM * CPR && B_IT * NT == BN * CPR, "tile/threads mismatch");
  static_assert(LOADS <= 31, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int ntiles = p.tiles_m * p.tiles_n;
  auto remap = [&](int t) {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = t & 7, idx = t >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  };
  const long long bz = blockIdx.z;
  const T* __restrict__ A = reinterpret_cast<const T*>(p.a) + bz * p.stride_a;
  const T* __restrict__ W = reinterpret_cast<const T*>(p.w) + bz * p.stride_w;
  const T* __restrict__ A2 = reinterpret_cast<const T*>(p.a2);
  const T* zero = reinterpret_cast<const T*>(p.zero);

  // ---- loader descriptors of the output tile whose K tiles are being fetched ---------------------------------------
  int a_koff[A_IT], b_koff[B_IT];
  long long a_row[A_IT], a_row2[A_IT], b_row[B_IT];
  int a_pix[A_IT], a_iy0[A_IT], a_ix0[A_IT];
  int tap = 0, c0 = 0;
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int idx = tid + it * NT, row = idx / CPR, slot = idx % CPR;
    a_koff[it] = ((slot ^ swz_key<RB>(row)) * CH);
  }
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int idx = tid + it * NT, row = idx / CPR, slot = idx % CPR;
    b_koff[it] = ((slot ^ swz_key<RB>(row)) * CH);
  }
  auto setup_fetch = [&](int tile) {                // tile >= ntiles: every source becomes the zero page (uniform issue count)
    const bool live = tile < ntiles;
    const int t = remap(live ? tile : 0);
    int tile_m, tile_n;
    tile_coords(p, t, tile_m, tile_n);
    tap = 0; c0 = 0;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int row = (tid + it * NT) / CPR;
      const int m = tile_m * BM + row;
      if (MODE == FYC_GEMM_PLAIN) {
        a_row[it] = (live && m < p.M) ? (long long)m * p.lda : -1;
        a_row2[it] = (long long)m * p.lda2;
        a_pix[it] = a_iy0[it] = a_ix0[it] = 0;
      } else {
        const int hw = p.Hout * p.Wout;
        const int fr = m / hw, rem = m - fr * hw, oy = rem / p.Wout, ox = rem - oy * p.Wout;
        a_row[it] = (live && m < p.M) ? 0 : -1;
        a_row2[it] = 0;
        a_pix[it] = fr * p.Hin * p.Win;
        a_iy0[it] = oy * p.conv_stride - p.conv_pad;
        a_ix0[it] = ox * p.conv_stride - p.conv_pad;
      }
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int row = (tid + it * NT) / CPR;
      const int n = tile_n * BN + row;
      b_row[it] = (live && n < p.N) ? (long long)n * p.ldw : -1;
    }
  };
  const int KT = (p.K + BK - 1) / BK;
  auto src_a = [&](int it, int k0) -> const T* {
    if (MODE == FYC_GEMM_PLAIN) {
      const int k = k0 + a_koff[it];
      if (a_row[it] < 0 || k >= p.K) return zero;
      if (A2 != nullptr && k >= p.k_split) return A2 + a_row2[it] + (k - p.k_split);
      return A + a_row[it] + k;
    } else {
      const int ky = tap / 3, kx = tap - 3 * ky;
      const int iy = a_iy0[it] + ky, ix = a_ix0[it] + kx;
      if (MODE == FYC_GEMM_CONV3X3) {
        const bool ok = a_row[it] >= 0 && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
        return ok ? A + (long long)(a_pix[it] + iy * p.Win + ix) * p.Cin + c0 + a_koff[it] : zero;
      } else {
        const bool ok = a_row[it] >= 0 && (unsigned)iy < (unsigned)p.Hout && (unsigned)ix < (unsigned)p.Wout;
        int sy, sx;
        if (p.up_exact2) { sy = iy >> 1; sx = ix >> 1; }
        else {
          sy = min((int)floorf((float)iy * p.up_sh), p.Hin - 1);
          sx = min((int)floorf((float)ix * p.up_sw), p.Win - 1);
        }
        return ok ? A + (long long)(a_pix[it] + sy * p.Win + sx) * p.Cin + c0 + a_koff[it] : zero;
      }
    }
  };
  auto src_b = [&](int it, int k0) -> const T* {
    const int k = k0 + b_koff[it];
    return (b_row[it] >= 0 && k < p.K) ? W + b_row[it] + k : zero;
  };

  // ---- fetch side of the stream: element (f_tile, f_kt) goes global -> register set --------------------------------
  f32x4 rbuf[2][LOADS];
  int f_tile = blockIdx.x, f_kt = 0;
  auto fetch = [&](auto RS) {                       // returns via `f_tile < ntiles` before the call whether anything was issued
    constexpr int rs = decltype(RS)::value;
    const int k0 = f_kt * BK;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) rbuf[rs][it] = gload16(src_a(it, k0));
#pragma unroll
    for (int it = 0; it < B_IT; ++it) rbuf[rs][A_IT + it] = gload16(src_b(it, k0));
    if (MODE != FYC_GEMM_PLAIN) {
      if (++tap == 9) { tap = 0; c0 += BK; }
    }
    if (++f_kt == KT) {
      f_kt = 0;
      f_tile += gridDim.x;
      setup_fetch(f_tile);
    }
  };
  auto lstore = [&](auto RS) {                      // register set -> LDS stage of the same parity (the DMA kernel's LDS image)
    constexpr int rs = decltype(RS)::value;
    char* sA = smem + rs * STAGE;
    char* sB = sA + A_BYTES;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) *reinterpret_cast<f32x4*>(sA + (it * NT + tid) * 16) = rbuf[rs][it];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) *reinterpret_cast<f32x4*>(sB + (it * NT + tid) * 16) = rbuf[rs][A_IT + it];
  };

  f32x4 acc[WTM][WTN];
  const int g = lane >> 4, r16 = lane & 15;
  const int sw = swz_key<RB>(r16);
  auto compute = [&](int stage) {
    const char* sA = smem + stage * STAGE + (wm * WTM * 16 + r16) * RB;
    const char* sB = smem + stage * STAGE + A_BYTES + (wn * WTN * 16 + r16) * RB;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      const int coff = ((4 * s + g) ^ sw) * 16;
      Frag af[WTM], bf[WTN];
#pragma unroll
      for (int i = 0; i < WTM; ++i) af[i] = *reinterpret_cast<const Frag*>(sA + i * 16 * RB + coff);
#pragma unroll
      for (int j = 0; j < WTN; ++j) bf[j] = *reinterpret_cast<const Frag*>(sB + j * 16 * RB + coff);
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) acc[i][j] = Tr::mma(bf[j], af[i], acc[i][j]);
    }
  };

  // ---- pipeline ------------------------------------------------------------------------------------------------------
  // invariant at the top of the step for element e (parity P): LDS[P] holds e (ds_writes issued), register set 1-P holds
  // e+1 and register set P holds e+2 (both possibly still in flight, e+2 issued after e+1).
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  int c_tile = blockIdx.x, c_kt = 0;                // compute side of the stream
  if (c_tile >= ntiles) return;
  // Every step issues exactly LOADS loads (past the end of the stream they read the zero page), so "the tile issued two
  // steps ago has landed" is always s_waitcnt vmcnt(LOADS): loads return in order and only the newest tile may be pending.
  setup_fetch(f_tile);
  fetch(I0{});                                       // e0 -> R0
  fetch(I1{});                                       // e1 -> R1
  wait_vmcnt<LOADS>();
  lstore(I0{});                                      // e0 -> LDS0
  fetch(I0{});                                       // e2 -> R0

  auto step = [&](auto PP) -> bool {
    constexpr int P = decltype(PP)::value;
    typedef std::integral_constant<int, 1 - P> Q;
    if (c_kt == 0) {
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's ds_writes of element e are done
    __builtin_amdgcn_s_barrier();                        // everyone's are; everyone finished reading LDS[1-P]
    wait_vmcnt<LOADS>();                                 // e+1 has landed in register set 1-P; e+2 may stay in flight
    lstore(Q{});
    fetch(Q{});                                          // e+3 -> the register set just drained
    compute(P);
    if (++c_kt == KT) {
      const int t = remap(c_tile);
      int tile_m, tile_n;
    tile_coords(p, t, tile_m, tile_n);
      gemm_epilogue<T, BM, BN, WGM, WGN, EPI, STAGE>(p, acc, tile_m, tile_n, bz, smem + P * STAGE, wave, lane);
      c_kt = 0;
      c_tile += gridDim.x;
      if (c_tile >= ntiles) return false;
    }
    return true;
  };
  for (;;) {
    if (!step(I0{})) break;
    if (!step(I1{})) break;
  }
}

template <typename T, int BM, int BN, int WGM, int WGN, int MODE, int EPI>
int launch_staged(const GemmP& p, int batch, hipStream_t st) {
  constexpr int smem = 2 * (BM + BN) * 128;
  static_assert(smem <= 160 * 1024, "LDS budget");
  auto kern = fyc_gemm_staged_kernel<T, BM, BN, WGM, WGN, MODE, EPI>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_done = true;
  }
  GemmP q = p;
  q.tiles_m = (p.M + BM - 1) / BM;
  q.tiles_n = (p.N + BN - 1) / BN;
  q.strip = (q.tiles_n > 4 && g_fyc_tuning[4] >= 0) ? (g_fyc_tuning[4] > 0 ? g_fyc_tuning[4] : (q.tiles_n >= 16 ? 8 : 4)) : 0;   // measured: profiles/r01_gemm_strip_order.txt
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n_cu = pr.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  int occ = (160 * 1024) / smem;
  if (occ < 1) occ = 1;
  long long resident = (long long)n_cu * occ / (batch > 0 ? batch : 1);
  if (resident < n_cu) resident = n_cu;
  const long long ntiles = (long long)q.tiles_m * q.tiles_n;
  dim3 grid((unsigned)(ntiles < resident ? ntiles : resident), 1, batch);
  hipLaunchKernelGGL(kern, grid, dim3(WGM * WGN * 64), smem, st, q);
  FYC_CHECK_LAUNCH("fyc_gemm(staged)");
  return 0;
}

int run_bf16_staged(const GemmP& p, int batch, int cfg, hipStream_t st);   // tile configs 12..: gemm_bf16_staged.hip

}  // namespace fycg
