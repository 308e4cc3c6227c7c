// Row-panel linear layer for the short-K projections of the transformer blocks (reference animatediff/models/attention.py:270
// proj_in, diffusers/models/attention.py:600-623 to_out; motion_module.py:191 proj_in, :270-283 to_out):
//
//   out = [GroupNorm](x) W^T + bias (+ residual)          x [rows][K], W [N][K], K in {320, 640}, N in {320, 640}
//
// fyc_gemm runs these K <= 640 layers at 0.11-0.17 of the MFMA peak: 5-10 K tiles of MFMA work per output tile against an epilogue
// of the same length, with the matrix pipe idle during the epilogue and the operand fill exposed at every tile seam.  Here, as in
// fyc_ff_block (ff_block.hip has the measurements behind each choice):
//   * a workgroup = 4 wave64 (one per SIMD, 512 registers each) owns 128 rows, a wave 32 of them for all columns; the 32 x K
//     activations live in registers as MFMA operands (80 / 160 VGPRs) - read once, in fragment shape, no LDS traffic;
//   * W arrives pre-packed in MFMA fragment order (engine/weights.py::pack_panel_linear) as one stream of 40-KiB stages
//     (2 k-steps x 20 column blocks = one 320-column pass per K/64 stages), by LDS-DMA issued from inline asm (counted lgkmcnt
//     waits for the fragment reads) and dealt out between the first MFMAs of the previous stage, 2-deep ring;
//   * GroupNorm of the INPUT is applied to the operand registers (per-(sample, channel) scale / shift from the producer's
//     channel sums - the same f64 sums fyc_gn_apply_cs uses): the normalised tensor of a proj_in never exists in HBM, one full
//     read + write pass less per transformer / motion module (the north_star's "fused GroupNorm" for the 1x1 consumers);
//   * per 320-column pass: residual tile by DMA into LDS beside the ring, bias + residual + accumulators in f32, one rounding
//     to bf16, rows leave as 16-byte stores over whole 640-byte segments.
// Built for bf16, rows % 128 == 0, (K, N) in {320, 640}^2; other shapes keep fyc_gemm.  AGPR accumulators (see _build.py).
#include <mutex>

#include "fyc_common.h"

namespace {

constexpr int ROWS = 128, NT = 256, PN = 320;     // rows per workgroup, threads, columns per pass
constexpr int NB = PN / 16;                        // 20 column blocks per pass
constexpr int PIECE = 1024;
constexpr int NPIECE = 2 * NB;                     // 40 pieces per stage: 2 k-steps x 20 column blocks
constexpr int STAGE_BYTES = NPIECE * PIECE;        // 40960
constexpr int RING_BYTES = 2 * STAGE_BYTES;        // 81920
constexpr int TILE_BYTES = ROWS * PN * 2;          // 81920: residual / output tile of a pass, beside the ring
constexpr int LDS_BYTES = RING_BYTES + TILE_BYTES; // 163840 = all of a CU's LDS
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

struct PLP {
  const bf16_t* x; const bf16_t* res; bf16_t* out;
  const char* ws;                 // packed weight stream: (N / 320) passes x (K / 64) stages x 40 KiB
  const float* bias;              // [N] or null
  const double* gn_cs;            // [samples * gn_stat_samples][K][2] channel {sum, sum sq} of x, or null: no GroupNorm
  const float* gn_gamma; const float* gn_beta;
  int gn_rows_per_sample, gn_stat_samples, gn_groups;
  float gn_eps;
  int N;
};

// 1 KiB global -> LDS by DMA from inline asm (see ff_block.hip: the builtin form degrades every lgkmcnt wait to lgkmcnt(0))
__device__ __forceinline__ void dma16(const char* gbase, unsigned voff, unsigned lds_dst) {          // wave-uniform base + 32-bit lane offset
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
__device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
template <typename Frag> __device__ __forceinline__ Frag frag(const char* sl, int piece) { return *reinterpret_cast<const Frag*>(sl + piece * PIECE); }

// KS = K / 32 MFMA k-steps (10 or 20); NPASS = N / 320 column passes (1 or 2), unrolled: the accumulators of a pass are dead
// before the next one starts, and a runtime loop around this much unrolled code made the register allocator spill (ff_block.hip)
// E16: the 16-bit element type of x / residual / out / the weight stream (bf16_t or f16_t; PLP's pointers are typed bf16_t for both)
template <typename E16, int KS, int NPASS>
__global__ void __launch_bounds__(NT) panel_linear_kernel(const PLP p) {
  typedef typename Pair16<E16>::Vec8 Frag;
  constexpr int K = KS * 32, SPP = KS / 2;          // stages per pass
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const tile = smem + RING_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r16 = lane & 15;
  const unsigned lane16 = (unsigned)lane * 16u;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const long long row0 = (long long)blockIdx.x * ROWS;

  // piece n (of 10) of this wave of stage T -> slot T & 1
  auto dma_piece = [&](int T, int n) {
    const int q = wave + 4 * n;
    dma16(p.ws + (long long)T * STAGE_BYTES + q * PIECE, lane16, lds0 + (T & 1) * STAGE_BYTES + q * PIECE);
  };

  Frag xa[2][KS];
  {
    const bf16_t* xr = p.x + (row0 + wave * 32 + r16) * K + g * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s = 0; s < KS; ++s) xa[i][s] = *reinterpret_cast<const Frag*>(xr + (long long)i * 16 * K + s * 32);
  }
#pragma unroll
  for (int n = 0; n < 10; ++n) dma_piece(0, n);

  if (p.gn_cs != nullptr) {
    // GroupNorm of the input on the operand registers.  Per-channel {scale, shift} of the tile's sample -> LDS table (the tile
    // region is idle here): thread c folds the statistics samples (frames) of the norm's sample, the cpg threads of a group meet in LDS.
    double* dsum = reinterpret_cast<double*>(tile);                  // [K][2]
    float* tab = reinterpret_cast<float*>(tile + K * 16);            // [K][2] = {scale, shift}
    const int sample = (int)(row0 / p.gn_rows_per_sample), cpg = K / p.gn_groups;
    for (int c = tid; c < K; c += NT) {
      double s = 0.0, q = 0.0;
      for (int f = 0; f < p.gn_stat_samples; ++f) {
        const double* src = p.gn_cs + (((long long)sample * p.gn_stat_samples + f) * K + c) * 2;
        s += src[0]; q += src[1];
      }
      dsum[2 * c] = s; dsum[2 * c + 1] = q;
    }
    __syncthreads();
    for (int c = tid; c < K; c += NT) {
      const int c0 = (c / cpg) * cpg;
      double s = 0.0, q = 0.0;
      for (int j = 0; j < cpg; ++j) { s += dsum[2 * (c0 + j)]; q += dsum[2 * (c0 + j) + 1]; }
      const double inv_cnt = 1.0 / ((double)p.gn_rows_per_sample * cpg);
      const double mean = s * inv_cnt;
      double var = q * inv_cnt - mean * mean;
      var = var > 0.0 ? var : 0.0;
      const float rstd = (float)(1.0 / sqrt(var + (double)p.gn_eps));
      const float sc = rstd * p.gn_gamma[c];
      tab[2 * c] = sc;
      tab[2 * c + 1] = p.gn_beta[c] - (float)mean * sc;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      float sc[8], sh[8];
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(tab + 2 * (32 * s + 8 * g + e));      // {scale, shift} of channels k, k + 1
        sc[e] = t[0]; sh[e] = t[1]; sc[e + 1] = t[2]; sh[e + 1] = t[3];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        u32x4 t = __builtin_bit_cast(u32x4, xa[i][s]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          t[e] = Pair16<E16>::pack(__builtin_fmaf(Pair16<E16>::lo(t[e]), sc[2 * e], sh[2 * e]),
                                 __builtin_fmaf(Pair16<E16>::hi(t[e]), sc[2 * e + 1], sh[2 * e + 1]));
        xa[i][s] = __builtin_bit_cast(Frag, t);
      }
    }
    __syncthreads();                                                   // the table is dead: the tile region may take the residual DMA
  }

  f32x4 oacc[2][NB];
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) oacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < SPP; ++st) {
      const int T = pass * SPP + st;
      dma_landed_barrier();                                            // stage T landed; slot (T+1)&1 is free
      if (st == SPP - 1 && p.res != nullptr) {                         // last stage of the pass: its residual tile (128 x 640-byte row segments)
#pragma unroll
        for (int n = 0; n < TILE_BYTES / PIECE / 4; ++n) {
          const int c = (wave + 4 * n) * 64 + lane, row = c / 40, cg = c - row * 40;
          dma16v(p.res + (row0 + row) * p.N + pass * PN + cg * 8, lds0 + RING_BYTES + (wave + 4 * n) * PIECE);
        }
      }
      const char* sl = smem + (T & 1) * STAGE_BYTES + lane16;
#pragma unroll
      for (int u = 0; u < 2 * NB; ++u) {
        const int sk = u / NB, j = u % NB;
        const Frag wf = frag<Frag>(sl, u);
#pragma unroll
        for (int i = 0; i < 2; ++i) oacc[i][j] = mfma(wf, xa[i][2 * st + sk], oacc[i][j]);
        if (u < 10 && T + 1 < NPASS * SPP) dma_piece(T + 1, u);
      }
    }
    // ---- epilogue of the pass -----------------------------------------------------------------------------------------------------
    dma_landed_barrier();                                              // residual tile landed (and the next pass's first stage)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        E16* a = reinterpret_cast<E16*>(tile + (wave * 32 + i * 16 + r16) * (PN * 2)) + j * 16 + g * 4;
        f32x4 bo = {0.f, 0.f, 0.f, 0.f};
        if (p.bias != nullptr) bo = *reinterpret_cast<const f32x4*>(p.bias + pass * PN + j * 16 + g * 4);
        float rr[4] = {0.f, 0.f, 0.f, 0.f}, v[4];
        if (p.res != nullptr) ElemIO<E16>::ld4(a, rr);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = oacc[i][j][r] + bo[r] + rr[r];
        ElemIO<E16>::st4(a, v);
      }
    __syncthreads();
    {
      const int cg = tid % 40, rsl = tid / 40;                         // 240 threads: column group of 8 channels x row slice
      if (rsl < 6) {
        char* dst = reinterpret_cast<char*>(p.out + row0 * p.N + pass * PN) + cg * 16;
        const char* src = tile + cg * 16;
#pragma unroll 2
        for (int row = rsl; row < ROWS; row += 6)
          *reinterpret_cast<u32x4*>(dst + (long long)row * p.N * 2) = *reinterpret_cast<const u32x4*>(src + row * (PN * 2));
      }
    }
    if (pass + 1 < NPASS) __syncthreads();                            // the tile region is reused by the next pass
  }
}

template <typename T, int KS, int NPASS>
int launch(const PLP& p, int rows, hipStream_t st) {
  auto kern = panel_linear_kernel<T, KS, NPASS>;
  {
    constexpr int kMaxDev = 64;
    static std::mutex mu;
    static bool attr_done[kMaxDev] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    if (dev < 0 || dev >= kMaxDev || !attr_done[dev]) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      if (e != hipSuccess) FYC_FAIL(-3, "fyc_panel_linear: %d bytes of dynamic LDS refused: %s", LDS_BYTES, hipGetErrorString(e));
      if (dev >= 0 && dev < kMaxDev) attr_done[dev] = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(rows / ROWS)), dim3(NT), LDS_BYTES, st, p);
  FYC_CHECK_LAUNCH("fyc_panel_linear");
  return 0;
}

}  // namespace

extern "C" int64_t fyc_panel_linear_wstream_bytes(int32_t N, int32_t K) {
  if (N <= 0 || K <= 0 || N % PN != 0 || K % 64 != 0) return 0;
  return (int64_t)(N / PN) * (K / 64) * STAGE_BYTES;
}

extern "C" int fyc_panel_linear_supported(const fyc_panel_linear_args* a) {
  if (a == nullptr || (a->dtype != FYC_BF16 && a->dtype != FYC_F16) || a->rows <= 0 || a->rows % ROWS != 0) return 0;
  if (!((a->K == 320 || a->K == 640) && (a->N == 320 || a->N == 640))) return 0;
  if (a->gn_cs != nullptr && (a->gn_rows_per_sample <= 0 || a->gn_rows_per_sample % ROWS != 0 || a->rows % a->gn_rows_per_sample != 0 ||
                              a->gn_groups <= 0 || a->K % a->gn_groups != 0 || a->gn_stat_samples <= 0)) return 0;
  static std::mutex mu;
  static int64_t lds_cap = -1;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (lds_cap < 0) {
      int64_t caps[8];
      lds_cap = (fyc_device_caps(caps) == 0) ? caps[1] : 0;
    }
  }
  return (lds_cap > 0 && lds_cap < LDS_BYTES) ? 0 : 1;
}

extern "C" int fyc_panel_linear(const fyc_panel_linear_args* a, void* stream) {
  FYC_REQUIRE(a && a->x && a->out && a->wstream, "fyc_panel_linear: null pointer");
  FYC_REQUIRE(fyc_panel_linear_supported(a), "fyc_panel_linear: built for bf16 / f16, rows %% 128 == 0, K and N in {320, 640}, GroupNorm samples of whole 128-row tiles (got rows=%d K=%d N=%d gn_rows_per_sample=%d)",
              a->rows, a->K, a->N, a->gn_rows_per_sample);
  FYC_REQUIRE(a->gn_cs == nullptr || (a->gn_gamma != nullptr && a->gn_beta != nullptr), "fyc_panel_linear: gn_cs needs gn_gamma / gn_beta");
  FYC_REQUIRE(a->x != a->out && a->residual != a->x, "fyc_panel_linear: out must not alias x (residual may alias out)");
  FYC_REQUIRE(((uintptr_t)a->x % 16) == 0 && ((uintptr_t)a->out % 16) == 0 && ((uintptr_t)a->wstream % 16) == 0 && ((uintptr_t)a->bias % 16) == 0 &&
              ((uintptr_t)a->residual % 16) == 0 && ((uintptr_t)a->gn_cs % 16) == 0, "fyc_panel_linear: operands must be 16-byte aligned");
  PLP p;
  p.x = (const bf16_t*)a->x; p.res = (const bf16_t*)a->residual; p.out = (bf16_t*)a->out; p.ws = (const char*)a->wstream; p.bias = a->bias;
  p.gn_cs = a->gn_cs; p.gn_gamma = a->gn_gamma; p.gn_beta = a->gn_beta; p.gn_rows_per_sample = a->gn_rows_per_sample;
  p.gn_stat_samples = a->gn_stat_samples; p.gn_groups = a->gn_groups; p.gn_eps = a->gn_eps; p.N = a->N;
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == FYC_F16) {
    if (a->K == 320) return a->N == 320 ? launch<f16_t, 10, 1>(p, a->rows, st) : launch<f16_t, 10, 2>(p, a->rows, st);
    return a->N == 320 ? launch<f16_t, 20, 1>(p, a->rows, st) : launch<f16_t, 20, 2>(p, a->rows, st);
  }
  if (a->K == 320) return a->N == 320 ? launch<bf16_t, 10, 1>(p, a->rows, st) : launch<bf16_t, 10, 2>(p, a->rows, st);
  return a->N == 320 ? launch<bf16_t, 20, 1>(p, a->rows, st) : launch<bf16_t, 20, 2>(p, a->rows, st);
}
