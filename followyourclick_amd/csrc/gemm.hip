// GEMM / implicit-GEMM 3x3 convolution on MFMA for gfx950 (CDNA4).
//
//   out[m][n] = (sum_k A[m][k] * W[n][k] + bias[n] + rowbias[m/rpb][n] + residual[m][n]) * out_scale
//
// Data layout: activations channels-last, so a 3x3 convolution is a GEMM whose A rows are
// gathered from shifted pixels: K is ordered (ky, kx, ci) and every 128-byte K-tile lies inside one
// filter tap, i.e. an A-tile row is one contiguous 128-B channel slice of one input pixel (or the
// zero page at the image border / beyond M).  Nearest-2x upsampling (Upsample3D) is folded into
// that gather: no 4x tensor is ever written.
//
// Tile: BM x BN x 128 B of K per step, WGM x WGN wave64s, each wave a (BM/WGM) x (BN/WGN) sub-tile
// of 16x16 MFMA blocks.  Operands are staged global -> LDS by direct DMA (global_load_lds, 16 B per
// lane): the LDS image is lane-linear, so the bank-conflict XOR swizzle is applied to the *source*
// chunk each lane fetches and again on the ds_read_b128 address (same involution on both sides).
// Two LDS stages: the DMA of tile k+1 is in flight while tile k feeds the matrix cores.
//
// MFMA orientation: acc = mfma(Wfrag, Afrag) computes the transposed block D[n][m], so every lane
// ends up with 4 *consecutive output channels* of one row: the epilogue (bias, time-embedding row,
// residual, GEGLU gate, head split) is vectorised over channels and stores 8/16 B per lane.
//
// f32 parity mode uses the same kernel with v_mfma_f32_16x16x4_f32 (exact f32, 1/16 rate).
#include "fyc_common.h"

namespace {

struct GemmP {
  const char* a; const char* w; const float* bias; const float* rowbias; const char* residual; char* out;
  char* seg_out[3]; int seg_transposed[3]; int seg_ld[3];
  int M, N, K, lda, ldw, ldo, ldr, ldrb;
  long long stride_a, stride_w, stride_o;
  int mode, epilogue;
  int Hout, Wout, Hin, Win, Cin, conv_stride;
  int rows_per_batch, seg_cols, heads, tokens, head_dim;
  float out_scale;
  int tiles_m, tiles_n;
  const char* zero;
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  typedef bf16x8 Frag;
  __device__ static __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
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

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <typename T, int BM, int BN, int WGM, int WGN, int MODE, bool GLDS>
__global__ void __launch_bounds__(WGM* WGN * 64) fyc_gemm_kernel(const GemmP p) {
  typedef Mma<T> Tr;
  typedef typename Tr::Frag Frag;
  constexpr int NT = WGM * WGN * 64;
  constexpr int CH = 16 / (int)sizeof(T);  // elements per 16-B chunk
  constexpr int BK = 8 * CH;               // elements per K tile (one 128-B LDS row)
  constexpr int A_IT = BM * 8 / NT, B_IT = BN * 8 / NT;
  constexpr int WTM = BM / WGM / 16, WTN = BN / WGN / 16;
  constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
  static_assert(A_IT * NT == BM * 8 && B_IT * NT == BN * 8, "tile/threads mismatch");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  // XCD-aware tile order: block b runs on XCD b%8; give each XCD a contiguous run of tiles
  // (n fastest) so the A panel of a row-block is fetched into one L2 only.  Bijective for any count.
  int t = blockIdx.x;
  {
    const int nt = p.tiles_m * p.tiles_n, q = nt >> 3, r = nt & 7, xcd = t & 7, idx = t >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_n = t % p.tiles_n, tile_m = t / p.tiles_n;
  const long long bz = blockIdx.z;
  const T* __restrict__ A = reinterpret_cast<const T*>(p.a) + bz * p.stride_a;
  const T* __restrict__ W = reinterpret_cast<const T*>(p.w) + bz * p.stride_w;
  const T* zero = reinterpret_cast<const T*>(p.zero);

  // ---- per-thread loader descriptors ------------------------------------------------------
  int a_koff[A_IT];
  long long a_row[A_IT];   // PLAIN: m*lda, or -1 when the row is outside M
  int a_pix[A_IT], a_iy0[A_IT], a_ix0[A_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int idx = tid + it * NT, row = idx >> 3, slot = idx & 7;
    a_koff[it] = ((slot ^ (row & 7)) * CH);
    const int m = tile_m * BM + row;
    if (MODE == FYC_GEMM_PLAIN) {
      a_row[it] = (m < p.M) ? (long long)m * p.lda : -1;
      a_pix[it] = a_iy0[it] = a_ix0[it] = 0;
    } else {
      const int hw = p.Hout * p.Wout;
      const int fr = m / hw, rem = m - fr * hw, oy = rem / p.Wout, ox = rem - oy * p.Wout;
      a_row[it] = (m < p.M) ? 0 : -1;
      a_pix[it] = fr * p.Hin * p.Win;
      a_iy0[it] = oy * p.conv_stride - 1;
      a_ix0[it] = ox * p.conv_stride - 1;
    }
  }
  int b_koff[B_IT];
  long long b_row[B_IT];
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int idx = tid + it * NT, row = idx >> 3, slot = idx & 7;
    b_koff[it] = ((slot ^ (row & 7)) * CH);
    const int n = tile_n * BN + row;
    b_row[it] = (n < p.N) ? (long long)n * p.ldw : -1;
  }

  const int KT = (p.K + BK - 1) / BK;
  int tap = 0, c0 = 0;  // conv: filter tap and channel offset of the K tile being *issued*

  auto src_a = [&](int it, int k0) -> const T* {
    if (MODE == FYC_GEMM_PLAIN) {
      const int k = k0 + a_koff[it];
      return (a_row[it] >= 0 && k < p.K) ? A + a_row[it] + k : zero;
    } else {
      const int ky = tap / 3, kx = tap - 3 * ky;
      const int iy = a_iy0[it] + ky, ix = a_ix0[it] + kx;
      if (MODE == FYC_GEMM_CONV3X3) {
        const bool ok = a_row[it] >= 0 && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
        return ok ? A + (long long)(a_pix[it] + iy * p.Win + ix) * p.Cin + c0 + a_koff[it] : zero;
      } else {  // nearest-2x upsampled input: virtual size (2*Hin, 2*Win)
        const bool ok = a_row[it] >= 0 && (unsigned)iy < (unsigned)(2 * p.Hin) && (unsigned)ix < (unsigned)(2 * p.Win);
        return ok ? A + (long long)(a_pix[it] + (iy >> 1) * p.Win + (ix >> 1)) * p.Cin + c0 + a_koff[it] : zero;
      }
    }
  };
  auto src_b = [&](int it, int k0) -> const T* {
    const int k = k0 + b_koff[it];
    return (b_row[it] >= 0 && k < p.K) ? W + b_row[it] + k : zero;
  };
  auto advance_tap = [&]() {
    if (MODE != FYC_GEMM_PLAIN) {
      c0 += BK;
      if (c0 >= p.Cin) { c0 = 0; ++tap; }
    }
  };

  f32x4 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int sw = lane & 7, g = lane >> 4, r16 = lane & 15;
  auto compute = [&](int stage) {
    const char* sA = smem + stage * STAGE + (wm * WTM * 16 + r16) * 128;
    const char* sB = smem + stage * STAGE + A_BYTES + (wn * WTN * 16 + r16) * 128;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int coff = ((4 * s + g) ^ sw) * 16;
      Frag af[WTM], bf[WTN];
#pragma unroll
      for (int i = 0; i < WTM; ++i) af[i] = *reinterpret_cast<const Frag*>(sA + i * 16 * 128 + coff);
#pragma unroll
      for (int j = 0; j < WTN; ++j) bf[j] = *reinterpret_cast<const Frag*>(sB + j * 16 * 128 + coff);
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) acc[i][j] = Tr::mma(bf[j], af[i], acc[i][j]);
    }
  };

  if (GLDS) {
    auto issue = [&](int kt, int stage) {
      char* sA = smem + stage * STAGE;
      char* sB = sA + A_BYTES;
      const int k0 = kt * BK;
#pragma unroll
      for (int it = 0; it < A_IT; ++it) glds16(src_a(it, k0), sA + (it * NT + wave * 64) * 16);
#pragma unroll
      for (int it = 0; it < B_IT; ++it) glds16(src_b(it, k0), sB + (it * NT + wave * 64) * 16);
      advance_tap();
    };
    issue(0, 0);
    for (int kt = 0; kt < KT; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (kt + 1 < KT) issue(kt + 1, (kt + 1) & 1);
      compute(kt & 1);
    }
  } else {
    u32x4 ra[A_IT], rb[B_IT];
    auto fetch = [&](int kt) {
      const int k0 = kt * BK;
#pragma unroll
      for (int it = 0; it < A_IT; ++it) ra[it] = *reinterpret_cast<const u32x4*>(src_a(it, k0));
#pragma unroll
      for (int it = 0; it < B_IT; ++it) rb[it] = *reinterpret_cast<const u32x4*>(src_b(it, k0));
      advance_tap();
    };
    auto commit = [&](int stage) {
      char* sA = smem + stage * STAGE;
      char* sB = sA + A_BYTES;
#pragma unroll
      for (int it = 0; it < A_IT; ++it) *reinterpret_cast<u32x4*>(sA + (tid + it * NT) * 16) = ra[it];
#pragma unroll
      for (int it = 0; it < B_IT; ++it) *reinterpret_cast<u32x4*>(sB + (tid + it * NT) * 16) = rb[it];
    };
    fetch(0);
    commit(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
      if (kt + 1 < KT) fetch(kt + 1);
      compute(kt & 1);
      if (kt + 1 < KT) commit((kt + 1) & 1);
      __syncthreads();
    }
  }

  // ---- epilogue: lane holds out[m][n0 .. n0+3] per (i, j) ---------------------------------
  T* O = reinterpret_cast<T*>(p.out) + bz * p.stride_o;
  const T* R = reinterpret_cast<const T*>(p.residual);
#pragma unroll
  for (int i = 0; i < WTM; ++i) {
    const int m = tile_m * BM + (wm * WTM + i) * 16 + r16;
    if (m >= p.M) continue;
    const float* rb = p.rowbias ? p.rowbias + (long long)(m / p.rows_per_batch) * p.ldrb : nullptr;
    if (p.epilogue == FYC_EPI_GEGLU) {
      // packed columns: [32b, 32b+16) = value channels 16b.., [32b+16, 32b+32) = their gates
#pragma unroll
      for (int j = 0; j + 1 < WTN; j += 2) {
        const int n = tile_n * BN + (wn * WTN + j) * 16 + g * 4;  // value column (packed index)
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float h = acc[i][j][r], gt = acc[i][j + 1][r];
          if (p.bias) { h += p.bias[n + r]; gt += p.bias[n + 16 + r]; }
          v[r] = h * (0.5f * gt * (1.0f + erff(gt * 0.70710678118654752440f)));
        }
        const int oc = (n >> 5) * 16 + (n & 15);
        ElemIO<T>::st4(O + (long long)m * p.ldo + oc, v);
      }
    } else {
#pragma unroll
      for (int j = 0; j < WTN; ++j) {
        const int n = tile_n * BN + (wn * WTN + j) * 16 + g * 4;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
        const bool full = (n + 3 < p.N);
        if (p.bias) {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (full || n + r < p.N) v[r] += p.bias[n + r];
        }
        if (rb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (full || n + r < p.N) v[r] += rb[n + r];
        }
        if (p.epilogue == FYC_EPI_LINEAR) {
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
    }
  }
}

template <typename T, int BM, int BN, int WGM, int WGN, int MODE, bool GLDS>
int launch(const GemmP& p, int batch, hipStream_t st) {
  constexpr int smem = 2 * (BM + BN) * 128;
  auto kern = fyc_gemm_kernel<T, BM, BN, WGM, WGN, MODE, GLDS>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_done = true;
  }
  GemmP q = p;
  q.tiles_m = (p.M + BM - 1) / BM;
  q.tiles_n = (p.N + BN - 1) / BN;
  dim3 grid(q.tiles_m * q.tiles_n, 1, batch);
  hipLaunchKernelGGL(kern, grid, dim3(WGM * WGN * 64), smem, st, q);
  FYC_CHECK_LAUNCH("fyc_gemm");
  return 0;
}

template <typename T, int MODE, bool GLDS>
int dispatch_tile(const GemmP& p, int batch, hipStream_t st) {
  // N that is a multiple of 128 (or large) takes the 128x128 tile; 320/960/64.. take 128x64.
  if (p.N % 128 == 0) return launch<T, 128, 128, 2, 2, MODE, GLDS>(p, batch, st);
  return launch<T, 128, 64, 2, 2, MODE, GLDS>(p, batch, st);
}

template <typename T, bool GLDS>
int dispatch_mode(const GemmP& p, int batch, hipStream_t st) {
  switch (p.mode) {
    case FYC_GEMM_PLAIN: return dispatch_tile<T, FYC_GEMM_PLAIN, GLDS>(p, batch, st);
    case FYC_GEMM_CONV3X3: return dispatch_tile<T, FYC_GEMM_CONV3X3, GLDS>(p, batch, st);
    case FYC_GEMM_CONV3X3_UP2: return dispatch_tile<T, FYC_GEMM_CONV3X3_UP2, GLDS>(p, batch, st);
  }
  FYC_FAIL(-2, "fyc_gemm: bad mode %d", p.mode);
}

}  // namespace

extern "C" int fyc_gemm(const fyc_gemm_args* a, void* stream) {
  FYC_REQUIRE(a != nullptr, "fyc_gemm: null args");
  FYC_REQUIRE(g_fyc_zero_page != nullptr, "fyc_gemm: fyc_init() not called");
  FYC_REQUIRE(a->dtype == FYC_F32 || a->dtype == FYC_BF16, "fyc_gemm: bad dtype %d", a->dtype);
  const int es = a->dtype == FYC_BF16 ? 2 : 4, ch = 16 / es;
  FYC_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "fyc_gemm: empty problem M=%d N=%d K=%d", a->M, a->N, a->K);
  FYC_REQUIRE(a->K % ch == 0, "fyc_gemm: K=%d must be a multiple of %d", a->K, ch);
  FYC_REQUIRE(a->ldw % ch == 0 && a->stride_w % ch == 0, "fyc_gemm: ldw/stride_w must keep 16-B alignment");
  FYC_REQUIRE(((uintptr_t)a->a % 16) == 0 && ((uintptr_t)a->w % 16) == 0, "fyc_gemm: operands must be 16-B aligned");
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.a = (const char*)a->a; p.w = (const char*)a->w; p.bias = a->bias; p.rowbias = a->rowbias;
  p.residual = (const char*)a->residual; p.out = (char*)a->out;
  p.M = a->M; p.N = a->N; p.K = a->K; p.lda = a->lda; p.ldw = a->ldw; p.ldo = a->ldo; p.ldr = a->ldr;
  p.stride_a = a->stride_a; p.stride_w = a->stride_w; p.stride_o = a->stride_o;
  p.mode = a->mode; p.epilogue = a->epilogue;
  p.Hout = a->Hout; p.Wout = a->Wout; p.Hin = a->Hin; p.Win = a->Win; p.Cin = a->Cin; p.conv_stride = a->conv_stride;
  p.rows_per_batch = a->rows_per_batch > 0 ? a->rows_per_batch : 1;
  p.ldrb = a->ldrb > 0 ? a->ldrb : a->N;
  p.out_scale = a->out_scale;
  p.zero = (const char*)g_fyc_zero_page;
  const int batch = a->batch > 0 ? a->batch : 1;
  if (a->mode == FYC_GEMM_PLAIN) {
    FYC_REQUIRE(a->lda % ch == 0 && a->stride_a % ch == 0, "fyc_gemm: lda/stride_a must keep 16-B alignment");
  } else {
    const int bk = 8 * ch;
    FYC_REQUIRE(a->mode == FYC_GEMM_CONV3X3 || a->mode == FYC_GEMM_CONV3X3_UP2, "fyc_gemm: bad mode %d", a->mode);
    FYC_REQUIRE(a->Cin > 0 && a->Cin % bk == 0, "fyc_gemm conv: Cin=%d must be a multiple of %d", a->Cin, bk);
    FYC_REQUIRE(a->K == 9 * a->Cin, "fyc_gemm conv: K=%d != 9*Cin", a->K);
    FYC_REQUIRE(a->Hout > 0 && a->Wout > 0 && a->Hin > 0 && a->Win > 0, "fyc_gemm conv: bad spatial dims");
    FYC_REQUIRE(a->M % (a->Hout * a->Wout) == 0, "fyc_gemm conv: M=%d not a multiple of Hout*Wout", a->M);
    FYC_REQUIRE(batch == 1, "fyc_gemm conv: batch must be 1");
    if (a->mode == FYC_GEMM_CONV3X3) {
      FYC_REQUIRE(a->conv_stride == 1 || a->conv_stride == 2, "fyc_gemm conv: stride %d", a->conv_stride);
      FYC_REQUIRE(a->Hout == (a->Hin + 2 - 3) / a->conv_stride + 1 && a->Wout == (a->Win + 2 - 3) / a->conv_stride + 1,
                  "fyc_gemm conv: output size mismatch");
    } else {
      p.conv_stride = 1;
      FYC_REQUIRE(a->Hout == 2 * a->Hin && a->Wout == 2 * a->Win, "fyc_gemm upconv: output must be 2x input");
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
  hipStream_t st = (hipStream_t)stream;
  const bool glds = g_fyc_gemm_staging == 0;
  if (a->dtype == FYC_BF16) return glds ? dispatch_mode<bf16_t, true>(p, batch, st) : dispatch_mode<bf16_t, false>(p, batch, st);
  return glds ? dispatch_mode<float, true>(p, batch, st) : dispatch_mode<float, false>(p, batch, st);
}
