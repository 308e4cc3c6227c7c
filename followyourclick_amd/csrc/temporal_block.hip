// One temporal self-attention sub-block of the AnimateDiff motion module in ONE kernel (reference motion_module.py:270-283,
// 371-464: norm -> to_q|to_k|to_v (+ positional encoding) -> attention over the frame axis -> to_out -> + residual):
//
//   out = x + Attn_F( LN(x) + pe ) Wo^T + bo            x, out: [(b f p)][C] tokens, attention over f at every pixel p
//
// The unfused schedule (fyc_row_stats, fyc_gemm with the folded LayerNorm, fyc_temporal_attention, fyc_gemm + residual) moves
// the 84-MB token tensor of the 64x64 level through HBM ten times and runs two K = C GEMMs whose epilogues dominate; here a
// workgroup owns 8 pixels x 16 frames = 128 token rows, keeps them in LDS (they are the A operand of the QKV projection and
// the residual), and walks the heads: QKV of one head (128 x 120, K = C) -> LDS, the 16 x 16 attention of each pixel on MFMA
// (one wave per pixel), the head's slice of the output projection accumulated in registers.  One read and one write of x.
//
// LayerNorm is folded as in fyc_gemm: LN(x) W^T = rstd (x (gamma W)^T - mean colsum) + (beta W^T + b); the positional table is a
// per-frame bias pe_f W^T.  Row order inside the tile: row = pixel * 16 + frame, so a 16-row MFMA block is one pixel.
//
// Built for the level where it pays (C = 320, 8 heads of 40, 16 frames, bf16); other shapes keep the unfused schedule.
#include <mutex>

#include "fyc_common.h"

namespace {

struct TBlockP {
  const bf16_t* x; bf16_t* out;
  const bf16_t* w_qkv;   // [H][128][C]: rows 0..39 q, 40..79 k, 80..119 v of the head (gamma folded in), 120..127 zero
  const float* colsum;   // [H][128]
  const float* bias;     // [H][128]
  const float* pe_bias;  // [F][H][128] or null
  const bf16_t* w_out;   // [H][C][48]: Wo[n][h*40 + k] for k < 40, zero for k >= 40
  const float* b_out;    // [C]
  int clips, pixels;
  float scale_log2e, eps;
};

constexpr int C_ = 320, H_ = 8, D_ = 40, F_ = 16, PIX = 8, ROWS = PIX * F_;        // 128 token rows per workgroup
constexpr int XP = C_ * 2 + 16;          // bytes per X row in LDS
constexpr int WP = 64 * 2 + 16;          // bytes per row of a 64-wide K tile of w_qkv
constexpr int QP = 128 * 2 + 16;         // bytes per row of the head's q|k|v (and, in place over q, the attention output)
constexpr int OP = 48 * 2 + 16;          // bytes per row of the head's w_out slice
constexpr int X_BYTES = ROWS * XP, W_BYTES = 2 * 128 * WP, Q_BYTES = ROWS * QP, S_BYTES = ROWS * 8;
constexpr int LDS_BYTES = X_BYTES + W_BYTES + Q_BYTES + S_BYTES;
static_assert(C_ * OP <= W_BYTES, "the w_out slice of a head reuses the w_qkv K-tile buffers");
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

template <typename T> struct BMma;
template <> struct BMma<bf16_t> { __device__ static __forceinline__ f32x4 k32(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); } };
template <> struct BMma<f16_t> { __device__ static __forceinline__ f32x4 k32(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); } };

// T: the 16-bit element type of x / out / the weights (bf16_t or f16_t; TBlockP's pointers are typed bf16_t for both: same arithmetic)
template <typename T>
__global__ void __launch_bounds__(512) temporal_block_kernel(const TBlockP p) {
  typedef typename Pair16<T>::Vec8 Frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sX = smem;
  char* sW = sX + X_BYTES;
  char* sQ = sW + W_BYTES;
  float* sS = reinterpret_cast<float*>(sQ + Q_BYTES);      // [ROWS][2] = {mean, rstd}

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r16 = lane & 15;
  const int wm = wave >> 2, wn = wave & 3;                  // 2 x 4 waves over (rows, columns)
  const int tiles_per_clip = p.pixels / PIX;
  const int clip = blockIdx.x / tiles_per_clip, p0 = (blockIdx.x - clip * tiles_per_clip) * PIX;
  const long long fstride = (long long)p.pixels * C_;      // elements between consecutive frames of a pixel
  const bf16_t* xb = p.x + ((long long)clip * F_ * p.pixels + p0) * C_;
  bf16_t* ob = p.out + ((long long)clip * F_ * p.pixels + p0) * C_;

  // ---- phase 0: the 128 token rows -> LDS (row = pixel * 16 + frame), LayerNorm statistics ---------------------------------
  constexpr int XCH = C_ / 8;                                // 16-byte chunks per row
  for (int c = tid; c < ROWS * XCH; c += 512) {
    const int row = c / XCH, ch = c - row * XCH;
    const int pl = row >> 4, f = row & 15;
    *reinterpret_cast<u32x4*>(sX + row * XP + ch * 16) = *reinterpret_cast<const u32x4*>(xb + f * fstride + pl * C_ + ch * 8);
  }
  __syncthreads();
  {
    const int row = tid >> 2, part = tid & 3;                // 4 threads per row, 10 chunks each
    float s = 0.f, q = 0.f;
    for (int ch = part * 10; ch < part * 10 + 10; ++ch) {
      float v[8];
      load8<T>(reinterpret_cast<const T*>(sX + row * XP + ch * 16), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s += v[e]; q = __builtin_fmaf(v[e], v[e], q); }
    }
    s += __shfl_xor(s, 1); q += __shfl_xor(q, 1);
    s += __shfl_xor(s, 2); q += __shfl_xor(q, 2);
    if (part == 0) {
      const float mu = s * (1.0f / C_);
      sS[2 * row] = mu;
      sS[2 * row + 1] = rsqrtf(fmaxf(q * (1.0f / C_) - mu * mu, 0.f) + p.eps);
    }
  }
  // (the first barrier of the head loop orders sS before its first use)

  f32x4 oacc[4][5];                                          // output projection: rows wm*64 + i*16, columns wn*80 + j*16
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) oacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const Frag zero8 = __builtin_bit_cast(Frag, (u32x4){0u, 0u, 0u, 0u});

  // The 8 x 5 K tiles of w_qkv are one stream: a thread keeps the five tiles of the NEXT head in registers (slot = K tile), loaded
  // while the current head computes - one head of work (several us) covers the L2 latency that a one-tile-ahead prefetch left
  // exposed at every K tile (a K tile is only 16 MFMAs per wave).
  constexpr int KT = C_ / 64;
  u32x4 wreg[KT][2];
  auto wload = [&](int h, int kt, u32x4 (&dst)[2]) {         // 128 rows x 8 chunks = 1024 chunks, 2 per thread
    const bf16_t* wh = p.w_qkv + (long long)h * 128 * C_;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = tid + u * 512, n = c >> 3, ch = c & 7;
      dst[u] = *reinterpret_cast<const u32x4*>(wh + (long long)n * C_ + kt * 64 + ch * 8);
    }
  };
  auto wstore = [&](int buf, const u32x4 (&src)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = tid + u * 512, n = c >> 3, ch = c & 7;
      *reinterpret_cast<u32x4*>(sW + buf * (128 * WP) + n * WP + ch * 16) = src[u];
    }
  };
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) wload(0, kt, wreg[kt]);

  for (int h = 0; h < H_; ++h) {
    // ---- phase A: q|k|v of head h = LN(x) W_h^T, 128 x 128 (120 used), K = C in five 64-wide tiles, double buffered -----------
    const bool more = h + 1 < H_;
    f32x4 qacc[4][2];                                        // rows wm*64 + i*16, columns wn*32 + j*16
#pragma unroll
    for (int i = 0; i < 4; ++i) { qacc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; qacc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    __syncthreads();                                         // previous head is done with sW (phase C) and sQ (phase C reads)
    wstore(0, wreg[0]);
    if (more) wload(h + 1, 0, wreg[0]);
    __syncthreads();
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const char* wb = sW + (kt & 1) * (128 * WP);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        Frag af[4], bf[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const Frag*>(sX + (wm * 64 + i * 16 + r16) * XP + (kt * 64 + ks * 32 + g * 8) * 2);
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const Frag*>(wb + (wn * 32 + j * 16 + r16) * WP + (ks * 32 + g * 8) * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) qacc[i][j] = BMma<T>::k32(bf[j], af[i], qacc[i][j]);
      }
      if (kt + 1 < KT) {
        wstore((kt + 1) & 1, wreg[kt + 1]);
        if (more) wload(h + 1, kt + 1, wreg[kt + 1]);
      }
      __syncthreads();
    }
    // epilogue A: fold the LayerNorm, add bias (+ positional bias of the row's frame), bf16 -> sQ[row][col]
    {
      const float* cs = p.colsum + h * 128;
      const float* bi = p.bias + h * 128;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wm * 64 + i * 16 + r16;              // frame of the row = r16
        const float mu = sS[2 * row], rs = sS[2 * row + 1];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = wn * 32 + j * 16 + g * 4;
          const f32x4 c4 = *reinterpret_cast<const f32x4*>(cs + col), b4 = *reinterpret_cast<const f32x4*>(bi + col);
          f32x4 pe4 = {0.f, 0.f, 0.f, 0.f};
          if (p.pe_bias) pe4 = *reinterpret_cast<const f32x4*>(p.pe_bias + ((long long)r16 * H_ + h) * 128 + col);
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = rs * (qacc[i][j][r] - mu * c4[r]) + b4[r] + pe4[r];
          ElemIO<T>::st4(reinterpret_cast<T*>(sQ + row * QP) + col, v);
        }
      }
    }
    __syncthreads();

    // the head's slice of w_out ([C][48] = C * 6 contiguous 16-byte chunks): requested now, parked in sW after phase B (sW has
    // been idle since the K loop's last barrier; phase B covers the latency and the QKV accumulators are dead)
    u32x4 oreg[4];
    {
      const bf16_t* wo = p.w_out + (long long)h * C_ * 48;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = tid + u * 512;
        oreg[u] = c < C_ * 6 ? *reinterpret_cast<const u32x4*>(wo + c * 8) : (u32x4){0u, 0u, 0u, 0u};
      }
    }

    // ---- phase B: attention over the 16 frames of pixel `wave` (rows wave*16 .. +15), as fyc_temporal_attention ------------
    {
      const char* qrow = sQ + (wave * 16 + r16) * QP;        // this lane's frame row
      Frag qf[2], kf[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int dd = 32 * ks + 8 * g;
        qf[ks] = dd < D_ ? *reinterpret_cast<const Frag*>(qrow + dd * 2) : zero8;
        kf[ks] = dd < D_ ? *reinterpret_cast<const Frag*>(qrow + (D_ + dd) * 2) : zero8;
      }
      // V^T fragments: lane (dv = 16 t + r16, quad g) holds frames 4g .. 4g+3 in k-slots 0..3 (k-slots 4..7: frames 16+, none)
      Frag vf[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int dv = t * 16 + r16;
        unsigned short e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          e[j] = dv < D_ ? *reinterpret_cast<const unsigned short*>(sQ + (wave * 16 + 4 * g + j) * QP + (2 * D_ + dv) * 2) : (unsigned short)0;
        const u32x4 pk = {(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16), 0u, 0u};
        vf[t] = __builtin_bit_cast(Frag, pk);
      }
      f32x4 s = {0.f, 0.f, 0.f, 0.f};                        // S^T: key frame 4g + r, query frame r16
      s = BMma<T>::k32(kf[0], qf[0], s);
      s = BMma<T>::k32(kf[1], qf[1], s);
      float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float e4[4], sum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) { e4[r] = __builtin_amdgcn_exp2f((s[r] - mx) * p.scale_log2e); sum += e4[r]; }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      const float inv = 1.0f / sum;
      const u32x4 ppk = {(unsigned)Pair16<T>::to_bits(e4[0] * inv) | ((unsigned)Pair16<T>::to_bits(e4[1] * inv) << 16),
                         (unsigned)Pair16<T>::to_bits(e4[2] * inv) | ((unsigned)Pair16<T>::to_bits(e4[3] * inv) << 16), 0u, 0u};
      const Frag pf = __builtin_bit_cast(Frag, ppk);
      // O^T = V^T P^T: channel 16 t + 4g + r of query frame r16 -> in place over the q columns of the lane's own row.  All of
      // this wave's reads of its 16 rows (q, k, v above) are complete: the MFMAs that consumed them have been issued in order.
      f32x4 o[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) o[t] = BMma<T>::k32(vf[t], pf, (f32x4){0.f, 0.f, 0.f, 0.f});
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        float v[4] = {o[t][0], o[t][1], o[t][2], o[t][3]};   // channels >= 40 are exact zeros (vf rows of zeros): the K padding
        ElemIO<T>::st4(reinterpret_cast<T*>(sQ + (wave * 16 + r16) * QP) + t * 16 + 4 * g, v);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = tid + u * 512, n = c / 6, ch = c - n * 6;
      if (c < C_ * 6) *reinterpret_cast<u32x4*>(sW + n * OP + ch * 16) = oreg[u];
    }
    __syncthreads();

    // ---- phase C: out += O_h (128 x 48) Wo_h^T (320 x 48); run as two 32-wide k-steps whose lanes beyond 48 carry zeros ------
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kk = ks * 32 + g * 8;
      Frag af[4], bf[5];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = kk < 48 ? *reinterpret_cast<const Frag*>(sQ + (wm * 64 + i * 16 + r16) * QP + kk * 2) : zero8;
#pragma unroll
      for (int j = 0; j < 5; ++j) bf[j] = kk < 48 ? *reinterpret_cast<const Frag*>(sW + (wn * 80 + j * 16 + r16) * OP + kk * 2) : zero8;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) oacc[i][j] = BMma<T>::k32(bf[j], af[i], oacc[i][j]);
    }
    // (the barrier at the top of the next head orders these reads before sW / sQ are overwritten)
  }

  // ---- epilogue: + bias + residual, in place over the token rows in LDS, then 16-byte rows back to HBM ------------------------
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wm * 64 + i * 16 + r16;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int col = wn * 80 + j * 16 + g * 4;
      T* xr = reinterpret_cast<T*>(sX + row * XP) + col;
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.b_out + col);
      float res[4], v[4];
      ElemIO<T>::ld4(xr, res);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = oacc[i][j][r] + b4[r] + res[r];
      ElemIO<T>::st4(xr, v);
    }
  }
  __syncthreads();
  for (int c = tid; c < ROWS * XCH; c += 512) {
    const int row = c / XCH, ch = c - row * XCH;
    const int pl = row >> 4, f = row & 15;
    *reinterpret_cast<u32x4*>(ob + f * fstride + pl * C_ + ch * 8) = *reinterpret_cast<const u32x4*>(sX + row * XP + ch * 16);
  }
}

}  // namespace

extern "C" int fyc_temporal_block_supported(const fyc_temporal_block_args* a) {
  if (!(a != nullptr && (a->dtype == FYC_BF16 || a->dtype == FYC_F16) && a->C == C_ && a->heads == H_ && a->d == D_ && a->frames == F_ && a->pixels > 0 &&
        a->pixels % PIX == 0 && a->clips > 0)) return 0;
  static std::mutex mu;                            // LDS per CU of this process's device, queried once (0: no device answered)
  static int64_t lds_cap = -1;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (lds_cap < 0) {
      int64_t caps[8];
      lds_cap = (fyc_device_caps(caps) == 0) ? caps[1] : 0;
    }
  }
  return (lds_cap > 0 && lds_cap < LDS_BYTES) ? 0 : 1;   // less LDS than the 153 KB tile needs: the engine keeps the unfused schedule
}

extern "C" int64_t fyc_temporal_block_wstream_bytes(void) { return fyc_temporal_block_rr_wstream_bytes(); }

extern "C" int fyc_temporal_block(const fyc_temporal_block_args* a, void* stream) {
  FYC_REQUIRE(a && a->x && a->out && a->b_out && (a->wstream || (a->w_qkv && a->colsum && a->bias && a->w_out)), "fyc_temporal_block: null pointer");
  FYC_REQUIRE(fyc_temporal_block_supported(a), "fyc_temporal_block: built for bf16 / f16, C=320, 8 heads of 40, 16 frames, pixels %% 8 == 0 (got C=%d heads=%d d=%d frames=%d pixels=%d)",
              a->C, a->heads, a->d, a->frames, a->pixels);
  FYC_REQUIRE(a->x != a->out, "fyc_temporal_block: in-place operation is not supported (tiles read rows of every frame)");
  if (a->wstream != nullptr) return fyc_temporal_block_rr_launch(a, stream);      // pre-packed stream: the register-resident kernel
  TBlockP p;
  p.x = (const bf16_t*)a->x; p.out = (bf16_t*)a->out; p.w_qkv = (const bf16_t*)a->w_qkv; p.colsum = a->colsum; p.bias = a->bias;
  p.pe_bias = a->pe_bias; p.w_out = (const bf16_t*)a->w_out; p.b_out = a->b_out; p.clips = a->clips; p.pixels = a->pixels;
  p.scale_log2e = a->scale * 1.44269504088896340736f; p.eps = a->eps;
  {  // dynamic LDS above 64 KB needs the function attribute once per device; one process may drive several GPUs from several threads
    constexpr int kMaxDev = 64;
    static std::mutex mu;
    static bool attr_done[kMaxDev] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    if (dev < 0 || dev >= kMaxDev || !attr_done[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(temporal_block_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(temporal_block_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      if (e != hipSuccess) FYC_FAIL(-3, "fyc_temporal_block: %d bytes of dynamic LDS refused: %s", LDS_BYTES, hipGetErrorString(e));
      if (dev >= 0 && dev < kMaxDev) attr_done[dev] = true;      // only after success: a failed call is retried
    }
  }
  if (a->dtype == FYC_F16) hipLaunchKernelGGL(temporal_block_kernel<f16_t>, dim3((unsigned)(a->clips * (a->pixels / PIX))), dim3(512), LDS_BYTES, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(temporal_block_kernel<bf16_t>, dim3((unsigned)(a->clips * (a->pixels / PIX))), dim3(512), LDS_BYTES, (hipStream_t)stream, p);
  FYC_CHECK_LAUNCH("fyc_temporal_block");
  return 0;
}
