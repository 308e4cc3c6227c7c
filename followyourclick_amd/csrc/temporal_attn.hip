// Temporal self-attention core of the AnimateDiff motion module (attention over the FRAME axis at
// every pixel; motion_module.py:371-464).  HBM-bound: arithmetic intensity ~F/2 FLOP/B.
//
// Input is the token-major output of the fused to_q|to_k|to_v GEMM, qkv[(b f p)][3C]; output is
// token-major o[(b f p)][C].  The reference's two physical transposes '(b f) d c <-> (b d) f c' are
// pure index arithmetic here: one wave64 owns one (clip b, pixel p, head h) task and gathers the
// F rows of that pixel (row stride P*3C) straight into MFMA operand registers.
//
// 16-bit path (v_mfma_f32_16x16x32_bf16 / _f16: template parameter T), F <= 32:
//   S^T = K Q^T (rows = key frames, cols = query frames), full softmax in registers (all keys of a
//   query are in 4 lanes x 4..8 regs), O^T = V^T P^T with the keys as the 32 MFMA k-slots.
// f32 path: scalar reference-precision kernel (parity mode only).
#include "fyc_common.h"

namespace {

struct TAttnP {
  const void* qkv; void* o;
  int clips, frames, pixels, heads, d;
  float scale;
  const char* zero;
};

template <typename T> struct TMma;
template <> struct TMma<bf16_t> { __device__ static __forceinline__ f32x4 k32(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); } };
template <> struct TMma<f16_t> { __device__ static __forceinline__ f32x4 k32(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); } };

template <typename T, int DP, int DVT, int NFT>
__global__ void __launch_bounds__(256) tattn_bf16_kernel(const TAttnP p) {
  typedef typename Pair16<T>::Vec8 Frag;
  constexpr int KS = DP / 32;
  const int lane = threadIdx.x & 63, g = lane >> 4, r16 = lane & 15;
  const long long task = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long ntask = (long long)p.clips * p.pixels * p.heads;
  if (task >= ntask) return;
  const int h = (int)(task % p.heads);
  const long long bp = task / p.heads;
  const int pix = (int)(bp % p.pixels), b = (int)(bp / p.pixels);
  const int F = p.frames, C = p.heads * p.d, ld = 3 * C;
  const T* base = reinterpret_cast<const T*>(p.qkv) + ((long long)b * F * p.pixels + pix) * ld + h * p.d;
  const long long fstride = (long long)p.pixels * ld;  // elements between consecutive frames of a pixel
  const T* zero = reinterpret_cast<const T*>(p.zero);

  // Q / K fragments: lane (frame = 16*tile + r16, quad g) holds 8 consecutive head channels
  Frag qf[NFT][KS], kf[NFT][KS];
#pragma unroll
  for (int ft = 0; ft < NFT; ++ft) {
    const int fr = ft * 16 + r16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int dd = 32 * ks + 8 * g;
      const bool ok = fr < F && dd < p.d;
      qf[ft][ks] = *reinterpret_cast<const Frag*>(ok ? base + fr * fstride + dd : zero);
      kf[ft][ks] = *reinterpret_cast<const Frag*>(ok ? base + fr * fstride + C + dd : zero);
    }
  }
  // V^T fragments: lane (dv = 16*t + r16, quad g): k-slot j<4 <-> frame 4g+j, j>=4 <-> frame 16+4g+(j-4).  The operand wants
  // the FRAMES of one channel in a lane while memory has the channels of one frame contiguous: V rows are fetched like Q / K
  // (16 B per lane) and transposed through a wave-private LDS tile (2-byte gathers straight from global memory were 12-24
  // narrow requests per lane and held the kernel at 3.0 TB/s).  Row pitch DP + 8 elements: the four frame groups of a read
  // land 16 banks apart.
  __shared__ __attribute__((aligned(16))) unsigned short vsh[4][NFT * 16][DP + 8];
  unsigned short (*vt)[DP + 8] = vsh[threadIdx.x >> 6];
#pragma unroll
  for (int ft = 0; ft < NFT; ++ft) {
    const int fr = ft * 16 + r16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int dd = 32 * ks + 8 * g;
      const bool ok = fr < F && dd < p.d;
      *reinterpret_cast<Frag*>(&vt[fr][dd]) = *reinterpret_cast<const Frag*>(ok ? base + fr * fstride + 2 * C + dd : zero);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // same wave wrote and reads: LDS ops of a wave complete in order
  Frag vf[DVT];
#pragma unroll
  for (int t = 0; t < DVT; ++t) {
    const int dv = t * 16 + r16;
    unsigned short e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int fr = (j < 4) ? 4 * g + j : 16 + 4 * g + (j - 4);
      const bool ok = dv < DP && (j < 4 || NFT > 1);     // rows >= F and channels >= d hold zeros
      e[j] = ok ? vt[fr][dv] : (unsigned short)0;
    }
    u32x4 pk;
#pragma unroll
    for (int i = 0; i < 4; ++i) pk[i] = (unsigned)e[2 * i] | ((unsigned)e[2 * i + 1] << 16);
    vf[t] = __builtin_bit_cast(Frag, pk);
  }

  const float sl2e = p.scale * 1.44269504088896340736f;
  T* obase = reinterpret_cast<T*>(p.o) + ((long long)b * F * p.pixels + pix) * C + h * p.d;
  const long long ofstride = (long long)p.pixels * C;

#pragma unroll
  for (int qt = 0; qt < NFT; ++qt) {
    // scores of query frame (qt*16 + r16) against key frames kt*16 + 4g + r
    f32x4 s[NFT];
#pragma unroll
    for (int kt = 0; kt < NFT; ++kt) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a = TMma<T>::k32(kf[kt][ks], qf[qt][ks], a);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (kt * 16 + 4 * g + r >= F) a[r] = -INFINITY;
      s[kt] = a;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NFT; ++kt) mx = fmaxf(mx, fmaxf(fmaxf(s[kt][0], s[kt][1]), fmaxf(s[kt][2], s[kt][3])));
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float pv[8], sum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) pv[j] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NFT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f((s[kt][r] - mx) * sl2e);
        pv[4 * kt + r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    u32x4 pk;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      pk[i] = (unsigned)Pair16<T>::to_bits(pv[2 * i] * inv) | ((unsigned)Pair16<T>::to_bits(pv[2 * i + 1] * inv) << 16);
    const Frag pf = __builtin_bit_cast(Frag, pk);
    const int fq = qt * 16 + r16;
#pragma unroll
    for (int t = 0; t < DVT; ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = TMma<T>::k32(vf[t], pf, acc);
      const int dd = t * 16 + 4 * g;
      if (fq < F && dd < p.d) {
        float v[4] = {acc[0], acc[1], acc[2], acc[3]};
        ElemIO<T>::st4(obase + fq * ofstride + dd, v);
      }
    }
  }
}

// parity-mode kernel: one thread per (b, pixel, head, query frame); f32 throughout.
__global__ void __launch_bounds__(256) tattn_f32_kernel(const TAttnP p) {
  const long long total = (long long)p.clips * p.pixels * p.heads * p.frames;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int fq = (int)(i % p.frames);
  long long t = i / p.frames;
  const int h = (int)(t % p.heads);
  t /= p.heads;
  const int pix = (int)(t % p.pixels), b = (int)(t / p.pixels);
  const int F = p.frames, C = p.heads * p.d, ld = 3 * C;
  const float* base = reinterpret_cast<const float*>(p.qkv) + ((long long)b * F * p.pixels + pix) * ld + h * p.d;
  const long long fs = (long long)p.pixels * ld;
  const float* q = base + fq * fs;
  float sc[32];
  float mx = -INFINITY;
  for (int fk = 0; fk < F; ++fk) {
    const float* k = base + fk * fs + C;
    float a = 0.f;
    for (int c = 0; c < p.d; ++c) a = fmaf(q[c], k[c], a);
    sc[fk] = a * p.scale;
    mx = fmaxf(mx, sc[fk]);
  }
  float sum = 0.f;
  for (int fk = 0; fk < F; ++fk) { sc[fk] = expf(sc[fk] - mx); sum += sc[fk]; }
  const float inv = 1.0f / sum;
  float* o = reinterpret_cast<float*>(p.o) + (((long long)b * F + fq) * p.pixels + pix) * C + h * p.d;
  for (int c = 0; c < p.d; ++c) {
    float a = 0.f;
    for (int fk = 0; fk < F; ++fk) a = fmaf(sc[fk] * inv, base[fk * fs + 2 * C + c], a);
    o[c] = a;
  }
}

template <typename T, int DP, int DVT>
int launch_t(const TAttnP& p, hipStream_t st) {
  const long long ntask = (long long)p.clips * p.pixels * p.heads;
  dim3 grid((unsigned)((ntask + 3) / 4));
  if (p.frames <= 16) hipLaunchKernelGGL((tattn_bf16_kernel<T, DP, DVT, 1>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((tattn_bf16_kernel<T, DP, DVT, 2>), grid, dim3(256), 0, st, p);
  FYC_CHECK_LAUNCH("fyc_temporal_attention");
  return 0;
}

template <typename T>
int launch_d(const TAttnP& p, hipStream_t st) {
  if (p.d <= 32) return launch_t<T, 32, 2>(p, st);
  if (p.d <= 48) return launch_t<T, 64, 3>(p, st);
  if (p.d <= 64) return launch_t<T, 64, 4>(p, st);
  if (p.d <= 80) return launch_t<T, 96, 5>(p, st);
  if (p.d <= 96) return launch_t<T, 96, 6>(p, st);
  if (p.d <= 128) return launch_t<T, 128, 8>(p, st);
  return launch_t<T, 160, 10>(p, st);
}

}  // namespace

extern "C" int fyc_temporal_attention(const fyc_tattn_args* a, void* stream) {
  FYC_REQUIRE(a && a->qkv && a->o, "fyc_temporal_attention: null pointer");
  FYC_REQUIRE(g_fyc_zero_page != nullptr, "fyc_temporal_attention: fyc_init() not called");
  FYC_REQUIRE(a->clips > 0 && a->frames > 0 && a->pixels > 0 && a->heads > 0, "fyc_temporal_attention: bad sizes");
  FYC_REQUIRE(a->frames <= 32, "fyc_temporal_attention: frames=%d > 32 unsupported", a->frames);
  FYC_REQUIRE(a->d % 8 == 0 && a->d >= 8 && a->d <= 160, "fyc_temporal_attention: head dim %d", a->d);
  TAttnP p;
  p.qkv = a->qkv; p.o = a->o; p.clips = a->clips; p.frames = a->frames; p.pixels = a->pixels; p.heads = a->heads; p.d = a->d;
  p.scale = a->scale; p.zero = (const char*)g_fyc_zero_page;
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == FYC_F32) {
    const long long total = (long long)p.clips * p.pixels * p.heads * p.frames;
    hipLaunchKernelGGL(tattn_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p);
    FYC_CHECK_LAUNCH("fyc_temporal_attention(f32)");
    return 0;
  }
  FYC_REQUIRE(a->dtype == FYC_BF16 || a->dtype == FYC_F16, "fyc_temporal_attention: bad dtype");
  return a->dtype == FYC_F16 ? launch_d<f16_t>(p, st) : launch_d<bf16_t>(p, st);
}
