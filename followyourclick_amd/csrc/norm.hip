// HBM-bound normalisation kernels (channels-last activations): GroupNorm statistics + apply(+SiLU),
// LayerNorm(+positional table), row softmax.  All statistics in f32/f64; 16-byte vector I/O.
#include "fyc_common.h"

namespace {

// ---- GroupNorm statistics ------------------------------------------------------------------
// grid = (row_chunks, samples).  A block walks `rows_per_block` rows of one sample; thread t owns the
// 8-channel chunk (t % C8) of rows (t / C8), (t / C8) + RPI, ...
// The sums are ORDER-FREE (bitwise repeatable; round 3 folded them with LDS float atomics and f64 device atomics, whose arrival
// order moved the last bits from run to run): the threads park their 16 partial sums in LDS, one thread per channel adds the
// row lanes of its channel in order, one thread per group adds its channels in order, and the block stores its {sum, sum sq}
// per group - straight into `stats` when one block covers the sample, else into part[sample][chunk][2][groups] (f32) that
// gn_stats_finish_kernel adds up chunk by chunk in f64.
template <typename T>
__global__ void __launch_bounds__(512) gn_stats_kernel(const T* __restrict__ x, double* __restrict__ stats, float* __restrict__ part,
                                                       int C, int groups, int rows_per_sample, int rows_per_block) {
  extern __shared__ float sh[];  // [nthr][16] parked sums, then [2][C] per channel
  const int tid = threadIdx.x, C8 = C >> 3, cpg = C / groups;
  const int sample = blockIdx.y;
  const int nthr = blockDim.x;
  float* chan = sh + nthr * 16;
  const int rpi = nthr / C8;  // rows per iteration (host guarantees blockDim >= C/8)
  const int row_begin = blockIdx.x * rows_per_block;
  const int row_end = min(row_begin + rows_per_block, rows_per_sample);
  const T* base = x + (long long)sample * rows_per_sample * C;
  if (tid < rpi * C8) {
    const int c8 = tid % C8, r0 = tid / C8;
    float s[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = q[i] = 0.f;
    // 4 independent 16-B loads in flight per thread (the kernel is pure streaming: latency, not math, bounds it)
    int r = row_begin + r0;
    for (; r + 3 * rpi < row_end; r += 4 * rpi) {
      float v0[8], v1[8], v2[8], v3[8];
      load8<T>(base + (long long)r * C + c8 * 8, v0);
      load8<T>(base + (long long)(r + rpi) * C + c8 * 8, v1);
      load8<T>(base + (long long)(r + 2 * rpi) * C + c8 * 8, v2);
      load8<T>(base + (long long)(r + 3 * rpi) * C + c8 * 8, v3);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        s[i] += (v0[i] + v1[i]) + (v2[i] + v3[i]);
        q[i] += (v0[i] * v0[i] + v1[i] * v1[i]) + (v2[i] * v2[i] + v3[i] * v3[i]);
      }
    }
    for (; r < row_end; r += rpi) {
      float v[8];
      load8<T>(base + (long long)r * C + c8 * 8, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) { s[i] += v[i]; q[i] += v[i] * v[i]; }
    }
    float* mine = sh + tid * 16;   // tid = r0 * C8 + c8
    *reinterpret_cast<f32x4*>(mine) = (f32x4){s[0], s[1], s[2], s[3]};
    *reinterpret_cast<f32x4*>(mine + 4) = (f32x4){s[4], s[5], s[6], s[7]};
    *reinterpret_cast<f32x4*>(mine + 8) = (f32x4){q[0], q[1], q[2], q[3]};
    *reinterpret_cast<f32x4*>(mine + 12) = (f32x4){q[4], q[5], q[6], q[7]};
  }
  __syncthreads();
  for (int c = tid; c < C; c += nthr) {               // rows of a channel, in order
    const float* src = sh + (c >> 3) * 16 + (c & 7);
    float cs = 0.f, cq = 0.f;
    for (int r0 = 0; r0 < rpi; ++r0) { cs += src[r0 * C8 * 16]; cq += src[r0 * C8 * 16 + 8]; }
    chan[c] = cs; chan[C + c] = cq;
  }
  __syncthreads();
  for (int gi = tid; gi < groups; gi += nthr) {       // channels of a group, in order
    float gs = 0.f, gq = 0.f;
    for (int c = gi * cpg; c < (gi + 1) * cpg; ++c) { gs += chan[c]; gq += chan[C + c]; }
    if (part == nullptr) {
      stats[((long long)sample * groups + gi) * 2 + 0] = (double)gs;
      stats[((long long)sample * groups + gi) * 2 + 1] = (double)gq;
    } else {
      float* dst = part + ((long long)sample * gridDim.x + blockIdx.x) * 2 * groups;
      dst[gi] = gs; dst[groups + gi] = gq;
    }
  }
}

// chunk partial sums of a sample -> stats, in chunk order (f64)
__global__ void __launch_bounds__(64) gn_stats_finish_kernel(const float* __restrict__ part, double* __restrict__ stats, int groups, int chunks) {
  const int sample = blockIdx.x;
  for (int gi = threadIdx.x; gi < groups; gi += 64) {
    double s = 0.0, q = 0.0;
    const float* src = part + (long long)sample * chunks * 2 * groups;
    for (int c = 0; c < chunks; ++c) { s += (double)src[c * 2 * groups + gi]; q += (double)src[c * 2 * groups + groups + gi]; }
    stats[((long long)sample * groups + gi) * 2 + 0] = s;
    stats[((long long)sample * groups + gi) * 2 + 1] = q;
  }
}

template <typename T>
__global__ void __launch_bounds__(256) gn_apply_kernel(const T* __restrict__ x, const double* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       T* __restrict__ y, long long chunks, int C, int groups,
                                                       int rows_per_sample, float eps, int silu) {
  const int C8 = C >> 3, cpg = C / groups;
  const double inv_cnt = 1.0 / ((double)rows_per_sample * cpg);
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < (unsigned)chunks; idx += gridDim.x * 256u) {
    const unsigned row = idx / (unsigned)C8;
    const int c8 = (int)(idx - row * (unsigned)C8);
    const int sample = (int)(row / (unsigned)rows_per_sample);
    float v[8];
    load8<T>(x + (size_t)idx * 8, v);
    // one integer division per chunk; the group index then advances incrementally
    int gi = (c8 * 8) / cpg, left = cpg - (c8 * 8 - gi * cpg);
    float mean = 0.f, rstd = 0.f;
    bool fresh = true;
    float g8[8], b8[8];
    *reinterpret_cast<f32x4*>(g8) = *reinterpret_cast<const f32x4*>(gamma + c8 * 8);
    *reinterpret_cast<f32x4*>(g8 + 4) = *reinterpret_cast<const f32x4*>(gamma + c8 * 8 + 4);
    *reinterpret_cast<f32x4*>(b8) = *reinterpret_cast<const f32x4*>(beta + c8 * 8);
    *reinterpret_cast<f32x4*>(b8 + 4) = *reinterpret_cast<const f32x4*>(beta + c8 * 8 + 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (fresh) {
        const double s = stats[((long long)sample * groups + gi) * 2], q = stats[((long long)sample * groups + gi) * 2 + 1];
        const double m = s * inv_cnt;
        double var = q * inv_cnt - m * m;   // f64 subtraction keeps the cancellation harmless
        var = var > 0.0 ? var : 0.0;
        mean = (float)m;
        rstd = rsqrtf((float)var + eps);
        fresh = false;
      }
      const float o = (v[i] - mean) * rstd * g8[i] + b8[i];
      v[i] = silu ? silu_f(o) : o;
      if (--left == 0) { left = cpg; ++gi; fresh = true; }
    }
    store8<T>(y + (size_t)idx * 8, v);
  }
}

// ---- per-(sample, channel) sums from the row-tile partials of the producing GEMM (fyc_gemm chan_parts) ----------------------
// block = 32 channels x 8 lanes of one output sample: the lanes stride over the row tiles that overlap the sample (up to 256
// for a cross-frame norm at 64x64), then fold through LDS.  Tiny (<= a few MB read) but latency-bound: keep it parallel.
__global__ void __launch_bounds__(256) chan_stats_reduce_kernel(const float* __restrict__ parts, double* __restrict__ cs, int N,
                                                              int tiles_m, int tile_rows, int slots, int cs_rows, int group, int nsamp) {
  __shared__ double red[2][8][32];
  const int c = threadIdx.x & 31, ln = threadIdx.x >> 5;
  const int n = blockIdx.x * 32 + c, o = blockIdx.y;      // output sample o = `group` consecutive statistics samples
  const long long row0 = (long long)o * group * cs_rows, row1 = row0 + (long long)group * cs_rows;
  const int t0 = (int)(row0 / tile_rows);
  int t1 = (int)((row1 - 1) / tile_rows);
  if (t1 >= tiles_m) t1 = tiles_m - 1;
  double s = 0.0, q = 0.0;
  if (n < N) {
    for (int t = t0 + ln; t <= t1; t += 8) {
      const int first = (int)(((long long)t * tile_rows) / cs_rows);
      for (int sl = 0; sl < slots; ++sl) {
        const int f = first + sl;
        if (f < o * group || f >= (o + 1) * group || f >= nsamp) continue;
        const float2 v = *reinterpret_cast<const float2*>(parts + (((long long)t * slots + sl) * N + n) * 2);
        s += (double)v.x; q += (double)v.y;
      }
    }
  }
  red[0][ln][c] = s; red[1][ln][c] = q;
  __syncthreads();
  if (ln == 0 && n < N) {
#pragma unroll
    for (int i = 1; i < 8; ++i) { s += red[0][i][c]; q += red[1][i][c]; }
    double* dst = cs + ((long long)o * N + n) * 2;
    dst[0] = s; dst[1] = q;
  }
}

// ---- GroupNorm apply from per-(sample, channel) sums (+SiLU), optional channel concat of two sources ---------------------
// The sums come from the epilogues of the GEMMs / convs that produced x1 (and x2): no statistics pass over the tensor.
// grid = (row chunks, samples).  Prologue: 8 lanes per group fold the group's channels (f64), every thread then keeps
// scale = rstd*gamma and shift = beta - mean*scale of ITS 8 channels in registers; the loop is load - fma - (silu) - store.
template <typename T>
__global__ void __launch_bounds__(512) gn_apply_cs_kernel(const T* __restrict__ x1, const double* __restrict__ cs1, int C1,
                                                          const T* __restrict__ x2, const double* __restrict__ cs2, int C2,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          T* __restrict__ y, int groups, int rows_per_sample, int rows_per_block,
                                                          float eps, int silu, int stat_samples) {
  extern __shared__ float sh[];   // [2][groups]: mean, rstd
  const int tid = threadIdx.x, nthr = blockDim.x, C = C1 + C2, C8 = C >> 3, cpg = C / groups;
  const int sample = blockIdx.y;
  {
    const double inv_cnt = 1.0 / ((double)rows_per_sample * cpg);
    for (int gi = tid >> 3; gi < groups; gi += nthr >> 3) {
      double s = 0.0, q = 0.0;
      // the sums come per statistics sample (a frame); a GroupNorm sample spans `stat_samples` of them (cross-frame norms: F)
      for (int c = gi * cpg + (tid & 7); c < (gi + 1) * cpg; c += 8) {
        for (int f = 0; f < stat_samples; ++f) {
          const long long ss = (long long)sample * stat_samples + f;
          const double* src = c < C1 ? cs1 + (ss * C1 + c) * 2 : cs2 + (ss * C2 + (c - C1)) * 2;
          s += src[0]; q += src[1];
        }
      }
#pragma unroll
      for (int o = 4; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
      if ((tid & 7) == 0) {
        const double m = s * inv_cnt;
        double var = q * inv_cnt - m * m;   // f64 subtraction keeps the cancellation harmless
        var = var > 0.0 ? var : 0.0;
        sh[gi] = (float)m;
        sh[groups + gi] = rsqrtf((float)var + eps);
      }
    }
  }
  __syncthreads();
  const int rpi = nthr / C8;
  if (tid >= rpi * C8) return;
  const int c8 = tid % C8, r0 = tid / C8;
  float sc[8], sf[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = c8 * 8 + i, gi = c / cpg;
    sc[i] = sh[groups + gi] * gamma[c];
    sf[i] = beta[c] - sh[gi] * sc[i];
  }
  const bool from1 = c8 * 8 < C1;
  const T* src = from1 ? x1 + (long long)sample * rows_per_sample * C1 + c8 * 8 : x2 + (long long)sample * rows_per_sample * C2 + (c8 * 8 - C1);
  const int ld = from1 ? C1 : C2;
  T* dst = y + (long long)sample * rows_per_sample * C + c8 * 8;
  const int row_begin = blockIdx.x * rows_per_block;
  const int row_end = min(row_begin + rows_per_block, rows_per_sample);
  auto fin = [&](float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float o = __builtin_fmaf(v[i], sc[i], sf[i]);
      v[i] = silu ? silu_f(o) : o;
    }
  };
  int r = row_begin + r0;
  for (; r + 3 * rpi < row_end; r += 4 * rpi) {
    float v0[8], v1[8], v2[8], v3[8];
    load8<T>(src + (long long)r * ld, v0);
    load8<T>(src + (long long)(r + rpi) * ld, v1);
    load8<T>(src + (long long)(r + 2 * rpi) * ld, v2);
    load8<T>(src + (long long)(r + 3 * rpi) * ld, v3);
    fin(v0); fin(v1); fin(v2); fin(v3);
    store8<T>(dst + (long long)r * C, v0);
    store8<T>(dst + (long long)(r + rpi) * C, v1);
    store8<T>(dst + (long long)(r + 2 * rpi) * C, v2);
    store8<T>(dst + (long long)(r + 3 * rpi) * C, v3);
  }
  for (; r < row_end; r += rpi) {
    float v[8];
    load8<T>(src + (long long)r * ld, v);
    fin(v);
    store8<T>(dst + (long long)r * C, v);
  }
}

// ---- LayerNorm: 16 lanes per row (4 rows per wave, 16 rows per block) -----------------------------
// A row of C = 320..1280 channels is only 640..2560 B: one wave per row leaves most lanes idle and
// little memory in flight.  Here 16 lanes own a row (lane p takes 16-B chunks p, p+16, ...), the two
// reductions are 4 xor-shuffles inside the 16-lane group, and a wave keeps 4 rows in flight.
template <typename T, int MAXCH>
__global__ void __launch_bounds__(256) layernorm_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ pe,
                                                        T* __restrict__ y, int rows, int C, float eps, int pe_div,
                                                        int pe_rows, float* __restrict__ stats = nullptr) {
  const int p = threadIdx.x & 15;
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool live = row < rows;
  const int C8 = C >> 3;
  const T* xr = x + (long long)(live ? row : 0) * C;
  float v[MAXCH][8];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < MAXCH; ++k) {
    const int c8 = p + k * 16;
    if (live && c8 < C8) {
      load8<T>(xr + c8 * 8, v[k]);
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[k][i];
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < MAXCH; ++k) {
    const int c8 = p + k * 16;
    if (live && c8 < C8) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = v[k][i] - mean; q += d * d; }
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o);
  if (!live) return;
  const float rstd = rsqrtf(q / (float)C + eps);
  if (stats) {                                   // statistics-only mode (LayerNorm folded into the consuming GEMM)
    if (p == 0) *reinterpret_cast<float2*>(stats + 2ll * row) = make_float2(mean, rstd);
    return;
  }
  const float* per = pe ? pe + (long long)((row / pe_div) % pe_rows) * C : nullptr;
  T* yr = y + (long long)row * C;
#pragma unroll
  for (int k = 0; k < MAXCH; ++k) {
    const int c8 = p + k * 16;
    if (c8 < C8) {
      float o8[8], g8[8], b8[8];
      *reinterpret_cast<f32x4*>(g8) = *reinterpret_cast<const f32x4*>(gamma + c8 * 8);
      *reinterpret_cast<f32x4*>(g8 + 4) = *reinterpret_cast<const f32x4*>(gamma + c8 * 8 + 4);
      *reinterpret_cast<f32x4*>(b8) = *reinterpret_cast<const f32x4*>(beta + c8 * 8);
      *reinterpret_cast<f32x4*>(b8 + 4) = *reinterpret_cast<const f32x4*>(beta + c8 * 8 + 4);
#pragma unroll
      for (int i = 0; i < 8; ++i) o8[i] = (v[k][i] - mean) * rstd * g8[i] + b8[i];
      if (per) {
#pragma unroll
        for (int i = 0; i < 8; ++i) o8[i] += per[c8 * 8 + i];
      }
      store8<T>(yr + c8 * 8, o8);
    }
  }
}

// ---- row softmax in place: one block per row -------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) softmax_rows_kernel(T* __restrict__ x, int cols_all, int ld, int causal_rows) {
  __shared__ float red[8];
  T* xr = x + (long long)blockIdx.x * ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // causal: query i = row % causal_rows attends to keys 0..i; masked probabilities are written as exact zeros
  const int cols = causal_rows > 0 ? min(cols_all, (int)(blockIdx.x % (unsigned)causal_rows) + 1) : cols_all;
  for (int c = cols + tid; c < cols_all; c += 256) ElemIO<T>::st(xr + c, 0.f);
  float m = -INFINITY;
  for (int c = tid; c < cols; c += 256) m = fmaxf(m, ElemIO<T>::ld(xr + c));
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.f;
  for (int c = tid; c < cols; c += 256) s += expf(ElemIO<T>::ld(xr + c) - m);
  s = wave_sum(s);
  if (lane == 0) red[4 + wave] = s;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (int c = tid; c < cols; c += 256) ElemIO<T>::st(xr + c, expf(ElemIO<T>::ld(xr + c) - m) * inv);
}

}  // namespace

namespace {
// row chunks a sample is cut into (grid.x) and rows per chunk: ~1024 blocks in total (4 per CU), >= 128 rows per block so the
// unrolled loop has work
void gn_stats_geometry(const fyc_gn_stats_args* a, int& chunks, int& rpb) {
  const int samples = a->rows / a->rows_per_sample;
  chunks = (int)ceil_div64(1024, samples);
  rpb = (int)ceil_div64(a->rows_per_sample, chunks);
  if (rpb < 128) rpb = 128;
  chunks = (int)ceil_div64(a->rows_per_sample, rpb);
}
}  // namespace

extern "C" int64_t fyc_gn_stats_workspace(const fyc_gn_stats_args* a) {
  if (a == nullptr || a->rows_per_sample <= 0 || a->rows <= 0 || a->groups <= 0) return 0;
  int chunks = 1, rpb = 0;
  gn_stats_geometry(a, chunks, rpb);
  return chunks > 1 ? (int64_t)(a->rows / a->rows_per_sample) * chunks * 2 * a->groups * (int64_t)sizeof(float) : 0;
}

extern "C" int fyc_gn_stats(const fyc_gn_stats_args* a, void* stream) {
  FYC_REQUIRE(a && a->x && a->stats, "fyc_gn_stats: null pointer");
  FYC_REQUIRE(a->C % 8 == 0 && a->C <= 4096 && a->groups > 0 && a->C % a->groups == 0, "fyc_gn_stats: C=%d groups=%d", a->C, a->groups);
  FYC_REQUIRE(a->rows_per_sample > 0 && a->rows % a->rows_per_sample == 0, "fyc_gn_stats: rows=%d rows_per_sample=%d", a->rows, a->rows_per_sample);
  FYC_REQUIRE(a->C / 8 <= 512, "fyc_gn_stats: C too large");
  hipStream_t st = (hipStream_t)stream;
  const int samples = a->rows / a->rows_per_sample;
  int chunks = 1, rpb = 0;
  gn_stats_geometry(a, chunks, rpb);
  const int64_t need = fyc_gn_stats_workspace(a);
  FYC_REQUIRE(chunks == 1 || (a->workspace != nullptr && a->workspace_bytes >= need && ((uintptr_t)a->workspace % 4) == 0),
              "fyc_gn_stats: %lld workspace bytes needed (fyc_gn_stats_workspace), got %lld", (long long)need, (long long)(a->workspace ? a->workspace_bytes : 0));
  float* part = chunks > 1 ? a->workspace : nullptr;
  dim3 grid(chunks, samples);
  const int c8 = a->C / 8;
  const int nthr = c8 >= 256 ? ((c8 + 63) / 64) * 64 : 256;  // thread t <-> fixed 8-channel chunk t % c8
  const size_t sh = sizeof(float) * ((size_t)nthr * 16 + 2 * a->C);
  if (a->dtype == FYC_BF16)
    hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, grid, dim3(nthr), sh, st, (const bf16_t*)a->x, a->stats, part, a->C, a->groups, a->rows_per_sample, rpb);
  else if (a->dtype == FYC_F16)
    hipLaunchKernelGGL(gn_stats_kernel<f16_t>, grid, dim3(nthr), sh, st, (const f16_t*)a->x, a->stats, part, a->C, a->groups, a->rows_per_sample, rpb);
  else if (a->dtype == FYC_F32)
    hipLaunchKernelGGL(gn_stats_kernel<float>, grid, dim3(nthr), sh, st, (const float*)a->x, a->stats, part, a->C, a->groups, a->rows_per_sample, rpb);
  else FYC_FAIL(-2, "fyc_gn_stats: bad dtype");
  FYC_CHECK_LAUNCH("fyc_gn_stats");
  if (chunks > 1) {
    hipLaunchKernelGGL(gn_stats_finish_kernel, dim3(samples), dim3(64), 0, st, part, a->stats, a->groups, chunks);
    FYC_CHECK_LAUNCH("fyc_gn_stats (finish)");
  }
  return 0;
}

extern "C" int fyc_gn_apply(const fyc_gn_apply_args* a, void* stream) {
  FYC_REQUIRE(a && a->x && a->stats && a->gamma && a->beta && a->y, "fyc_gn_apply: null pointer");
  FYC_REQUIRE(a->C % 8 == 0 && a->groups > 0 && a->C % a->groups == 0, "fyc_gn_apply: C=%d groups=%d", a->C, a->groups);
  FYC_REQUIRE(a->rows_per_sample > 0 && a->rows % a->rows_per_sample == 0, "fyc_gn_apply: rows/rows_per_sample");
  hipStream_t st = (hipStream_t)stream;
  const long long chunks = (long long)a->rows * (a->C / 8);
  FYC_REQUIRE(chunks < (1ll << 32) - 65536ll * 256, "fyc_gn_apply: tensor too large for 32-bit chunk index");
  int blocks = (int)(ceil_div64(chunks, 256) < 8192 ? ceil_div64(chunks, 256) : 8192);
  if (a->dtype == FYC_BF16)
    hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)a->x, a->stats, a->gamma, a->beta,
                       (bf16_t*)a->y, chunks, a->C, a->groups, a->rows_per_sample, a->eps, a->silu);
  else if (a->dtype == FYC_F16)
    hipLaunchKernelGGL(gn_apply_kernel<f16_t>, dim3(blocks), dim3(256), 0, st, (const f16_t*)a->x, a->stats, a->gamma, a->beta,
                       (f16_t*)a->y, chunks, a->C, a->groups, a->rows_per_sample, a->eps, a->silu);
  else if (a->dtype == FYC_F32)
    hipLaunchKernelGGL(gn_apply_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)a->x, a->stats, a->gamma, a->beta,
                       (float*)a->y, chunks, a->C, a->groups, a->rows_per_sample, a->eps, a->silu);
  else FYC_FAIL(-2, "fyc_gn_apply: bad dtype");
  FYC_CHECK_LAUNCH("fyc_gn_apply");
  return 0;
}

extern "C" int fyc_chan_stats_reduce(const fyc_chan_stats_reduce_args* a, void* stream) {
  FYC_REQUIRE(a && a->parts && a->cs, "fyc_chan_stats_reduce: null pointer");
  FYC_REQUIRE(a->rows > 0 && a->N > 0 && a->cs_rows > 0 && a->rows % a->cs_rows == 0 && a->tile_rows > 0 && a->slots >= 1,
              "fyc_chan_stats_reduce: rows=%d N=%d cs_rows=%d tile_rows=%d slots=%d", a->rows, a->N, a->cs_rows, a->tile_rows, a->slots);
  const int out_rows = a->out_rows > 0 ? a->out_rows : a->cs_rows;
  FYC_REQUIRE(out_rows % a->cs_rows == 0 && a->rows % out_rows == 0, "fyc_chan_stats_reduce: out_rows=%d must be a multiple of cs_rows=%d dividing rows=%d", out_rows, a->cs_rows, a->rows);
  const int tiles_m = (a->rows + a->tile_rows - 1) / a->tile_rows;
  dim3 grid((a->N + 31) / 32, a->rows / out_rows);
  hipLaunchKernelGGL(chan_stats_reduce_kernel, grid, dim3(256), 0, (hipStream_t)stream, a->parts, a->cs, a->N, tiles_m, a->tile_rows, a->slots, a->cs_rows,
                     out_rows / a->cs_rows, a->rows / a->cs_rows);
  FYC_CHECK_LAUNCH("fyc_chan_stats_reduce");
  return 0;
}

extern "C" int fyc_gn_apply_cs(const fyc_gn_apply_cs_args* a, void* stream) {
  FYC_REQUIRE(a && a->x1 && a->cs1 && a->gamma && a->beta && a->y, "fyc_gn_apply_cs: null pointer");
  FYC_REQUIRE((a->x2 == nullptr) == (a->C2 == 0) && (a->x2 == nullptr) == (a->cs2 == nullptr), "fyc_gn_apply_cs: x2 / cs2 / C2 must come together");
  const int C = a->C1 + a->C2;
  FYC_REQUIRE(a->C1 > 0 && a->C1 % 8 == 0 && a->C2 % 8 == 0 && a->groups > 0 && C % a->groups == 0 && C / 8 <= 512, "fyc_gn_apply_cs: C1=%d C2=%d groups=%d", a->C1, a->C2, a->groups);
  FYC_REQUIRE(a->rows_per_sample > 0 && a->rows % a->rows_per_sample == 0, "fyc_gn_apply_cs: rows=%d rows_per_sample=%d", a->rows, a->rows_per_sample);
  const int cs_rows = a->cs_rows > 0 ? a->cs_rows : a->rows_per_sample;
  FYC_REQUIRE(a->rows_per_sample % cs_rows == 0, "fyc_gn_apply_cs: rows_per_sample=%d is not a multiple of cs_rows=%d", a->rows_per_sample, cs_rows);
  const int stat_samples = a->rows_per_sample / cs_rows;
  hipStream_t st = (hipStream_t)stream;
  const int samples = a->rows / a->rows_per_sample;
  const int c8 = C / 8;
  const int nthr = c8 >= 256 ? ((c8 + 63) / 64) * 64 : 256;
  const int rpi = nthr / c8;
  // ~2048 blocks in total, at least 8 row iterations per block so that the prologue (statistics fold) amortises
  int chunks = (int)ceil_div64(2048, samples);
  int rpb = (int)ceil_div64(a->rows_per_sample, chunks);
  if (rpb < 8 * rpi) rpb = 8 * rpi;
  chunks = (int)ceil_div64(a->rows_per_sample, rpb);
  dim3 grid(chunks, samples);
  const size_t sh = sizeof(float) * 2 * a->groups;
  if (a->dtype == FYC_BF16)
    hipLaunchKernelGGL(gn_apply_cs_kernel<bf16_t>, grid, dim3(nthr), sh, st, (const bf16_t*)a->x1, a->cs1, a->C1, (const bf16_t*)a->x2, a->cs2, a->C2,
                       a->gamma, a->beta, (bf16_t*)a->y, a->groups, a->rows_per_sample, rpb, a->eps, a->silu, stat_samples);
  else if (a->dtype == FYC_F16)
    hipLaunchKernelGGL(gn_apply_cs_kernel<f16_t>, grid, dim3(nthr), sh, st, (const f16_t*)a->x1, a->cs1, a->C1, (const f16_t*)a->x2, a->cs2, a->C2,
                       a->gamma, a->beta, (f16_t*)a->y, a->groups, a->rows_per_sample, rpb, a->eps, a->silu, stat_samples);
  else if (a->dtype == FYC_F32)
    hipLaunchKernelGGL(gn_apply_cs_kernel<float>, grid, dim3(nthr), sh, st, (const float*)a->x1, a->cs1, a->C1, (const float*)a->x2, a->cs2, a->C2,
                       a->gamma, a->beta, (float*)a->y, a->groups, a->rows_per_sample, rpb, a->eps, a->silu, stat_samples);
  else FYC_FAIL(-2, "fyc_gn_apply_cs: bad dtype");
  FYC_CHECK_LAUNCH("fyc_gn_apply_cs");
  return 0;
}

extern "C" int fyc_layernorm(const fyc_layernorm_args* a, void* stream) {
  FYC_REQUIRE(a && a->x && a->gamma && a->beta && a->y, "fyc_layernorm: null pointer");
  FYC_REQUIRE(a->C % 8 == 0 && a->C <= 2048 && a->rows > 0, "fyc_layernorm: C=%d (multiple of 8, <= 2048)", a->C);
  FYC_REQUIRE(a->pe == nullptr || (a->pe_div > 0 && a->pe_rows > 0), "fyc_layernorm: pe_div/pe_rows");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((a->rows + 15) / 16);
  const int pd = a->pe ? a->pe_div : 1, pr = a->pe ? a->pe_rows : 1;
#define FYC_LN(T, MAXC8) hipLaunchKernelGGL((layernorm_kernel<T, MAXC8>), grid, dim3(256), 0, st, (const T*)a->x, a->gamma, a->beta, a->pe, (T*)a->y, a->rows, a->C, a->eps, pd, pr)
  const int need = (a->C / 8 + 15) / 16;  // 16-B chunks per lane
  if (a->dtype == FYC_BF16) {
    if (need <= 3) FYC_LN(bf16_t, 3); else if (need <= 5) FYC_LN(bf16_t, 5); else if (need <= 10) FYC_LN(bf16_t, 10); else FYC_LN(bf16_t, 16);
  } else if (a->dtype == FYC_F16) {
    if (need <= 3) FYC_LN(f16_t, 3); else if (need <= 5) FYC_LN(f16_t, 5); else if (need <= 10) FYC_LN(f16_t, 10); else FYC_LN(f16_t, 16);
  } else if (a->dtype == FYC_F32) {
    if (need <= 3) FYC_LN(float, 3); else if (need <= 5) FYC_LN(float, 5); else if (need <= 10) FYC_LN(float, 10); else FYC_LN(float, 16);
  } else FYC_FAIL(-2, "fyc_layernorm: bad dtype");
#undef FYC_LN
  FYC_CHECK_LAUNCH("fyc_layernorm");
  return 0;
}

extern "C" int fyc_row_stats(const fyc_row_stats_args* a, void* stream) {
  FYC_REQUIRE(a && a->x && a->stats, "fyc_row_stats: null pointer");
  FYC_REQUIRE(a->rows > 0 && a->C % 8 == 0 && a->C >= 8 && a->C <= 2048, "fyc_row_stats: rows=%d C=%d (C must be a multiple of 8 in [8, 2048])", a->rows, a->C);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((a->rows + 15) / 16);
#define FYC_RS(T, MAXC8) hipLaunchKernelGGL((layernorm_kernel<T, MAXC8>), grid, dim3(256), 0, st, (const T*)a->x, nullptr, nullptr, nullptr, (T*)nullptr, a->rows, a->C, a->eps, 1, 1, a->stats)
  const int need = (a->C / 8 + 15) / 16;
  if (a->dtype == FYC_BF16) {
    if (need <= 3) FYC_RS(bf16_t, 3); else if (need <= 5) FYC_RS(bf16_t, 5); else if (need <= 10) FYC_RS(bf16_t, 10); else FYC_RS(bf16_t, 16);
  } else if (a->dtype == FYC_F16) {
    if (need <= 3) FYC_RS(f16_t, 3); else if (need <= 5) FYC_RS(f16_t, 5); else if (need <= 10) FYC_RS(f16_t, 10); else FYC_RS(f16_t, 16);
  } else if (a->dtype == FYC_F32) {
    if (need <= 3) FYC_RS(float, 3); else if (need <= 5) FYC_RS(float, 5); else if (need <= 10) FYC_RS(float, 10); else FYC_RS(float, 16);
  } else FYC_FAIL(-2, "fyc_row_stats: bad dtype");
#undef FYC_RS
  FYC_CHECK_LAUNCH("fyc_row_stats");
  return 0;
}

extern "C" int fyc_softmax_rows(const fyc_softmax_args* a, void* stream) {
  FYC_REQUIRE(a && a->x && a->rows > 0 && a->cols > 0 && a->ld >= a->cols, "fyc_softmax_rows: bad args");
  FYC_REQUIRE(a->rows < (1ll << 31) && a->causal_rows >= 0, "fyc_softmax_rows: too many rows / negative causal_rows");
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == FYC_BF16)
    hipLaunchKernelGGL(softmax_rows_kernel<bf16_t>, dim3((unsigned)a->rows), dim3(256), 0, st, (bf16_t*)a->x, a->cols, a->ld, a->causal_rows);
  else if (a->dtype == FYC_F16)
    hipLaunchKernelGGL(softmax_rows_kernel<f16_t>, dim3((unsigned)a->rows), dim3(256), 0, st, (f16_t*)a->x, a->cols, a->ld, a->causal_rows);
  else if (a->dtype == FYC_F32)
    hipLaunchKernelGGL(softmax_rows_kernel<float>, dim3((unsigned)a->rows), dim3(256), 0, st, (float*)a->x, a->cols, a->ld, a->causal_rows);
  else FYC_FAIL(-2, "fyc_softmax_rows: bad dtype");
  FYC_CHECK_LAUNCH("fyc_softmax_rows");
  return 0;
}
