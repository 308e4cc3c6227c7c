// Microbenchmark (round 3): one wave per SIMD (4 waves per workgroup, one workgroup per CU), a stream of MFMAs with F other
// instructions per 16 cycles of matrix work between them: v_mfma_f32_16x16x32_bf16 (16-cycle) against v_mfma_f32_32x32x16_bf16
// (32-cycle) gaps.  Fillers: one ds_read_b128 per 32 MFMA cycles + scalar v_fma_f32 chains.  Prints cycles per 16x16x32-equivalent.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int BIG, int F>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) char smem[32768];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<float*>(smem)[i] = (float)i * 1e-6f;
  __syncthreads();
  bf16x8 b = *reinterpret_cast<const bf16x8*>(smem + lane * 16);
  f32x4 acc4[8];
  f32x16 acc16[2];
  for (int i = 0; i < 8; ++i) acc4[i] = (f32x4){0, 0, 0, 0};
  for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) acc16[i][e] = 0.f;
  float v[8];
  for (int e = 0; e < 8; ++e) v[e] = lane * 0.001f + e;
  const unsigned long long t0 = __builtin_readcyclecounter();
  bf16x8 a[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const bf16x8*>(smem + q * 1024 + lane * 16);
#pragma unroll 4
  for (int it = 0; it < iters; ++it) {
    // one "group" = 128 cycles of matrix work = 8 small or 4 big MFMAs, 4 fragment reads (for the NEXT group), 8 F VALU fillers
    bf16x8 n[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) n[q] = *reinterpret_cast<const bf16x8*>(smem + (((it + 1) * 4 + q) & 15) * 1024 + lane * 16);
    if constexpr (BIG) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc16[q & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], b, acc16[q & 1], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < 2 * F; ++f) v[(2 * q * F + f) & 7] = __builtin_fmaf(v[(2 * q * F + f) & 7], 1.0001f, 0.5f);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        acc4[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[q >> 1], b, acc4[q], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < F; ++f) v[(q * F + f) & 7] = __builtin_fmaf(v[(q * F + f) & 7], 1.0001f, 0.5f);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = n[q];
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc4[i][0] + v[i];
  for (int i = 0; i < 2; ++i) s += acc16[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int BIG, int F>
void run(float* out, unsigned long long* cyc, unsigned long long* h) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<BIG, F>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipMemcpy(h, cyc, 256 * 8, hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < 256; ++i) m += (double)h[i];
  m /= 256;
  printf("%s MFMA, %d VALU fillers + 0.5 ds_read_b128 per 16 matrix cycles: %.1f cycles per 16 matrix cycles\n", BIG ? "32x32x16 (32-cycle)" : "16x16x32 (16-cycle)", F, m / iters / 8);
}
int main() {
  float* out; unsigned long long *cyc, h[256];
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  run<0, 0>(out, cyc, h); run<1, 0>(out, cyc, h);
  run<0, 1>(out, cyc, h); run<1, 1>(out, cyc, h);
  run<0, 2>(out, cyc, h); run<1, 2>(out, cyc, h);
  run<0, 3>(out, cyc, h); run<1, 3>(out, cyc, h);
  run<0, 4>(out, cyc, h); run<1, 4>(out, cyc, h);
  run<0, 6>(out, cyc, h); run<1, 6>(out, cyc, h);
  return 0;
}
