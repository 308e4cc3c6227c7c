// Micro-benchmark: how fast can one CU fill LDS from L2-resident global memory?
//   mode 0: global_load_lds (16 B/lane DMA straight into LDS)           - what fyc_gemm_kernel uses
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128               - register-staged
//   mode 2: global_load_dwordx4 only (no LDS write)                     - upper bound of the vector-memory path
// One persistent block per CU streams `tiles` tiles of TILE bytes with a 2-stage ring, like the GEMM main loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int MODE, int NT, int TILE>
__global__ void __launch_bounds__(NT) fill_kernel(const char* __restrict__ src, float* __restrict__ sink, long long src_bytes, int tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int IT = TILE / (NT * 16);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // src_bytes = per-XCD window: block b runs on XCD b % 8 and stays inside its own window, so a small window is L2-resident
  const char* base = src + (long long)(blockIdx.x & 7) * src_bytes;
  long long off = ((long long)(blockIdx.x >> 3) * 7919 * TILE) % (src_bytes - TILE);
  off &= ~15ll;
  f32x4 acc = {0, 0, 0, 0};
  for (int t = 0; t < tiles; ++t) {
    char* st = smem + (t & 1) * TILE;
    const char* g = base + off;
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < IT; ++i) glds16(g + (i * NT + tid) * 16, st + (i * NT + wave * 64) * 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      f32x4 v[IT];
#pragma unroll
      for (int i = 0; i < IT; ++i) v[i] = *reinterpret_cast<const f32x4*>(g + (i * NT + tid) * 16);
      if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < IT; ++i) *reinterpret_cast<f32x4*>(st + (i * NT + tid) * 16) = v[i];
      } else {
#pragma unroll
        for (int i = 0; i < IT; ++i) acc += v[i];
      }
    }
    __builtin_amdgcn_s_barrier();
    if (MODE != 2 && (t & 63) == 63) acc += *reinterpret_cast<const f32x4*>(st + tid * 16);   // keep the LDS contents observable
    off = (off + TILE * 263ll) % (src_bytes - TILE);
    off &= ~15ll;
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

template <int MODE, int NT, int TILE>
void run(const char* name, const char* src, float* sink, long long bytes, int ncu, int bpc = 1) {
  auto k = fill_kernel<MODE, NT, TILE>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TILE);
  const int tiles = 4000;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(ncu * bpc), dim3(NT), 2 * TILE, 0, src, sink, bytes, tiles);
    hipEventRecord(b);
    hipEventSynchronize(b);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double tot = (double)ncu * bpc * tiles * TILE;
  printf("%-34s NT=%3d TILE=%3dK: %7.2f TB/s chip, %6.1f GB/s/CU, %5.1f B/clk/CU @2.4GHz\n", name, NT, TILE / 1024, tot / ms / 1e9, tot / ms / 1e6 / ncu,
         tot / ms / 1e6 / ncu / 2.4);
}

int main() {
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  const int ncu = pr.multiProcessorCount;
  char* src; float* sink;
  hipMalloc(&src, 8ll * (24 << 20)); hipMalloc(&sink, 64);
  hipMemset(src, 1, 8ll * (24 << 20));
  printf("CUs %d\n", ncu);
  for (long long bytes : {2ll << 20, 24ll << 20}) {
  printf("---- per-XCD window %lld MB (%s)\n", bytes >> 20, bytes <= (3 << 20) ? "L2-resident" : "Infinity Cache / HBM");
  run<0, 256, 32768>("global_load_lds", src, sink, bytes, ncu);
  run<1, 256, 32768>("global_load -> vgpr -> ds_write", src, sink, bytes, ncu);
  run<2, 256, 32768>("global_load only", src, sink, bytes, ncu);
  run<0, 512, 65536>("global_load_lds", src, sink, bytes, ncu);
  run<1, 512, 65536>("global_load -> vgpr -> ds_write", src, sink, bytes, ncu);
  run<2, 512, 65536>("global_load only", src, sink, bytes, ncu);
  run<0, 256, 65536>("global_load_lds", src, sink, bytes, ncu);
  run<1, 256, 65536>("global_load -> vgpr -> ds_write", src, sink, bytes, ncu);
  // two blocks per CU (grid 2x): does a second independent stream add fill bandwidth?
  run<0, 256, 32768>("global_load_lds, 2 blocks/CU", src, sink, bytes, ncu, 2);
  run<1, 256, 32768>("ld->vgpr->ds_write, 2 blocks/CU", src, sink, bytes, ncu, 2);
  run<2, 256, 32768>("global_load only, 2 blocks/CU", src, sink, bytes, ncu, 2);
  }
  return 0;
}
