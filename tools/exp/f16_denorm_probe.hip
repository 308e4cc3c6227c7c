// probe (GPU): what v_cvt_pk_f16_f32 / v_cvt_f16_f32 give for results in the f16 denormal range, and what v_mfma_f32_16x16x32_f16
// does with denormal / zero / 64.0 operands.  hipcc --offload-arch=gfx950 tools/exp/f16_denorm_probe.hip -o tools/exp/f16_denorm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void cvt(const float* in, unsigned* pk, unsigned short* one, float* ex) {
  int i = threadIdx.x;
  f2 v = {in[2 * i], in[2 * i + 1]};
  pk[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2));
  one[2 * i] = __builtin_bit_cast(unsigned short, (_Float16)in[2 * i]);
  one[2 * i + 1] = __builtin_bit_cast(unsigned short, (_Float16)in[2 * i + 1]);
  ex[i] = __builtin_amdgcn_exp2f(-(float)i * 2.0f);
  f2 e = {__builtin_amdgcn_exp2f(-(float)i * 2.0f), __builtin_amdgcn_exp2f(-(float)i * 2.0f - 1.0f)};
  pk[64 + i] = __builtin_bit_cast(unsigned, __builtin_convertvector(e, h2));
}
__global__ void mm(const unsigned short* abits, const unsigned short* bbits, float* out) {
  int l = threadIdx.x;
  h8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = __builtin_bit_cast(_Float16, abits[e]); b[e] = __builtin_bit_cast(_Float16, bbits[(l & 15) * 8 + e]); }
  f4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
  float h_in[128]; for (int i = 0; i < 128; ++i) h_in[i] = ldexpf(1.0f + 0.25f * (i & 3), -(i / 2) - 8);
  float *d_in, *d_ex, *d_out; unsigned* d_pk; unsigned short *d_one, *d_a, *d_b;
  hipMalloc(&d_in, 512); hipMalloc(&d_pk, 512); hipMalloc(&d_one, 256); hipMalloc(&d_ex, 256); hipMalloc(&d_a, 16); hipMalloc(&d_b, 256); hipMalloc(&d_out, 1024);
  hipMemcpy(d_in, h_in, 512, hipMemcpyHostToDevice);
  cvt<<<1, 64>>>(d_in, d_pk, d_one, d_ex);
  unsigned pk[128]; unsigned short one[128]; float ex[64];
  hipMemcpy(pk, d_pk, 512, hipMemcpyDeviceToHost); hipMemcpy(one, d_one, 256, hipMemcpyDeviceToHost); hipMemcpy(ex, d_ex, 256, hipMemcpyDeviceToHost);
  for (int i = 0; i < 24; ++i) printf("in %.4e %.4e  pk %04x %04x  scalar %04x %04x | exp2(-%d)=%.3e pk(exp2) %04x %04x\n", h_in[2 * i], h_in[2 * i + 1], pk[i] & 0xffff, pk[i] >> 16, one[2 * i], one[2 * i + 1], 2 * i, ex[i], pk[64 + i] & 0xffff, pk[64 + i] >> 16);
  // A: all 1.0; B column j: element e = pattern j
  unsigned short a[8], b[128];
  for (int e = 0; e < 8; ++e) a[e] = 0x3C00;
  unsigned short pat[16] = {0x3C00, 0x0001, 0x0040, 0x03FF, 0x0400, 0x0000, 0x5400, 0x7C00, 0x7E00, 0x8001, 0x0200, 0x0100, 0x0010, 0x0002, 0x83FF, 0x7BFF};
  for (int j = 0; j < 16; ++j) for (int e = 0; e < 8; ++e) b[j * 8 + e] = pat[j];
  hipMemcpy(d_a, a, 16, hipMemcpyHostToDevice); hipMemcpy(d_b, b, 256, hipMemcpyHostToDevice);
  mm<<<1, 64>>>(d_a, d_b, d_out);
  float out[256]; hipMemcpy(out, d_out, 1024, hipMemcpyDeviceToHost);
  for (int j = 0; j < 16; ++j) printf("B pattern %04x (x32 summed with A = 1): D[0][%d] = %.6e\n", pat[j], j, out[j * 4]);
  return 0;
}
