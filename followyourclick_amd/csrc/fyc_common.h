// Shared device/host helpers for libfyc_hip.so (gfx950 only; wave64, no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/fyc.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define FYC_WAVE 64

// ---- error plumbing (host) ----------------------------------------------------------------
extern thread_local char g_fyc_err[512];
extern const void* g_fyc_zero_page;
// csrc/temporal_block_rr.hip (register-resident form of fyc_temporal_block, taken when the caller passes `wstream`)
int64_t fyc_temporal_block_rr_wstream_bytes();
int64_t fyc_temporal_block_rr_lds_bytes();
int fyc_temporal_block_rr_launch(const fyc_temporal_block_args* a, void* stream);
#ifdef FYC_TRACE
extern unsigned long long* g_fyc_trace;   // timing builds: device buffer for the GEMM kernels' s_memtime stamps (fyc_set_trace)
#endif
extern int g_fyc_tuning[16];  // [1] forced GEMM tile config, [2] forced ring depth, [3] attention variant, [4] GEMM tile-order strip width (-1 = row-major)

#define FYC_FAIL(code, ...)                                   \
  do {                                                        \
    snprintf(g_fyc_err, sizeof(g_fyc_err), __VA_ARGS__);      \
    return (code);                                            \
  } while (0)

#define FYC_REQUIRE(cond, ...)                                \
  do {                                                        \
    if (!(cond)) FYC_FAIL(-2, __VA_ARGS__);                   \
  } while (0)

#define FYC_CHECK_LAUNCH(name)                                                        \
  do {                                                                                \
    hipError_t _e = hipGetLastError();                                                \
    if (_e != hipSuccess) FYC_FAIL(-3, "%s launch failed: %s", name, hipGetErrorString(_e)); \
  } while (0)

// ---- dtype helpers (device) ---------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }
// round-to-nearest-even, NaN preserved (same rounding torch uses for float->bfloat16)
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
// two f32 -> packed bf16x2 in one v_cvt_pk_bf16_f32 (round-to-nearest-even in hardware)
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  f32x2 f = {a, b};
  bf16x2 h = __builtin_convertvector(f, bf16x2);
  return __builtin_bit_cast(unsigned, h);
}

typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

// The two 16-bit storage formats (FYC_BF16, FYC_F16) behind one interface: a pair of consecutive elements in a 32-bit word.
// bf16 unpacks with a shift / mask and packs with v_cvt_pk_bf16_f32; f16 unpacks with v_cvt_f32_f16 (+ SDWA for the upper half) and
// packs with v_cvt_pk_f16_f32 - one instruction per value or pair either way, round-to-nearest-even both.
template <typename T> struct Pair16;
template <> struct Pair16<bf16_t> {
  typedef bf16x8 Vec8;
  typedef bf16x4 Vec4;
  static constexpr unsigned short ONE = 0x3F80;
  __device__ static __forceinline__ float lo(unsigned u) { return __uint_as_float(u << 16); }
  __device__ static __forceinline__ float hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
  __device__ static __forceinline__ unsigned pack(float a, float b) { return pack_bf16x2(a, b); }
  __device__ static __forceinline__ float from_bits(unsigned short b) { return bf16_bits_to_f32(b); }
  __device__ static __forceinline__ unsigned short to_bits(float f) { return f32_to_bf16_bits(f); }
};
template <> struct Pair16<f16_t> {
  typedef f16x8 Vec8;
  typedef f16x4 Vec4;
  static constexpr unsigned short ONE = 0x3C00;
  __device__ static __forceinline__ float lo(unsigned u) { return (float)__builtin_bit_cast(f16x2, u)[0]; }
  __device__ static __forceinline__ float hi(unsigned u) { return (float)__builtin_bit_cast(f16x2, u)[1]; }
  __device__ static __forceinline__ unsigned pack(float a, float b) {
    f32x2 f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, f16x2));
  }
  __device__ static __forceinline__ float from_bits(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }
  __device__ static __forceinline__ unsigned short to_bits(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
};
// value of v after one rounding to T (identity for float)
template <typename T> __device__ __forceinline__ float round_through(float v) { return Pair16<T>::from_bits(Pair16<T>::to_bits(v)); }
template <> __device__ __forceinline__ float round_through<float>(float v) { return v; }

template <typename T> struct ElemIO;
template <> struct ElemIO<float> {
  static constexpr int VEC = 4;  // elements per 16-byte chunk
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
  // 4 consecutive elements
  __device__ static __forceinline__ void ld4(const float* p, float v[4]) {
    f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
  __device__ static __forceinline__ void st4(float* p, const float v[4]) {
    f32x4 t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = t;
  }
};
template <typename T> struct ElemIO16 {
  static constexpr int VEC = 8;
  __device__ static __forceinline__ float ld(const T* p) { return Pair16<T>::from_bits(*reinterpret_cast<const unsigned short*>(p)); }
  __device__ static __forceinline__ void st(T* p, float v) { *reinterpret_cast<unsigned short*>(p) = Pair16<T>::to_bits(v); }
  __device__ static __forceinline__ void ld4(const T* p, float v[4]) {
    u32x2 t = *reinterpret_cast<const u32x2*>(p);
    v[0] = Pair16<T>::lo(t[0]); v[1] = Pair16<T>::hi(t[0]);
    v[2] = Pair16<T>::lo(t[1]); v[3] = Pair16<T>::hi(t[1]);
  }
  __device__ static __forceinline__ void st4(T* p, const float v[4]) {
    u32x2 t;
    t[0] = Pair16<T>::pack(v[0], v[1]);
    t[1] = Pair16<T>::pack(v[2], v[3]);
    *reinterpret_cast<u32x2*>(p) = t;
  }
};
template <> struct ElemIO<bf16_t> : ElemIO16<bf16_t> {};
template <> struct ElemIO<f16_t> : ElemIO16<f16_t> {};

// 8 consecutive elements as floats (two 16-B loads for f32, one for bf16)
template <typename T> __device__ __forceinline__ void load8(const T* p, float v[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float v[8]) {
  ElemIO<float>::ld4(p, v); ElemIO<float>::ld4(p + 4, v + 4);
}
template <typename T> __device__ __forceinline__ void load8_16(const T* p, float v[8]) {
  u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = Pair16<T>::lo(t[i]);
    v[2 * i + 1] = Pair16<T>::hi(t[i]);
  }
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float v[8]) { load8_16<bf16_t>(p, v); }
template <> __device__ __forceinline__ void load8<f16_t>(const f16_t* p, float v[8]) { load8_16<f16_t>(p, v); }
template <typename T> __device__ __forceinline__ void store8(T* p, const float v[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float v[8]) {
  ElemIO<float>::st4(p, v); ElemIO<float>::st4(p + 4, v + 4);
}
template <typename T> __device__ __forceinline__ void store8_16(T* p, const float v[8]) {
  u32x4 t;
#pragma unroll
  for (int i = 0; i < 4; ++i) t[i] = Pair16<T>::pack(v[2 * i], v[2 * i + 1]);
  *reinterpret_cast<u32x4*>(p) = t;
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float v[8]) { store8_16<bf16_t>(p, v); }
template <> __device__ __forceinline__ void store8<f16_t>(f16_t* p, const float v[8]) { store8_16<f16_t>(p, v); }

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// exact-erf GELU, x * Phi(x) (F.gelu default, reference diffusers/models/attention.py:815), with erfc from
// Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, no cancellation on the negative side): ~12 VALU ops
// instead of ~40 for erff() - the GEGLU epilogue was 24 % of the K=320 FF GEMM (profiles/r01_gemm_epilogue_ablation.txt)
__device__ __forceinline__ float gelu_erf_f(float g) {
  const float z = fabsf(g) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  float poly = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  poly = __builtin_fmaf(poly, t, 1.421413741f);
  poly = __builtin_fmaf(poly, t, -0.284496736f);
  poly = __builtin_fmaf(poly, t, 0.254829592f);
  const float e = poly * t * __expf(-z * z);      // erfc(z)
  return 0.5f * g * (g >= 0.f ? 2.0f - e : e);
}

// GEGLU gate for the bf16 production path: h * gelu(g) on value pairs with packed f32 VALU (v_pk_fma_f32 / v_pk_mul_f32).
// erf(x / sqrt2) = x * R(x^2) with the minimax polynomial R of degree 7 on |x| <= 4.2 (constrained to reach exactly 1 at the
// clamp; max |error| of x * Phi(x) over all x: 9e-5, 20x below half a bf16 ulp at unit scale) - no v_rcp / v_exp, which are
// quarter rate: 80 gates per lane and tile were ~5 us of the ~24 us a 256 x 320 tile of the K = 320 FF1 layer takes.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// the same gate on one value with scalar VALU (for kernels that run one wave per SIMD, where packed f32 beside MFMAs is slow)
__device__ __forceinline__ float geglu_one(float h, float g) {
  const float xc = __builtin_amdgcn_fmed3f(g, -4.2f, 4.2f);
  const float u = xc * xc;
  float r = -1.803605736e-09f;
  r = __builtin_fmaf(r, u, 1.588336848e-07f);
  r = __builtin_fmaf(r, u, -6.076054488e-06f);
  r = __builtin_fmaf(r, u, 1.337840930e-04f);
  r = __builtin_fmaf(r, u, -1.901335453e-03f);
  r = __builtin_fmaf(r, u, 1.859653848e-02f);
  r = __builtin_fmaf(r, u, -1.310565435e-01f);
  r = __builtin_fmaf(r, u, 7.969318188e-01f);
  const float phi = __builtin_fmaf(xc * r, 0.5f, 0.5f);       // Phi(g)
  return h * (g * phi);
}
__device__ __forceinline__ f32x2 geglu_pair(f32x2 h, f32x2 g) {
  const f32x2 xc = {__builtin_amdgcn_fmed3f(g.x, -4.2f, 4.2f), __builtin_amdgcn_fmed3f(g.y, -4.2f, 4.2f)};
  const f32x2 u = xc * xc;
  f32x2 r = (f32x2){-1.803605736e-09f, -1.803605736e-09f};
  r = __builtin_elementwise_fma(r, u, (f32x2){1.588336848e-07f, 1.588336848e-07f});
  r = __builtin_elementwise_fma(r, u, (f32x2){-6.076054488e-06f, -6.076054488e-06f});
  r = __builtin_elementwise_fma(r, u, (f32x2){1.337840930e-04f, 1.337840930e-04f});
  r = __builtin_elementwise_fma(r, u, (f32x2){-1.901335453e-03f, -1.901335453e-03f});
  r = __builtin_elementwise_fma(r, u, (f32x2){1.859653848e-02f, 1.859653848e-02f});
  r = __builtin_elementwise_fma(r, u, (f32x2){-1.310565435e-01f, -1.310565435e-01f});
  r = __builtin_elementwise_fma(r, u, (f32x2){7.969318188e-01f, 7.969318188e-01f});
  const f32x2 half = {0.5f, 0.5f};
  const f32x2 phi = __builtin_elementwise_fma(xc * r, half, half);     // Phi(g)
  return h * (g * phi);
}

// Maximum over the two lane halves (swap32) / over lane rows 16 apart (swap16) of a wave by gfx950's v_permlane*_swap: plain VALU, no
// LDS queue.  The empty asm is REQUIRED: hipcc 7.2 simplifies fmaxf(r[0], r[1]) of __builtin_amdgcn_permlane*_swap(x, x, ...) to r[0]
// (the optimised IR keeps extractvalue 0 only - it takes the two results of a swap of equal operands for equal values), so the
// "maximum" silently covered half of the lanes.  Found in round 4 when the f16 attention overflowed: its running max ignored the keys
// of lanes 32..63, the lazy rescale never fired for them and a late score 2^16 above the max became an infinite probability (bf16's
// exponent range had hidden the same defect since round 1).  SUMS ARE AFFECTED THE SAME WAY (round-4 advice): with the swap's operand
// held in one `unsigned` hipcc 7.2 emits `v_permlane32_swap v1, v2; v_add_f32 v1, v1, v1` = 2 * r[0] - the other lane half is dropped;
// the row-panel kernels' sums were correct only because they happened to call __float_as_uint(v) twice.  swap32_sum / swap16_sum below
// carry the same barrier; tests/test_build_checks.py disassembles both forms.
__device__ __forceinline__ float swap32_max(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  unsigned a = r[0], b = r[1];
  asm volatile("" : "+v"(a), "+v"(b));
  return fmaxf(__uint_as_float(a), __uint_as_float(b));
}
__device__ __forceinline__ float swap16_max(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  unsigned a = r[0], b = r[1];
  asm volatile("" : "+v"(a), "+v"(b));
  return fmaxf(__uint_as_float(a), __uint_as_float(b));
}

__device__ __forceinline__ float swap32_sum(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  unsigned a = r[0], b = r[1];
  asm volatile("" : "+v"(a), "+v"(b));
  return __uint_as_float(a) + __uint_as_float(b);
}
__device__ __forceinline__ float swap16_sum(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  unsigned a = r[0], b = r[1];
  asm volatile("" : "+v"(a), "+v"(b));
  return __uint_as_float(a) + __uint_as_float(b);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
