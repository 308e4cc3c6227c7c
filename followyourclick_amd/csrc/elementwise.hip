// Small HBM-bound kernels around the UNet/VAE: layout changes at the API boundary
// ((b,c,f,h,w) f32 <-> channels-last), channel concat, guidance + DDIM update, casts.
#include "fyc_common.h"

namespace {

template <typename T>
__global__ void __launch_bounds__(256) concat_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y,
                                                     long long chunks, int c1_8, int c2_8) {
  const int ct8 = c1_8 + c2_8;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < chunks; idx += (long long)gridDim.x * 256) {
    const long long row = idx / ct8;
    const int c = (int)(idx - row * ct8);
    const u32x4* src = (c < c1_8) ? reinterpret_cast<const u32x4*>(a) + (row * c1_8 + c) * (int)(sizeof(T) * 8 / 16)
                                  : reinterpret_cast<const u32x4*>(b) + (row * c2_8 + (c - c1_8)) * (int)(sizeof(T) * 8 / 16);
    u32x4* dst = reinterpret_cast<u32x4*>(y) + idx * (int)(sizeof(T) * 8 / 16);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(T) * 8 / 16); ++k) dst[k] = src[k];
  }
}

__global__ void __launch_bounds__(256) silu_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = silu_f(x[i]);
}

template <typename T>
__global__ void __launch_bounds__(256) cast_from_kernel(const float* __restrict__ x, T* __restrict__ y, long long rows, int cols, int ld) {
  const long long n = rows * ld;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long r = i / ld;
    const int c = (int)(i - r * ld);
    ElemIO<T>::st(y + i, c < cols ? x[r * cols + c] : 0.f);
  }
}
template <typename T>
__global__ void __launch_bounds__(256) cast_to_kernel(const T* __restrict__ x, float* __restrict__ y, long long rows, int cols, int ld) {
  const long long n = rows * cols;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long r = i / cols;
    const int c = (int)(i - r * cols);
    y[i] = ElemIO<T>::ld(x + r * ld + c);
  }
}

// one thread per output pixel-row (frame, pixel): writes c_pad channels
template <typename T>
__global__ void __launch_bounds__(256) unet_input_kernel(const float* __restrict__ lat, const float* __restrict__ mask,
                                                         const float* __restrict__ first, T* __restrict__ x, int B, int F,
                                                         int HW, int CL, int c_pad, int cfg_dup, int mask_frames, int mode) {
  const long long total = (long long)cfg_dup * B * F * HW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int p = (int)(i % HW);
    const long long bf = i / HW;
    const int f = (int)(bf % F);
    const int b = (int)((bf / F) % B);  // CFG duplicates share the same latents
    T* o = x + i * c_pad;
    for (int c = 0; c < CL; ++c) ElemIO<T>::st(o + c, lat[(((long long)b * CL + c) * F + f) * HW + p]);
    if (mode == 1) {          // use_first_frame_condition_concat: the clean first-frame latents beside the latents of EVERY frame
      for (int c = 0; c < CL; ++c) ElemIO<T>::st(o + CL + c, first ? first[((long long)b * CL + c) * HW + p] : 0.f);
      for (int c = 2 * CL; c < c_pad; ++c) ElemIO<T>::st(o + c, 0.f);
      continue;
    }
    float m;
    if (mask) {
      const int mf = mask_frames > 1 ? f : 0;
      m = fminf(fmaxf(mask[((long long)b * mask_frames + mf) * HW + p], 0.f), 1.f);
    } else {
      m = (f == 0) ? 1.f : 0.f;
    }
    ElemIO<T>::st(o + CL, m);
    for (int c = 0; c < CL; ++c)
      ElemIO<T>::st(o + CL + 1 + c, (f == 0 && first) ? first[((long long)b * CL + c) * HW + p] : 0.f);
    for (int c = 2 * CL + 1; c < c_pad; ++c) ElemIO<T>::st(o + c, 0.f);
  }
}

template <typename T>
__global__ void __launch_bounds__(256) cfg_ddim_kernel(const T* __restrict__ pred, float* __restrict__ lat,
                                                       const float* __restrict__ coef, int B, int F, int HW, int CL, int ld,
                                                       int cfg, float guidance, int pred_type, int clip,
                                                       const T* __restrict__ single, float video_scale,
                                                       const float* __restrict__ noise, float sigma, int reclip) {
  const float sa = coef[0], sb = coef[1], sap = coef[2], sbp = coef[3];
  const long long total = (long long)B * F * HW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int p = (int)(i % HW);
    const long long bf = i / HW;
    const int f = (int)(bf % F), b = (int)(bf / F);
    const T* pu = pred + i * ld;                                   // uncond (or the only) half
    const T* pc = pred + ((long long)B * F * HW + i) * ld;         // cond half
    for (int c = 0; c < CL; ++c) {
      float v = ElemIO<T>::ld(pu + c);
      if (cfg) {
        const float u = v, cnd = ElemIO<T>::ld(pc + c);
        if (single) { const float s1 = ElemIO<T>::ld(single + i * ld + c); v = s1 + video_scale * (u - s1) + guidance * (cnd - u); }
        else v = u + guidance * (cnd - u);
      }
      float* lp = lat + (((long long)b * CL + c) * F + f) * HW + p;
      const float x = *lp;
      float x0, eps;
      if (pred_type == 1) { x0 = sa * x - sb * v; eps = sa * v + sb * x; }
      else if (pred_type == 0) { x0 = (x - sb * v) / sa; eps = v; }
      else { x0 = v; eps = v; }
      if (clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
      if (reclip) eps = (x - sa * x0) / sb;                        // use_clipped_model_output (scheduling_ddim.py:342-344)
      float nx = sap * x0 + sbp * eps;
      if (noise) nx += sigma * noise[(((long long)b * CL + c) * F + f) * HW + p];      // eta > 0 (:346-363)
      *lp = nx;
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* __restrict__ z, T* __restrict__ x, int N, int C, int HW,
                                                           int c_pad, float scale) {
  const long long total = (long long)N * HW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int p = (int)(i % HW);
    const long long n = i / HW;
    T* o = x + i * c_pad;
    for (int c = 0; c < C; ++c) ElemIO<T>::st(o + c, z[(n * C + c) * HW + p] * scale);
    for (int c = C; c < c_pad; ++c) ElemIO<T>::st(o + c, 0.f);
  }
}

template <typename T>
__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ y, int N, int C, int HW,
                                                           int ld, float mul, float add, float lo, float hi) {
  const long long total = (long long)N * C * HW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int p = (int)(i % HW);
    const long long nc = i / HW;
    const int c = (int)(nc % C);
    const long long n = nc / C;
    const float v = ElemIO<T>::ld(x + (n * HW + p) * ld + c) * mul + add;
    y[i] = fminf(fmaxf(v, lo), hi);
  }
}

// (O, I, 3, 3) f32 -> [O][slab][tap][c in slab] (see include/fyc.h): one thread per output element
template <typename T>
__global__ void __launch_bounds__(256) pack_conv3x3_kernel(const float* __restrict__ w, T* __restrict__ out, int O, int I, int Ip, int slab) {
  const long long total = (long long)O * 9 * Ip;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int k = (int)(i % (9 * Ip));
    const long long o = i / (9 * Ip);
    const int sl = k / (9 * slab), r = k - sl * 9 * slab, tap = r / slab, c = sl * slab + (r - tap * slab);
    ElemIO<T>::st(out + i, c < I ? w[(o * I + c) * 9 + tap] : 0.f);
  }
}

// value rows [0, O/2) and gate rows [O/2, O) interleaved in blocks of 16 (weight and bias)
template <typename T>
__global__ void __launch_bounds__(256) pack_geglu_kernel(const float* __restrict__ w, const float* __restrict__ b, T* __restrict__ wo,
                                                         float* __restrict__ bo, int O, int I) {
  const long long total = (long long)O * I;
  const int half = O / 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int col = (int)(i % I), row = (int)(i / I);
    const int blk = row >> 5, in = row & 31;
    const int src = (in < 16) ? blk * 16 + in : half + blk * 16 + (in - 16);
    ElemIO<T>::st(wo + i, w[(long long)src * I + col]);
    if (col == 0 && b != nullptr) bo[row] = b[src];
  }
}

inline int grid_for(long long n) {
  long long b = ceil_div64(n, 256);
  return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}


// out[r][c8..] = table[ids[r]][c8..] + pos[r % seq][c8..]: one thread per 8 channels
template <typename T>
__global__ void __launch_bounds__(256) embed_tokens_kernel(const long long* __restrict__ ids, const float* __restrict__ table,
                                                           const float* __restrict__ pos, T* __restrict__ out, long long rows,
                                                           int seq, int C, int vocab) {
  const int C8 = C >> 3;
  const long long total = rows * C8;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / C8;
    const int c = (int)(i - r * C8) * 8;
    long long id = ids[r];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);   // host validates ids; clamp keeps a bad id from reading out of bounds
    const float* t = table + id * C + c;
    const float* pp = pos + (long long)(r % seq) * C + c;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = t[e] + pp[e];
    store8<T>(out + r * C + c, v);
  }
}

// one thread per output element; reads are P-contiguous runs of the image rows
template <typename T>
__global__ void __launch_bounds__(256) patchify_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int Cin, int H,
                                                       int W, int P, int ld) {
  const int gh = H / P, gw = W / P, K = Cin * P * P;
  const long long total = (long long)B * gh * gw * ld;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int k = (int)(i % ld);
    const long long patch = i / ld;
    float v = 0.f;
    if (k < K) {
      const int px = k % P, py = (k / P) % P, c = k / (P * P);
      const int gx = (int)(patch % gw), gy = (int)((patch / gw) % gh), b = (int)(patch / ((long long)gw * gh));
      v = img[(((long long)b * Cin + c) * H + gy * P + py) * W + gx * P + px];
    }
    ElemIO<T>::st(out + i, v);
  }
}

}  // namespace

#define FYC_DT(a, call_bf16, call_f16, call_f32)               \
  do {                                                         \
    if ((a)->dtype == FYC_BF16) { call_bf16; }                 \
    else if ((a)->dtype == FYC_F16) { call_f16; }              \
    else if ((a)->dtype == FYC_F32) { call_f32; }              \
    else FYC_FAIL(-2, "bad dtype %d", (a)->dtype);             \
  } while (0)

extern "C" int fyc_concat_channels(const fyc_concat_args* a, void* stream) {
  FYC_REQUIRE(a && a->a && a->b && a->y, "fyc_concat_channels: null pointer");
  FYC_REQUIRE(a->c1 % 8 == 0 && a->c2 % 8 == 0 && a->rows > 0, "fyc_concat_channels: channels must be multiples of 8");
  hipStream_t st = (hipStream_t)stream;
  const long long chunks = a->rows * ((a->c1 + a->c2) / 8);
  FYC_DT(a,
         hipLaunchKernelGGL(concat_kernel<bf16_t>, dim3(grid_for(chunks)), dim3(256), 0, st, (const bf16_t*)a->a, (const bf16_t*)a->b, (bf16_t*)a->y, chunks, a->c1 / 8, a->c2 / 8),
         hipLaunchKernelGGL(concat_kernel<f16_t>, dim3(grid_for(chunks)), dim3(256), 0, st, (const f16_t*)a->a, (const f16_t*)a->b, (f16_t*)a->y, chunks, a->c1 / 8, a->c2 / 8),
         hipLaunchKernelGGL(concat_kernel<float>, dim3(grid_for(chunks)), dim3(256), 0, st, (const float*)a->a, (const float*)a->b, (float*)a->y, chunks, a->c1 / 8, a->c2 / 8));
  FYC_CHECK_LAUNCH("fyc_concat_channels");
  return 0;
}

extern "C" int fyc_silu_f32(const fyc_silu_args* a, void* stream) {
  FYC_REQUIRE(a && a->x && a->y && a->n > 0, "fyc_silu_f32: bad args");
  hipLaunchKernelGGL(silu_kernel, dim3(grid_for(a->n)), dim3(256), 0, (hipStream_t)stream, a->x, a->y, (long long)a->n);
  FYC_CHECK_LAUNCH("fyc_silu_f32");
  return 0;
}

extern "C" int fyc_cast_from_f32(const fyc_cast_args* a, void* stream) {
  FYC_REQUIRE(a && a->x && a->y && a->rows > 0 && a->cols > 0 && a->ld >= a->cols, "fyc_cast_from_f32: bad args");
  hipStream_t st = (hipStream_t)stream;
  const long long n = a->rows * a->ld;
  FYC_DT(a,
         hipLaunchKernelGGL(cast_from_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, a->x, (bf16_t*)a->y, (long long)a->rows, a->cols, a->ld),
         hipLaunchKernelGGL(cast_from_kernel<f16_t>, dim3(grid_for(n)), dim3(256), 0, st, a->x, (f16_t*)a->y, (long long)a->rows, a->cols, a->ld),
         hipLaunchKernelGGL(cast_from_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, a->x, (float*)a->y, (long long)a->rows, a->cols, a->ld));
  FYC_CHECK_LAUNCH("fyc_cast_from_f32");
  return 0;
}

extern "C" int fyc_cast_to_f32(const fyc_cast_to_args* a, void* stream) {
  FYC_REQUIRE(a && a->x && a->y && a->rows > 0 && a->cols > 0 && a->ld >= a->cols, "fyc_cast_to_f32: bad args");
  hipStream_t st = (hipStream_t)stream;
  const long long n = a->rows * a->cols;
  FYC_DT(a,
         hipLaunchKernelGGL(cast_to_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, (const bf16_t*)a->x, a->y, (long long)a->rows, a->cols, a->ld),
         hipLaunchKernelGGL(cast_to_kernel<f16_t>, dim3(grid_for(n)), dim3(256), 0, st, (const f16_t*)a->x, a->y, (long long)a->rows, a->cols, a->ld),
         hipLaunchKernelGGL(cast_to_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, (const float*)a->x, a->y, (long long)a->rows, a->cols, a->ld));
  FYC_CHECK_LAUNCH("fyc_cast_to_f32");
  return 0;
}

extern "C" int fyc_unet_input(const fyc_unet_input_args* a, void* stream) {
  FYC_REQUIRE(a && a->latents && a->x, "fyc_unet_input: null pointer");
  FYC_REQUIRE(a->B > 0 && a->F > 0 && a->HW > 0 && a->c_latent > 0 && a->c_pad >= 2 * a->c_latent + (a->mode == 1 ? 0 : 1), "fyc_unet_input: bad dims");
  FYC_REQUIRE(a->mode == 0 || a->mode == 1, "fyc_unet_input: mode %d", a->mode);
  FYC_REQUIRE(a->cfg_dup == 1 || a->cfg_dup == 2, "fyc_unet_input: cfg_dup must be 1 or 2");
  FYC_REQUIRE(a->mask == nullptr || a->mask_frames == 1 || a->mask_frames == a->F, "fyc_unet_input: mask_frames must be 1 or F");
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)a->cfg_dup * a->B * a->F * a->HW;
  const int mf = a->mask ? a->mask_frames : 1;
  FYC_DT(a,
         hipLaunchKernelGGL(unet_input_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, a->latents, a->mask, a->first, (bf16_t*)a->x, a->B, a->F, a->HW, a->c_latent, a->c_pad, a->cfg_dup, mf, a->mode),
         hipLaunchKernelGGL(unet_input_kernel<f16_t>, dim3(grid_for(n)), dim3(256), 0, st, a->latents, a->mask, a->first, (f16_t*)a->x, a->B, a->F, a->HW, a->c_latent, a->c_pad, a->cfg_dup, mf, a->mode),
         hipLaunchKernelGGL(unet_input_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, a->latents, a->mask, a->first, (float*)a->x, a->B, a->F, a->HW, a->c_latent, a->c_pad, a->cfg_dup, mf, a->mode));
  FYC_CHECK_LAUNCH("fyc_unet_input");
  return 0;
}

extern "C" int fyc_cfg_ddim_step(const fyc_cfg_ddim_args* a, void* stream) {
  FYC_REQUIRE(a && a->pred && a->latents && a->coef, "fyc_cfg_ddim_step: null pointer");
  FYC_REQUIRE(a->B > 0 && a->F > 0 && a->HW > 0 && a->c_latent > 0 && a->ld >= a->c_latent, "fyc_cfg_ddim_step: bad dims");
  FYC_REQUIRE(a->pred_type >= 0 && a->pred_type <= 2, "fyc_cfg_ddim_step: pred_type %d", a->pred_type);
  FYC_REQUIRE(a->pred_single == nullptr || a->cfg, "fyc_cfg_ddim_step: pred_single needs classifier-free guidance (cfg = 1)");
  FYC_REQUIRE(a->variance_noise == nullptr || a->sigma >= 0.f, "fyc_cfg_ddim_step: sigma %g", (double)a->sigma);
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)a->B * a->F * a->HW;
  FYC_DT(a,
         hipLaunchKernelGGL(cfg_ddim_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, (const bf16_t*)a->pred, a->latents, a->coef, a->B, a->F, a->HW, a->c_latent, a->ld, a->cfg, a->guidance, a->pred_type, a->clip_sample, (const bf16_t*)a->pred_single, a->video_scale, a->variance_noise, a->sigma, a->clipped_model_output),
         hipLaunchKernelGGL(cfg_ddim_kernel<f16_t>, dim3(grid_for(n)), dim3(256), 0, st, (const f16_t*)a->pred, a->latents, a->coef, a->B, a->F, a->HW, a->c_latent, a->ld, a->cfg, a->guidance, a->pred_type, a->clip_sample, (const f16_t*)a->pred_single, a->video_scale, a->variance_noise, a->sigma, a->clipped_model_output),
         hipLaunchKernelGGL(cfg_ddim_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, (const float*)a->pred, a->latents, a->coef, a->B, a->F, a->HW, a->c_latent, a->ld, a->cfg, a->guidance, a->pred_type, a->clip_sample, (const float*)a->pred_single, a->video_scale, a->variance_noise, a->sigma, a->clipped_model_output));
  FYC_CHECK_LAUNCH("fyc_cfg_ddim_step");
  return 0;
}

extern "C" int fyc_nchw_to_nhwc(const fyc_nchw_in_args* a, void* stream) {
  FYC_REQUIRE(a && a->z && a->x && a->N > 0 && a->C > 0 && a->HW > 0 && a->c_pad >= a->C, "fyc_nchw_to_nhwc: bad args");
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)a->N * a->HW;
  FYC_DT(a,
         hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, a->z, (bf16_t*)a->x, a->N, a->C, a->HW, a->c_pad, a->scale),
         hipLaunchKernelGGL(nchw_to_nhwc_kernel<f16_t>, dim3(grid_for(n)), dim3(256), 0, st, a->z, (f16_t*)a->x, a->N, a->C, a->HW, a->c_pad, a->scale),
         hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, a->z, (float*)a->x, a->N, a->C, a->HW, a->c_pad, a->scale));
  FYC_CHECK_LAUNCH("fyc_nchw_to_nhwc");
  return 0;
}

extern "C" int fyc_nhwc_to_nchw(const fyc_nhwc_out_args* a, void* stream) {
  FYC_REQUIRE(a && a->x && a->y && a->N > 0 && a->C > 0 && a->HW > 0 && a->ld >= a->C, "fyc_nhwc_to_nchw: bad args");
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)a->N * a->C * a->HW;
  FYC_DT(a,
         hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, (const bf16_t*)a->x, a->y, a->N, a->C, a->HW, a->ld, a->mul, a->add, a->lo, a->hi),
         hipLaunchKernelGGL(nhwc_to_nchw_kernel<f16_t>, dim3(grid_for(n)), dim3(256), 0, st, (const f16_t*)a->x, a->y, a->N, a->C, a->HW, a->ld, a->mul, a->add, a->lo, a->hi),
         hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, (const float*)a->x, a->y, a->N, a->C, a->HW, a->ld, a->mul, a->add, a->lo, a->hi));
  FYC_CHECK_LAUNCH("fyc_nhwc_to_nchw");
  return 0;
}

extern "C" int fyc_embed_tokens(const fyc_embed_args* a, void* stream) {
  FYC_REQUIRE(a && a->ids && a->table && a->pos && a->out, "fyc_embed_tokens: null pointer");
  FYC_REQUIRE(a->rows > 0 && a->seq > 0 && a->vocab > 0 && a->C > 0 && a->C % 8 == 0, "fyc_embed_tokens: bad dims rows=%lld seq=%d C=%d", (long long)a->rows, a->seq, a->C);
  hipStream_t st = (hipStream_t)stream;
  const long long n = a->rows * (a->C / 8);
  FYC_DT(a,
         hipLaunchKernelGGL(embed_tokens_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, (const long long*)a->ids, a->table, a->pos, (bf16_t*)a->out, (long long)a->rows, a->seq, a->C, a->vocab),
         hipLaunchKernelGGL(embed_tokens_kernel<f16_t>, dim3(grid_for(n)), dim3(256), 0, st, (const long long*)a->ids, a->table, a->pos, (f16_t*)a->out, (long long)a->rows, a->seq, a->C, a->vocab),
         hipLaunchKernelGGL(embed_tokens_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, (const long long*)a->ids, a->table, a->pos, (float*)a->out, (long long)a->rows, a->seq, a->C, a->vocab));
  FYC_CHECK_LAUNCH("fyc_embed_tokens");
  return 0;
}

extern "C" int fyc_patchify(const fyc_patchify_args* a, void* stream) {
  FYC_REQUIRE(a && a->image && a->out, "fyc_patchify: null pointer");
  FYC_REQUIRE(a->B > 0 && a->Cin > 0 && a->P > 0 && a->H > 0 && a->W > 0 && a->H % a->P == 0 && a->W % a->P == 0,
              "fyc_patchify: image %dx%d is not a multiple of the patch size %d", a->H, a->W, a->P);
  FYC_REQUIRE(a->ld >= a->Cin * a->P * a->P, "fyc_patchify: ld=%d < Cin*P*P", a->ld);
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)a->B * (a->H / a->P) * (a->W / a->P) * a->ld;
  FYC_DT(a,
         hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, a->image, (bf16_t*)a->out, a->B, a->Cin, a->H, a->W, a->P, a->ld),
         hipLaunchKernelGGL(patchify_kernel<f16_t>, dim3(grid_for(n)), dim3(256), 0, st, a->image, (f16_t*)a->out, a->B, a->Cin, a->H, a->W, a->P, a->ld),
         hipLaunchKernelGGL(patchify_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, a->image, (float*)a->out, a->B, a->Cin, a->H, a->W, a->P, a->ld));
  FYC_CHECK_LAUNCH("fyc_patchify");
  return 0;
}

extern "C" int fyc_pack_conv3x3(const fyc_pack_conv3x3_args* a, void* stream) {
  FYC_REQUIRE(a && a->w && a->out && a->O > 0 && a->I > 0, "fyc_pack_conv3x3: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int Ip = (a->I + 63) / 64 * 64;
  const long long n = (long long)a->O * 9 * Ip;
  FYC_DT(a,
         hipLaunchKernelGGL(pack_conv3x3_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, a->w, (bf16_t*)a->out, a->O, a->I, Ip, 64),
         hipLaunchKernelGGL(pack_conv3x3_kernel<f16_t>, dim3(grid_for(n)), dim3(256), 0, st, a->w, (f16_t*)a->out, a->O, a->I, Ip, 64),
         hipLaunchKernelGGL(pack_conv3x3_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, a->w, (float*)a->out, a->O, a->I, Ip, 32));
  FYC_CHECK_LAUNCH("fyc_pack_conv3x3");
  return 0;
}

extern "C" int fyc_pack_geglu(const fyc_pack_geglu_args* a, void* stream) {
  FYC_REQUIRE(a && a->w && a->w_out && a->O > 0 && a->I > 0, "fyc_pack_geglu: bad args");
  FYC_REQUIRE(a->O % 32 == 0, "fyc_pack_geglu: O=%d must be a multiple of 32 (16 value + 16 gate rows per block)", a->O);
  FYC_REQUIRE((a->b == nullptr) == (a->b_out == nullptr), "fyc_pack_geglu: bias in and out must both be given or both be null");
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)a->O * a->I;
  FYC_DT(a,
         hipLaunchKernelGGL(pack_geglu_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, a->w, a->b, (bf16_t*)a->w_out, a->b_out, a->O, a->I),
         hipLaunchKernelGGL(pack_geglu_kernel<f16_t>, dim3(grid_for(n)), dim3(256), 0, st, a->w, a->b, (f16_t*)a->w_out, a->b_out, a->O, a->I),
         hipLaunchKernelGGL(pack_geglu_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, a->w, a->b, (float*)a->w_out, a->b_out, a->O, a->I));
  FYC_CHECK_LAUNCH("fyc_pack_geglu");
  return 0;
}
