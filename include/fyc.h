/*
 * fyc.h - C ABI of libfyc_hip.so, the MI355X (gfx950) compute library behind
 * FollowYourClick's denoising hot path (AnimationPipeline.__call__ -> UNet3DConditionModel.forward
 * -> DDIMScheduler.step -> vae.decode).
 *
 * The reference is 100 % Python and has no FFI of its own, so there is nothing
 * to bind one-to-one; each entry point below names the reference op sequence (file:line under
 * /root/reference) it replaces.  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types.
 *  - every op: int fyc_<op>(const fyc_<op>_args*, void* hip_stream); 0 = ok, <0 = error,
 *    fyc_last_error() returns the thread-local message.  Never throws, never exits.
 *  - the callee never allocates, frees or synchronises: all tensors/workspaces are raw device
 *    pointers owned by the caller; work is enqueued on `hip_stream` (hipGraph-capturable).
 *  - activations are channels-last: [frames = B*F][H][W][C] (== token-major [rows][C]), element type
 *    `dtype` (FYC_BF16 production, FYC_F16 = the reference's deployed fp16-autocast precision class, FYC_F32 parity mode); statistics, biases, norm affine
 *    parameters, time-embedding rows and latents are always f32.
 */
#ifndef FYC_H
#define FYC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version = major * 100 + minor.  A host compiled against this header MUST compare fyc_version() with FYC_VERSION before its first
 * call and refuse a library whose MAJOR differs: argument structs grow at the end between majors (round 3 appended `wstream` to
 * fyc_temporal_block_args and widened the tuning table to 16 keys without bumping the number: a round-2 host would have passed a short
 * struct whose missing tail the library reads as a pointer).  History: 100 = rounds 1-3 (see above), 200 = round 4, 201 = FYC_F16, 300 = round 5: fields were APPENDED to two argument structs - variance_noise / sigma / clipped_model_output behind the DDIM step's arguments, mode behind the UNet input's - a struct change, hence a new major: a 2xx host passes shorter structs whose missing tail this library would read; 301 = round 6, no struct change: fyc_gemm accepts chan_parts for problems it runs split-K (fyc_gemm_workspace_bytes() > 0; the finish kernel writes them in 128-row tiles, fyc_gemm_stat_layout answers for it), tile config 11, tuning keys 10-13. */
#define FYC_VERSION 301

/* FYC_F16 (minor version 1): IEEE half storage with f32 accumulation - every op that takes FYC_BF16 takes it, same layouts, same
 * packed weight streams (16-bit elements), v_mfma_*_f16 instead of v_mfma_*_bf16; the packers cast to the `dtype` they are given */
typedef enum { FYC_F32 = 0, FYC_BF16 = 1, FYC_F16 = 2 } fyc_dtype;

/* ---- library ------------------------------------------------------------------------- */
int fyc_version(void);
const char* fyc_last_error(void);
/* zero_page: >= 256 bytes of zeroed device memory owned by the caller for the life of the process
 * (source for out-of-range / padding lanes of the direct-to-LDS loaders). */
int fyc_init(const void* zero_page);
/* fills caps[0..7]: CU count, LDS bytes/CU, wave size, gfx arch number (950), clock kHz, L2 bytes, 0, 0 */
int fyc_device_caps(int64_t* caps);
/* tuning knobs for A/B measurements (0 = automatic): key 0 = 1 disables split-K, key 1 = GEMM tile config (1: 128x128/4 waves, 2: 128x64/4,
 * 3: 256x128/8, 4: 256x64/4, 5: 256x320/8, 6: 128x320/8, 7: 256x256/8, 8/10: 128x320 / 128x128 with 64-byte K tiles, 11: 128x160/4 waves, two workgroups per CU), key 2 = GEMM LDS ring depth (2..4), key 3 = attention kernel variant,
 * key 4 = column-strip width of the GEMM tile order (-1: row-major), key 5 = 1 disables the wave-role stagger of the 8-wave GEMM tiles, key 6 = 1 disables the LDS-staged wide epilogues, key 7 = 1 only the wide head-split one,
 * keys 8 / 9 only act in a library built with FYC_GEMM_VARIANTS=1 (tools/exp/gemm_variants/: the round-4 main-loop experiments, measured slower,
 * not part of the product library): key 8 = 1: no s_setprio around the MFMA phases of the ping-pong loop; key 9 = 2: the ping-pong loop (tile
 * configs 21 / 22 / 23) wherever it is built, 3: the overlapped-epilogue kernel (config 31).  In the product library key 9 is ignored and a
 * request for one of those tile configs runs its one-phase twin (5 / 6 / 7 / 6): the one-phase loop always;
 * key 10 = v > 0: split-K for M <= 4096 keeps at least v K tiles per slice (default 16) and starts at K >= 128 v (default 2048);
 * key 11 = v > 0: every other GEMM block of an XCD starts v x 1024 cycles late (phase shift between the CUs' epilogues, A/B);
 * key 12 = 1: the GEMM epilogues load their per-row / per-column inputs themselves instead of finding them pre-staged in LDS (A/B);
 * key 13 = 1: the generic pass 1 of the packed LINEAR epilogue instead of its specialised copies (A/B);
 * key 14 = 1: the 32x32x16 matrix instruction in the K loop of the 256-row tiles (tile configs 12 / 13 / 14 = twins of 5 / 7 / 3) - only in a
 * library built with -DFYC_GEMM_MI32 (measured 8-12 % slower, not part of the product library); key 15 = 1 / 2: the pack-first head-split
 * epilogue on / off for every problem (default: M <= 8192 only) */
int fyc_set_tuning(int key, int value);

/* ---- GEMM / implicit-GEMM convolution --------------------------------------------------
 * out[m][n] = ( sum_k A[m][k] * W[n][k] + bias[n] + rowbias[m / rows_per_batch][n] + residual[m][n] ) * out_scale
 * Replaces: nn.Linear / 1x1 Conv2d everywhere on the path (diffusers/models/attention.py:600-623,
 * 772-775; animatediff/models/attention.py:270-304; motion_module.py:191,199), InflatedConv3d 3x3
 * (animatediff/models/resnet.py:19-27, 296-342), Downsample3D (:188-196), Upsample3D (:137-170),
 * and the baddbmm/bmm of the materialised attention used in f32 parity mode and the VAE
 * (diffusers/models/attention.py:342-368, 649-678).
 */
/* CONV3X3_UP2: 3x3 conv over the nearest-neighbour upsampling of the input to (Hout, Wout); 2x is the fast path,
 * any Hout >= Hin works (Upsample3D with a forwarded `upsample_size`, reference resnet.py:152-157, unet.py:644-645) */
enum { FYC_GEMM_PLAIN = 0, FYC_GEMM_CONV3X3 = 1, FYC_GEMM_CONV3X3_UP2 = 2 };
enum { FYC_EPI_LINEAR = 0, FYC_EPI_GEGLU = 1, FYC_EPI_HEADS = 2 };
/* pointwise activation of the LINEAR epilogue (conditioning encoders): exact erf GELU (CLIP-ViT-H `gelu`, ip_adapter/resampler.py:17)
 * and x*sigmoid(1.702x) (CLIP-ViT-L text encoder `quick_gelu`) */
enum { FYC_ACT_NONE = 0, FYC_ACT_GELU = 1, FYC_ACT_QUICK_GELU = 2 };

typedef struct {
  const void* a;         /* PLAIN: [batch][M][lda]; CONV: NHWC input [frames][Hin][Win][Cin] */
  const void* a2;        /* PLAIN only, optional: K columns >= k_split come from a2[m][k - k_split] (row pitch lda2): A = [a | a2] */
  const void* w;         /* [batch?][Nw][ldw], K contiguous; conv: K = 9*Cin ordered (slab, ky, kx, c) with 128-byte channel slabs */
  const float* bias;     /* [N] or NULL (GEGLU: packed order) */
  const float* rowbias;  /* [M / rows_per_batch][ldrb] or NULL (ResnetBlock3D time_emb_proj add) */
  const void* residual;  /* [M][ldr] or NULL */
  void* out;             /* LINEAR/GEGLU: [batch][M][ldo] */
  void* seg_out[3];      /* HEADS: per column segment (q,k,v): [b][heads][tok][d] or transposed [b][heads][d][tok] */
  int32_t seg_transposed[3];
  int32_t seg_ld[3];     /* transposed segments: row pitch in elements (>= tokens; 0 = tokens) */
  int32_t M, N, K;
  int32_t lda, ldw, ldo, ldr;
  int32_t ldrb;          /* row pitch of rowbias in elements (0 = N) */
  int32_t k_split, lda2; /* dual-source A (a2 != NULL): k_split must be a multiple of 64 elements */
  int64_t stride_a, stride_w, stride_o; /* batch strides in elements (0 = shared) */
  int32_t batch;
  int32_t mode;          /* FYC_GEMM_* */
  int32_t epilogue;      /* FYC_EPI_* */
  int32_t Hout, Wout, Hin, Win, Cin, conv_stride; /* conv modes */
  int32_t conv_pad;      /* leading (top/left) zero padding: 1 (default conv) or 0 (diffusers Downsample2D padding=0: F.pad (0,1,0,1)); trailing pad is always 1 */
  int32_t rows_per_batch;
  int32_t seg_cols, heads, tokens; /* HEADS: columns per segment (=heads*d), tokens per batch element */
  float out_scale;
  int32_t dtype;
  int32_t tile;          /* 0 = automatic; else tile config id | ring depth << 8 (see fyc_set_tuning) */
  int32_t act;           /* LINEAR epilogue only: FYC_ACT_* applied to (acc + bias + rowbias) before residual / out_scale */
  /* LayerNorm folded into the GEMM that consumes it (all epilogues):  LN(x) W^T = rstd (x (gamma*W)^T - mean * colsum) + beta W^T.
   * `a` is the UN-normalised input, `w` = W scaled by gamma along K, `bias` already contains beta W^T, ln_colsum[n] = sum_k w[n][k],
   * ln_stats[m] = {mean, rstd} of row m (fyc_row_stats).  acc := rstd[m] * (acc - mean[m] * ln_colsum[n]) before bias.  NULL = off. */
  const float* ln_stats;
  const float* ln_colsum;
  /* ln_nparts > 0: ln_stats is not {mean, rstd} but the `row_parts` buffer of the GEMM that produced `a`: [M][ln_nparts][2] partial
   * {sum, sum of squares} per row; mean / rstd are derived in the epilogue over the K columns with ln_eps.  0 = {mean, rstd}. */
  int32_t ln_nparts; float ln_eps;
  /* Statistics of the OUTPUT, accumulated by the epilogue that writes it (LINEAR epilogue, batch <= 1), so that the GroupNorm /
   * LayerNorm consuming `out` needs no read pass of its own (reference resnet.py:299,322, attention.py:269,383,412,418,
   * motion_module.py:181,261,267).  The sums are of the values AS STORED (rounded to `dtype`); plain stores, nothing to zero.
   *  chan_parts [row tiles][slots][N][2] floats: per (row tile, sample slot, channel) {sum, sum of squares}; a GroupNorm sample
   *    (cs_rows rows: one frame, H*W) that a row tile touches gets slot = sample - first sample of the tile.  Row tile height and
   *    slot count come from fyc_gemm_stat_layout(); fyc_chan_stats_reduce folds the tiles into per-(sample, channel) sums.
   *    cs_rows must be a multiple of 16 and a row tile may touch at most 4 samples.
   *  row_parts [M][row_nparts][2] floats: per row and column tile {sum, sum of squares}; row_nparts must equal
   *    fyc_gemm_row_parts(args).  Consumed through ln_stats / ln_nparts of the next GEMM. */
  float* chan_parts; int32_t cs_rows;
  float* row_parts; int32_t row_nparts;
  /* Optional scratch for split-K (small M, long K: the 8x8-latent level).  fyc_gemm_workspace_bytes(args) says how much the
   * problem would like (0: no split); with less, or NULL, the GEMM runs unsplit.  16-byte aligned device memory, contents
   * undefined before and after the call. */
  void* workspace; int64_t workspace_bytes;
} fyc_gemm_args;
int fyc_gemm(const fyc_gemm_args* a, void* stream);
/* scratch bytes fyc_gemm can use for these arguments (split-K partial sums); 0 = none needed */
int64_t fyc_gemm_workspace_bytes(const fyc_gemm_args* a);
/* number of column tiles fyc_gemm will use for these arguments (M, N, K, mode, dtype, batch, tile are read) = row_nparts */
int fyc_gemm_row_parts(const fyc_gemm_args* a);
/* layout of chan_parts for these arguments (also cs_rows is read): returns the number of row tiles, *tile_rows = rows per tile,
 * *slots = sample slots per tile; chan_parts holds row_tiles * slots * N * 2 floats */
int fyc_gemm_stat_layout(const fyc_gemm_args* a, int32_t* tile_rows, int32_t* slots);

/* cs[o][n] = {sum, sum of squares} over the rows of output sample o (out_rows rows each, a multiple of cs_rows; 0 = cs_rows) of
 * channel n, in f64, from the row-tile partials a fyc_gemm epilogue wrote with cs_rows rows per statistics sample (chan_parts;
 * tile_rows / slots from fyc_gemm_stat_layout).  out_rows = H*W for a per-frame GroupNorm, F*H*W for a cross-frame one.
 * cs: [rows / out_rows][N][2] doubles. */
typedef struct { const float* parts; double* cs; int32_t rows, N, cs_rows, tile_rows, slots, out_rows; } fyc_chan_stats_reduce_args;
int fyc_chan_stats_reduce(const fyc_chan_stats_reduce_args* a, void* stream);

/* ---- fused flash attention (bf16 MFMA, online softmax) ----------------------------------
 * o[b][tok][h*d + i] = softmax_k( q[b,h,tok,:] . k[b,h,key,:] * scale ) @ v
 * Replaces CrossAttention._attention / xformers.memory_efficient_attention for spatial attn1 and
 * attn2 (diffusers/models/attention.py:649-678, 723-730) and IPCrossAttention's two cores
 * (animatediff/models/attention.py:88-120): with o_accumulate=1 the result is added to `o`
 * scaled by o_scale (out = attn_text + scale * attn_ip).
 * q,k: [BH][N][d]; vt: [BH][d][ldvt] (transposed V written by the HEADS epilogue). kv_batch_div: K/V batch
 * index = (b / kv_batch_div) * heads + h  (text K/V are shared by the F frames of a clip).
 */
typedef struct {
  const void* q; const void* k; const void* vt; void* o;
  int32_t batch, heads, n_q, n_k, d;
  int32_t ldo;           /* row stride of o in elements (token-major [batch*n_q][ldo]) */
  int32_t ldvt;          /* row pitch of vt in elements: multiple of 8, >= n_k; pad keys must be finite (zero) */
  int32_t kv_batch_div;
  int32_t o_accumulate;
  float scale, o_scale;
  int32_t dtype;         /* FYC_BF16 only */
} fyc_attn_args;
int fyc_attention(const fyc_attn_args* a, void* stream);

/* ---- temporal self-attention core of the motion module ------------------------------------
 * For every (clip b, pixel p, head h): softmax over the F frames (motion_module.py:371-464 with
 * mm_attn_cross.py:148-177).  qkv is the token-major output of the fused to_q|to_k|to_v GEMM,
 * [(b f p)][3C]; o is token-major [(b f p)][C].  The '(b f) d c -> (b d) f c' transposes of the
 * reference are index arithmetic here.
 */
typedef struct {
  const void* qkv; void* o;
  int32_t clips, frames, pixels, heads, d;
  float scale;
  int32_t dtype;
} fyc_tattn_args;
int fyc_temporal_attention(const fyc_tattn_args* a, void* stream);

/* ---- normalisation -----------------------------------------------------------------------
 * GroupNorm statistics over `rows_per_sample` rows x (C/groups) channels:
 *   cross-frame (ResnetBlock3D.norm1/norm2, conv_norm_out: nn.GroupNorm on the 5-D tensor,
 *   resnet.py:299,322; unet.py:665): rows_per_sample = F*H*W;
 *   per-frame (Transformer3DModel.norm, TemporalTransformer3DModel.norm, VAE): rows_per_sample = H*W.
 * stats: [samples][groups][2] doubles (sum, sum of squares), written by the call.
 * The sums are bitwise repeatable (fixed reduction order, no atomics).  Large samples are cut into row chunks whose partial
 * sums pass through `workspace` (float, fyc_gn_stats_workspace() bytes; 0 = one block per sample, no workspace needed).
 */
typedef struct {
  const void* x; double* stats;
  int32_t rows, C, groups, rows_per_sample;
  int32_t dtype;
  int32_t pad_;
  float* workspace; int64_t workspace_bytes;
} fyc_gn_stats_args;
int fyc_gn_stats(const fyc_gn_stats_args* a, void* stream);
int64_t fyc_gn_stats_workspace(const fyc_gn_stats_args* a);   /* bytes of `workspace` this problem needs (x / stats may be NULL) */

/* y = (x - mean) * rstd * gamma[c] + beta[c], optional SiLU */
typedef struct {
  const void* x; const double* stats; const float* gamma; const float* beta; void* y;
  int32_t rows, C, groups, rows_per_sample;
  float eps; int32_t silu;
  int32_t dtype;
} fyc_gn_apply_args;
int fyc_gn_apply(const fyc_gn_apply_args* a, void* stream);

/* GroupNorm apply (+SiLU) from per-(sample, channel) sums (fyc_gemm chan_parts -> fyc_chan_stats_reduce), with the
 * channel concat of the up blocks folded in (torch.cat([hidden, skip], dim=1) -> norm1, reference unet_blocks.py:763,885;
 * resnet.py:299-302): y[r][0:C1] from x1, y[r][C1:C1+C2] from x2 (x2 = NULL, C2 = 0: one source).  Groups are taken over the
 * concatenated channel index and may straddle the two sources.  cs1 / cs2: [rows / cs_rows][C][2] doubles. */
typedef struct {
  const void* x1; const double* cs1; const void* x2; const double* cs2;
  const float* gamma; const float* beta; void* y;
  int32_t C1, C2, rows, groups, rows_per_sample;
  float eps; int32_t silu;
  int32_t dtype;
  int32_t cs_rows;       /* rows per statistics sample of cs1 / cs2 (0 = rows_per_sample): the producers accumulate per frame (H*W
                          * rows) so that their atomics spread over F times more addresses; a cross-frame norm (rows_per_sample =
                          * F*H*W) adds up the F frame sums here */
} fyc_gn_apply_cs_args;
int fyc_gn_apply_cs(const fyc_gn_apply_cs_args* a, void* stream);

/* LayerNorm over C (eps 1e-5, affine) + optional additive table pe[(row / pe_div) % pe_rows][c]
 * (motion module: PositionalEncoding added to the normalised tokens, motion_module.py:272-278,377) */
typedef struct {
  const void* x; const float* gamma; const float* beta; const float* pe; void* y;
  int32_t rows, C; float eps;
  int32_t pe_div, pe_rows;
  int32_t dtype;
} fyc_layernorm_args;
int fyc_layernorm(const fyc_layernorm_args* a, void* stream);

/* per-row LayerNorm statistics for the folded form above: stats[r] = {mean, 1/sqrt(var + eps)} over the C channels of row r
 * (nn.LayerNorm's biased variance; reference attention.py:383,412,418, motion_module.py:261,267) */
typedef struct { const void* x; float* stats; int32_t rows, C; float eps; int32_t dtype; } fyc_row_stats_args;
int fyc_row_stats(const fyc_row_stats_args* a, void* stream);

/* row softmax (in place, f32 math): x [rows][ld], first `cols` columns (materialised attention).
 * causal_rows = n > 0: row r belongs to query (r % n) and only columns <= r % n take part, the rest become 0
 * (CLIP text encoder's causal mask, transformers CLIPTextTransformer) */
typedef struct { void* x; int64_t rows; int32_t cols, ld; int32_t dtype; int32_t causal_rows; } fyc_softmax_args;
int fyc_softmax_rows(const fyc_softmax_args* a, void* stream);

/* ---- elementwise / layout ---------------------------------------------------------------- */
/* y[r][0:c1] = a[r][:], y[r][c1:c1+c2] = b[r][:]  (torch.cat(dim=1) of unet_blocks.py:763,885) */
typedef struct { const void* a; const void* b; void* y; int64_t rows; int32_t c1, c2; int32_t dtype; } fyc_concat_args;
int fyc_concat_channels(const fyc_concat_args* a, void* stream);

/* y = silu(x) (f32, small: time embeddings) */
typedef struct { const float* x; float* y; int64_t n; } fyc_silu_args;
int fyc_silu_f32(const fyc_silu_args* a, void* stream);

/* f32 [rows][cols] -> dtype [rows][ld] (cols <= ld, tail zero filled) and back */
typedef struct { const float* x; void* y; int64_t rows; int32_t cols, ld; int32_t dtype; } fyc_cast_args;
int fyc_cast_from_f32(const fyc_cast_args* a, void* stream);
typedef struct { const void* x; float* y; int64_t rows; int32_t cols, ld; int32_t dtype; } fyc_cast_to_args;
int fyc_cast_to_f32(const fyc_cast_to_args* a, void* stream);

/* Build the UNet input: latents (B,4,F,h,w) f32, mask (B,1,F|1,h,w) f32 or NULL, first-frame latents
 * (B,4,h,w) f32 -> channels-last [cfg_dup*B*F][h*w][c_pad] with channels
 * [latent(4) | mask(1) | first_frame_block(4) | 0...]  (pipeline_animation.py:693-711). */
typedef struct {
  const float* latents; const float* mask; const float* first; void* x;
  int32_t B, F, HW, c_latent, c_pad, cfg_dup, mask_frames;
  int32_t dtype;
  /* (version 300) mode 0: [latents | mask | first-frame block at frame 0] (use_first_frame_mask_condition_concat, pipeline_animation.py:
   * 693-704); mode 1: [latents | first-frame latents repeated on EVERY frame] (use_first_frame_condition_concat, unet.py:580-586: the
   * UNet itself concatenates `reference_images_latent`), `mask` unused */
  int32_t mode;
} fyc_unet_input_args;
int fyc_unet_input(const fyc_unet_input_args* a, void* stream);

/* Classifier-free guidance + DDIM update in one pass (pipeline_animation.py:754-767,
 * scheduling_ddim.py:308-349).  pred: channels-last [cfg*B*F][HW][ld] (uncond half first);
 * latents (B,4,F,h,w) f32 updated in place.  coef (device f32[4], host-precomputed from the
 * alphas_cumprod table, so the loop has no GPU->CPU sync) = {sqrt(abar_t), sqrt(1-abar_t),
 * sqrt(abar_prev), sqrt(1-abar_prev)}.  pred_type: 0 epsilon, 1 v_prediction, 2 sample. */
typedef struct {
  const void* pred; float* latents; const float* coef;
  int32_t B, F, HW, c_latent, ld, cfg; float guidance;
  int32_t pred_type, clip_sample;
  int32_t dtype;
  /* optional third prediction (pipeline_animation.py:738-760, `video_scale > 0`): per-frame ("single frame") unconditional
   * prediction [B*F][HW][ld];  v = single + video_scale * (uncond - single) + guidance * (cond - uncond).  NULL = plain CFG. */
  const void* pred_single; float video_scale;
  /* (version 300) stochastic DDIM, eta > 0 (scheduling_ddim.py:336-365): prev = ... + sigma * variance_noise, with
   * sigma = eta * sqrt(variance_t) and coef[3] = sqrt(1 - abar_prev - sigma^2) both computed by the host; variance_noise has the
   * latents' shape and layout (B,4,F,h,w) f32 (the caller draws it: torch.randn(..., generator) as the reference does).  NULL = eta 0.
   * clipped_model_output != 0: `use_clipped_model_output` (:342-344) - the noise direction is re-derived from the (clipped)
   * x_0: eps = (x - sqrt(abar_t) x_0) / sqrt(1 - abar_t). */
  const float* variance_noise; float sigma; int32_t clipped_model_output;
} fyc_cfg_ddim_args;
int fyc_cfg_ddim_step(const fyc_cfg_ddim_args* a, void* stream);

/* VAE ends: z (N,4,h,w) f32 * scale -> channels-last [N][hw][c_pad];
 * image channels-last [N][HW][ld] -> (N,3,H,W) f32 = clamp(x/2+0.5, 0, 1) (pipeline_animation.py:402,410) */
typedef struct { const float* z; void* x; int32_t N, C, HW, c_pad; float scale; int32_t dtype; } fyc_nchw_in_args;
int fyc_nchw_to_nhwc(const fyc_nchw_in_args* a, void* stream);
typedef struct { const void* x; float* y; int32_t N, C, HW, ld; float mul, add, lo, hi; int32_t dtype; } fyc_nhwc_out_args;
int fyc_nhwc_to_nchw(const fyc_nhwc_out_args* a, void* stream);

/* ---- conditioning encoders: CLIP text / vision front-ends ------------------------------------------
 * out[r][:] = table[ids[r]][:] + pos[r % seq][:]   (transformers CLIPTextEmbeddings.forward: token + position embedding;
 * called from the reference at pipeline_animation.py:183-186).  Tables f32 [vocab][C] / [seq][C]; ids int64; out [rows][C]. */
typedef struct { const int64_t* ids; const float* table; const float* pos; void* out; int64_t rows; int32_t seq, C, vocab; int32_t dtype; } fyc_embed_args;
int fyc_embed_tokens(const fyc_embed_args* a, void* stream);

/* Non-overlapping patch unfold (the im2col of CLIPVisionEmbeddings' Conv2d(3, C, kernel=stride=P, bias=False); reference
 * ip_adapter/my_ip_adapter.py:132, 280-283 calls the vision tower): image (B,Cin,H,W) f32 NCHW ->
 * out [B*(H/P)*(W/P)][ld], column c*P*P + py*P + px (= the flattened conv weight's K order), columns >= Cin*P*P zero. */
typedef struct { const float* image; void* out; int32_t B, Cin, H, W, P, ld; int32_t dtype; } fyc_patchify_args;
int fyc_patchify(const fyc_patchify_args* a, void* stream);

/* ---- fused temporal self-attention sub-block (motion_module.py:270-283, 371-464) -------------------------------------------
 * out = x + Attn_F(LayerNorm(x) + pe) Wo^T + bo for token rows x [(clip, frame, pixel)][C], attention over the frame axis at
 * every pixel: replaces fyc_row_stats + fyc_gemm(to_q|k|v, LayerNorm folded) + fyc_temporal_attention + fyc_gemm(to_out, +x)
 * with one read and one write of x.  Weights are per head:
 *   w_qkv [heads][128][C]  rows 0..d-1 = gamma-scaled to_q rows of the head, d..2d-1 to_k, 2d..3d-1 to_v, rest zero
 *   colsum, bias [heads][128] f32 (sum_k gamma_k W[n][k];  beta W^T + b),  pe_bias [frames][heads][128] f32 = pe_f W^T or NULL
 *   w_out [heads][C][48]   Wo[n][head*d + k] for k < d, zero for k >= d;  b_out [C] f32
 * Built for dtype bf16, C = 320, 8 heads of 40, 16 frames, pixels % 8 == 0 (fyc_temporal_block_supported says so without
 * launching); x and out must not alias.
 * `wstream` (optional, 16-byte aligned, fyc_temporal_block_wstream_bytes() bytes): the same weights pre-packed for the
 * register-resident kernel (csrc/temporal_block_rr.hip; w_qkv / colsum / bias / pe_bias / w_out are then not read and may be
 * NULL): 16 stages x 73 pieces x 1 KiB, a piece = one MFMA operand fragment of a 16 x 32 block B in lane order (byte 16 l =
 * B[l & 15][8 (l >> 4) .. +8], bf16).  Head h:
 *   stage 2 h      pieces s * 6 + b (s < 10): rows 16 b .. +16 (b < 3) of the head's gamma-scaled to_q rows zero-padded to 48,
 *                  or rows 16 (b - 3) .. +16 (b >= 3) of its to_k rows, columns 32 s .. +32;
 *                  pieces 60..65 = f32 table [16 frames][96]: (beta W^T + b + pe_f W^T) of the 48 q and the 48 k features
 *   stage 2 h + 1  pieces s * 3 + b: to_v rows 16 b .. +16 (padded to 48), columns 32 s .. +32;
 *                  pieces 30 + 20 t + j (t < 2, j < 20): Wo rows 16 j .. +16 with k-slot 8 g + e of k-step t = head feature
 *                  4 g + e (t = 0, e < 4), 16 + 4 g + e - 4 (t = 0, e >= 4), 32 + 4 g + e (t = 1, e < 4; zero from feature 40), zero (t = 1, e >= 4);
 *                  pieces 70..72 = f32 table [48 v features][16 frames] of the same bias sum.
 * This kernel feeds the projections the normalised tokens (x - mean) rstd rounded to bf16 (engine/weights.py::pack_temporal_block
 * builds the stream). */
typedef struct {
  const void* x; void* out;
  const void* w_qkv; const float* colsum; const float* bias; const float* pe_bias;
  const void* w_out; const float* b_out;
  int32_t clips, frames, pixels, heads, d, C;
  float scale, eps;
  int32_t dtype;
  const void* wstream;
} fyc_temporal_block_args;
int fyc_temporal_block(const fyc_temporal_block_args* a, void* stream);
int64_t fyc_temporal_block_wstream_bytes(void);
int fyc_temporal_block_supported(const fyc_temporal_block_args* a);

/* ---- fused GEGLU feed-forward block (diffusers/models/attention.py:772-775, 819-821; animatediff/models/attention.py:489-564;
 *      motion_module.py:270-283) ------------------------------------------------------------------------------------------
 *   h   = GEGLU( LayerNorm(x) W1^T + b1 )                      [rows][hidden]
 *   out = residual + b_out + [x | h] [Wp | Wp W2]^T            [rows][C]        (FF2 merged with the block's output projection)
 * replaces fyc_row_stats + fyc_gemm(FYC_EPI_GEGLU, LayerNorm folded) + fyc_gemm(a2 = h) - the hidden activation never leaves
 * the CU, x is read once.  `wstream` is the pre-packed weight stream (fyc_ff_block_wstream_bytes() bytes, 16-byte aligned):
 * 92 half-stages x 32 pieces x 1 KiB (the kernel runs them through a 4-deep LDS ring, requesting half h + 2 while half h
 * computes), a piece = the A operand of one v_mfma_f32_32x32x16_bf16, i.e. a 32 x 16 weight block B in lane order (byte 16 l of
 * the piece = B[l % 32][8 (l / 32) .. +8], bf16); unused pieces are zero:
 *   half t < 10        pieces 10 s + j (s < 2, j < 10): Wp rows 32 j .. +32, columns 16 (2 t + s) .. +16  (Wp = merged weight [:, :C])
 *   half 10 + 2 c      pieces 2 s + v (s < 14): the 32 value rows (v = 0) / the 32 gate rows (v = 1) of W1 of hidden chunk c (units
 *                      32 c .. +32 in natural order; LayerNorm weight folded in: W1 * gamma), columns 16 s .. +16; for c >= 1,
 *                      piece 28 = f32 bias (b1 + W1 beta) [32 value | 32 gate] of chunk c - 1
 *   half 11 + 2 c      pieces 2 (s - 14) + v (s = 14 .. 19): the remaining W1 columns of chunk c; for c >= 1, pieces 12 + 10 sg + j
 *                      (sg < 2, j < 10) = rows 32 j .. +32 of W2' = merged weight [:, C:], k-slot 8 kh + e (kh < 2, e < 8) of k-step sg =
 *                      hidden unit 32 (c - 1) + 16 sg + 8 (e / 4) + 4 kh + e % 4
 *   halves 90, 91      piece 28 / pieces 12 + ...: the same for hidden chunk 39.
 * The kernel feeds FF1 the normalised tokens (x - mean) rstd rounded to bf16 (mean / variance over C in f32, two-pass).
 * (engine/weights.py::pack_ff_block builds it.)  chan_parts (optional): [rows / 128][C][2] f32 = per 128-row tile and channel
 * {sum, sum of squares} of the values as stored - fyc_gemm's chan_parts layout with tile_rows = 128 and one slot, for
 * fyc_chan_stats_reduce; needs cs_rows % 128 == 0.  Built for dtype bf16, C = 320, hidden = 1280, rows % 128 == 0
 * (fyc_ff_block_supported says so without launching); out must not alias x. */
typedef struct {
  const void* x; const void* residual; void* out;
  const void* wstream; const float* b_out;
  float* chan_parts; int32_t cs_rows;
  int32_t rows, C, hidden;
  float eps;
  int32_t dtype;
} fyc_ff_block_args;
int fyc_ff_block(const fyc_ff_block_args* a, void* stream);
int fyc_ff_block_supported(const fyc_ff_block_args* a);
int64_t fyc_ff_block_wstream_bytes(void);

/* ---- row-panel linear for the short-K projections (animatediff/models/attention.py:270, diffusers/models/attention.py:600-623;
 *      motion_module.py:191, 270-283) ----------------------------------------------------------------------------------------------
 *   out = [GroupNorm](x) W^T + bias (+ residual)       x [rows][K], out / residual [rows][N] bf16, K and N in {320, 640}, rows % 128 == 0
 * replaces fyc_gemm (and, with gn_cs, the fyc_gn_apply_cs pass in front of a proj_in: GroupNorm is applied to the operand registers,
 * from the same per-(statistics sample, channel) f64 {sum, sum of squares} `gn_cs` [rows / (gn_rows_per_sample / gn_stat_samples)][K][2];
 * a GroupNorm sample = gn_rows_per_sample rows = gn_stat_samples consecutive statistics samples; gn_rows_per_sample % 128 == 0).
 * `wstream` (fyc_panel_linear_wstream_bytes(N, K) bytes): (N / 320) passes x (K / 64) stages x 40 pieces x 1 KiB; piece s * 20 + j of
 * stage t of pass P = MFMA operand fragment (byte 16 l = B[l & 15][8 (l >> 4) .. +8]) of the weight block rows 320 P + 16 j .. +16,
 * columns 32 (2 t + s) .. +32 (engine/weights.py::pack_panel_linear).  residual may alias out; out must not alias x. */
typedef struct {
  const void* x; const void* residual; void* out;
  const void* wstream; const float* bias;
  const double* gn_cs; const float* gn_gamma; const float* gn_beta;
  int32_t gn_rows_per_sample, gn_stat_samples, gn_groups;
  float gn_eps;
  int32_t rows, N, K;
  int32_t dtype;
} fyc_panel_linear_args;
int fyc_panel_linear(const fyc_panel_linear_args* a, void* stream);
int fyc_panel_linear_supported(const fyc_panel_linear_args* a);
int64_t fyc_panel_linear_wstream_bytes(int32_t N, int32_t K);

/* ---- weight layouts fyc_gemm expects (one-time, at load): the state-dict tensors of the reference, f32 on the device --------
 * fyc_pack_conv3x3: Conv2d / InflatedConv3d weight (O, I, 3, 3) (animatediff/models/resnet.py:20-27; diffusers resnet.py Conv2d)
 *   -> [O][slab][ky][kx][c in slab], one slab = 128 bytes of input channels (64 bf16 / 32 f32), I zero-padded to a multiple
 *   of 64: the K order of FYC_GEMM_CONV3X3 (ldw = 9 * pad64(I)).
 * fyc_pack_geglu: GEGLU projection ff.net.0.proj (diffusers/models/attention.py:819-821: rows [0, O/2) value, [O/2, O) gate)
 *   -> blocks of 16 value rows followed by their 16 gate rows (weight, cast to `dtype`) and the same order for the f32 bias:
 *   the column order FYC_EPI_GEGLU consumes.  O % 32 == 0. */
typedef struct { const float* w; void* out; int32_t O, I; int32_t dtype; } fyc_pack_conv3x3_args;
int fyc_pack_conv3x3(const fyc_pack_conv3x3_args* a, void* stream);
typedef struct { const float* w; const float* b; void* w_out; float* b_out; int32_t O, I; int32_t dtype; } fyc_pack_geglu_args;
int fyc_pack_geglu(const fyc_pack_geglu_args* a, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FYC_H */
