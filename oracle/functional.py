"""CPU oracle: a functional torch restatement of the FollowYourClick hot path.

TEST INFRASTRUCTURE - NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this package; the product
(``followyourclick_amd``) never does and has no CPU fallback.

Every function restates, in plain fp32 (or fp64) torch on the CPU, what the
reference computes for the sampling loop ``AnimationPipeline.__call__`` ->
``UNet3DConditionModel.forward`` -> ``DDIMScheduler.step`` -> ``vae.decode``.
Weights arrive as a flat ``dict`` with the *reference's* state-dict key names
(SURVEY.md 8a''), so the same tensors can be loaded into the real reference
model (``oracle/refshim.py``, container only) and into the HIP engine.

Pinning: the reference ships no tests / golden vectors for this path
(SURVEY.md 4), so the oracle is pinned against the reference *executed here*:
``oracle/make_golden.py`` runs the real reference modules on seeded inputs and
stores inputs+outputs under ``tests/golden/``; ``tests/test_oracle_golden.py``
checks this restatement against those vectors (CPU, no reference needed).

Citations are ``file:line`` relative to ``/root/reference``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------
@dataclass
class UNetConfig:
    """Subset of UNet3DConditionModel.__init__ kwargs that changes the math
    (animatediff/models/unet.py:43-104) for the shipped inference YAML
    (configs/inference/inference_img_embed_mask_condition_zero_snr_.yaml:1-17)."""

    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    cross_attention_dim: int = 768
    attention_head_dim: int = 8          # used as the head COUNT (unet_blocks.py:437-440)
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock3D",) * 3 + ("DownBlock3D",)
    up_block_types: Tuple[str, ...] = ("UpBlock3D",) + ("CrossAttnUpBlock3D",) * 3
    use_motion_module: bool = True
    motion_module_resolutions: Tuple[int, ...] = (1, 2, 4, 8)
    motion_module_mid_block: bool = False
    motion_num_attention_heads: int = 8
    motion_num_transformer_block: int = 1
    motion_attention_blocks: int = 2     # ("Temporal_Self", "Temporal_Self")
    temporal_position_encoding_max_len: int = 24
    use_fps_condition: bool = True
    use_camera_motion_condition: bool = False   # unet.py:134-137, 538-544
    use_first_frame_mask_condition_concat: bool = True
    use_first_frame_condition_concat: bool = False
    use_ip_cross_attention: bool = False
    ip_scale: float = 1.0
    ip_num_tokens: int = 4
    # IPCrossAttention.__init__ overwrites CrossAttention.scale (= dim_head**-0.5, the softmax
    # temperature; diffusers/models/attention.py:544) with the IP mixing weight
    # (animatediff/models/attention.py:42).  The reference's non-xformers `_attention` path (what the
    # CPU oracle run executes) therefore uses `scale` as the softmax temperature of attn2, while
    # the deployed xformers path (scripts/inference.py:157-158 asserts it) uses d**-0.5.
    # True = reproduce the CPU path bit-for-bit (used only to pin this restatement);
    # False = the deployed (xformers) semantics, which is what the engine implements.
    ip_reference_cpu_scale_quirk: bool = False
    sample_size: int = 64

    @property
    def conv_in_channels(self) -> int:
        # unet.py:121-126
        if self.use_first_frame_condition_concat:
            return self.in_channels * 2
        if self.use_first_frame_mask_condition_concat:
            return self.in_channels * 2 + 1
        return self.in_channels

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4  # unet.py:108


def tiny_unet_config(**kw) -> UNetConfig:
    """A structurally identical but small UNet used by the parity tests."""
    base = dict(block_out_channels=(64, 128, 256, 256), cross_attention_dim=64, sample_size=8)
    base.update(kw)
    return UNetConfig(**base)


@dataclass
class VAEConfig:
    """AutoencoderKL decoder half (diffusers/models/vae.py:147-206, 545-563)."""
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32


# --------------------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------------------
def _lin(sd: SD, p: str, x: Tensor, bias: bool = True) -> Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias") if bias else None)


def sinusoid(values: Tensor, dim: int) -> Tensor:
    """Timesteps(dim, flip_sin_to_cos=True, freq_shift=0): [cos | sin] halves.
    diffusers/models/embeddings.py:21-64, unet.py:129."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    ang = values[:, None].float() * freqs[None, :]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


def timestep_mlp(sd: SD, p: str, x: Tensor) -> Tensor:
    """TimestepEmbedding: linear_1 -> SiLU -> linear_2 (embeddings.py:67-92)."""
    return _lin(sd, p + ".linear_2", F.silu(_lin(sd, p + ".linear_1", x)))


def unet_time_embedding(sd: SD, cfg: UNetConfig, t: Tensor, fps: Optional[Tensor], flow: Optional[Tensor],
                        dtype=torch.float32, camera: Optional[Tensor] = None) -> Tensor:
    """emb = time_embedding(sin(t)) [+ camera_motion_embedding(sin(camera type))] + fps_embedding(sin(fps)) + motion_embedding(sin(flow))
    (unet.py:526-558)."""
    c0 = cfg.block_out_channels[0]
    emb = timestep_mlp(sd, "time_embedding", sinusoid(t, c0).to(dtype))
    if cfg.use_camera_motion_condition and camera is not None:      # unet.py:538-544
        emb = emb + timestep_mlp(sd, "camera_motion_embedding", sinusoid(camera, c0).to(dtype))
    if cfg.use_fps_condition and fps is not None:
        emb = emb + timestep_mlp(sd, "fps_embedding", sinusoid(fps, c0).to(dtype))
        emb = emb + timestep_mlp(sd, "motion_embedding", sinusoid(flow, c0).to(dtype))
    return emb


def conv_frames(x: Tensor, w: Tensor, b: Optional[Tensor], stride: int = 1, padding: int = 1) -> Tensor:
    """InflatedConv3d: Conv2d applied per frame (resnet.py:19-27). x is (b,c,f,h,w)."""
    B, C, Fr, H, W = x.shape
    y = F.conv2d(x.permute(0, 2, 1, 3, 4).reshape(B * Fr, C, H, W), w, b, stride=stride, padding=padding)
    return y.reshape(B, Fr, y.shape[1], y.shape[2], y.shape[3]).permute(0, 2, 1, 3, 4)


def group_norm_cross_frame(x: Tensor, w: Tensor, b: Tensor, groups: int, eps: float) -> Tensor:
    """nn.GroupNorm on the 5-D tensor: statistics over (C/G, F, H, W) per batch element
    (resnet.py:240,263,299,322; unet.py:345,665; use_inflated_groupnorm=False)."""
    return F.group_norm(x, groups, w, b, eps)


def group_norm_per_frame(x: Tensor, w: Tensor, b: Tensor, groups: int, eps: float) -> Tensor:
    """GroupNorm applied to '(b f) c h w' (attention.py:222,269; motion_module.py:184,188)."""
    B, C, Fr, H, W = x.shape
    y = F.group_norm(x.permute(0, 2, 1, 3, 4).reshape(B * Fr, C, H, W), groups, w, b, eps)
    return y.reshape(B, Fr, C, H, W).permute(0, 2, 1, 3, 4)


def resnet_block3d(sd: SD, p: str, x: Tensor, emb: Tensor, groups: int, eps: float) -> Tensor:
    """ResnetBlock3D.forward (resnet.py:296-342), time_embedding_norm='default',
    output_scale_factor=1."""
    h = F.silu(group_norm_cross_frame(x, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], groups, eps))
    h = conv_frames(h, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"])
    temb = _lin(sd, p + ".time_emb_proj", F.silu(emb))
    if temb.shape[0] == h.shape[0] + 1:
        # use_first_frame_condition (resnet.py:310-317): the extra last row is the embedding of timestep 0 and goes to frame 0,
        # frames 1.. get the row of their own batch element
        h = h + temb[:-1, :, None, None, None]
        h[:, :, 0] = h[:, :, 0] - temb[:-1, :, None, None] + temb[-1:, :, None, None]
    else:
        h = h + temb[:, :, None, None, None]
    h = F.silu(group_norm_cross_frame(h, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], groups, eps))
    h = conv_frames(h, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"])
    if (p + ".conv_shortcut.weight") in sd:
        x = conv_frames(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"], padding=0)
    return x + h


def attention_core(q: Tensor, k: Tensor, v: Tensor, heads: int, scale: Optional[float] = None) -> Tensor:
    """softmax(q k^T / sqrt(d)) v with heads folded as in reshape_heads_to_batch_dim
    (diffusers/models/attention.py:572-584, 649-678): channel = head*d + i."""
    B, N, C = q.shape
    d = C // heads

    def split(t):
        return t.reshape(B, t.shape[1], heads, d).permute(0, 2, 1, 3)

    qh, kh, vh = split(q), split(k), split(v)
    sc = d ** -0.5 if scale is None else scale
    # batch-chunked only to bound the (B, heads, N, Nk) score tensor at large shapes; same math
    # (the reference exposes the same valve as CrossAttention._sliced_attention, attention.py:680-716)
    step = max(1, int(2 ** 29 // max(1, heads * N * k.shape[1])))
    outs = []
    for b0 in range(0, B, step):
        s = torch.matmul(qh[b0:b0 + step], kh[b0:b0 + step].transpose(-1, -2)) * sc
        outs.append(torch.matmul(s.softmax(dim=-1), vh[b0:b0 + step]))
    o = torch.cat(outs) if len(outs) > 1 else outs[0]
    return o.permute(0, 2, 1, 3).reshape(B, N, C)


def cross_attention(sd: SD, p: str, x: Tensor, ctx: Optional[Tensor], heads: int) -> Tensor:
    """CrossAttention.forward (diffusers/models/attention.py:592-647): to_q/k/v bias-free,
    to_out.0 with bias, dropout 0."""
    c = x if ctx is None else ctx
    q = _lin(sd, p + ".to_q", x, bias=False)
    k = _lin(sd, p + ".to_k", c, bias=False)
    v = _lin(sd, p + ".to_v", c, bias=False)
    return _lin(sd, p + ".to_out.0", attention_core(q, k, v, heads))


def ip_cross_attention(sd: SD, p: str, x: Tensor, ctx: Tensor, heads: int, scale: float, num_tokens: int,
                       softmax_scale: Optional[float] = None) -> Tensor:
    """IPCrossAttention.forward (animatediff/models/attention.py:49-127): the last
    ``num_tokens`` context tokens are image tokens with their own K/V projections."""
    end = ctx.shape[1] - num_tokens
    text, ip = ctx[:, :end], ctx[:, end:]
    q = _lin(sd, p + ".to_q", x, bias=False)
    o = attention_core(q, _lin(sd, p + ".to_k", text, bias=False), _lin(sd, p + ".to_v", text, bias=False), heads,
                       softmax_scale)
    o_ip = attention_core(q, _lin(sd, p + ".to_k_ip", ip, bias=False), _lin(sd, p + ".to_v_ip", ip, bias=False), heads,
                          softmax_scale)
    return _lin(sd, p + ".to_out.0", o + scale * o_ip)


def feed_forward(sd: SD, p: str, x: Tensor) -> Tensor:
    """FeedForward with GEGLU (diffusers/models/attention.py:733-775, 800-821):
    h, gate = proj(x).chunk(2); h * gelu_erf(gate); Linear(4C->C)."""
    hg = _lin(sd, p + ".net.0.proj", x)
    h, g = hg.chunk(2, dim=-1)
    return _lin(sd, p + ".net.2", h * F.gelu(g))


def layer_norm(sd: SD, p: str, x: Tensor) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def transformer3d(sd: SD, p: str, x: Tensor, ctx: Tensor, cfg: UNetConfig) -> Tensor:
    """Transformer3DModel.forward + BasicTransformerBlock.forward
    (animatediff/models/attention.py:217-308, 489-564), use_linear_projection=False."""
    B, C, Fr, H, W = x.shape
    heads = cfg.attention_head_dim
    h = group_norm_per_frame(x, sd[p + ".norm.weight"], sd[p + ".norm.bias"], cfg.norm_num_groups, 1e-6)
    h = conv_frames(h, sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"], padding=0)
    tok = h.permute(0, 2, 3, 4, 1).reshape(B * Fr, H * W, C)                       # (b f) (h w) c
    ctx_f = ctx[:, None].expand(-1, Fr, -1, -1).reshape(B * Fr, ctx.shape[1], ctx.shape[2])  # attention.py:264
    t = p + ".transformer_blocks.0"
    tok = tok + cross_attention(sd, t + ".attn1", layer_norm(sd, t + ".norm1", tok), None, heads)
    if cfg.use_ip_cross_attention:
        tok = tok + ip_cross_attention(sd, t + ".attn2", layer_norm(sd, t + ".norm2", tok), ctx_f, heads,
                                       cfg.ip_scale, cfg.ip_num_tokens,
                                       cfg.ip_scale if cfg.ip_reference_cpu_scale_quirk else None)
    else:
        tok = tok + cross_attention(sd, t + ".attn2", layer_norm(sd, t + ".norm2", tok), ctx_f, heads)
    tok = tok + feed_forward(sd, t + ".ff", layer_norm(sd, t + ".norm3", tok))
    h = tok.reshape(B, Fr, H, W, C).permute(0, 4, 1, 2, 3)
    h = conv_frames(h, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"], padding=0)
    return h + x


def positional_encoding(channels: int, length: int) -> Tensor:
    """PositionalEncoding.pe[:, :length] (motion_module.py:286-304)."""
    pos = torch.arange(length, dtype=torch.float32)[:, None]
    div = torch.exp(torch.arange(0, channels, 2, dtype=torch.float32) * (-math.log(10000.0) / channels))
    pe = torch.zeros(length, channels)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def temporal_attention(sd: SD, p: str, xn: Tensor, frames: int, heads: int) -> Tensor:
    """VersatileAttention.forward in 'Temporal' self mode (motion_module.py:371-464):
    '(b f) d c -> (b d) f c', + pe, q/k/v, attention over f, to_out, rearrange back."""
    BF, D, C = xn.shape
    B = BF // frames
    t = xn.reshape(B, frames, D, C).permute(0, 2, 1, 3).reshape(B * D, frames, C)
    t = t + positional_encoding(C, frames).to(t.dtype)[None]
    o = cross_attention(sd, p, t, None, heads)
    return o.reshape(B, D, frames, C).permute(0, 2, 1, 3).reshape(BF, D, C)


def motion_module(sd: SD, p: str, x: Tensor, cfg: UNetConfig) -> Tensor:
    """VanillaTemporalModule -> TemporalTransformer3DModel.forward -> TemporalTransformerBlock
    (motion_module.py:90-95, 157-208, 270-283)."""
    B, C, Fr, H, W = x.shape
    p = p + ".temporal_transformer"
    h = group_norm_per_frame(x, sd[p + ".norm.weight"], sd[p + ".norm.bias"], cfg.norm_num_groups, 1e-6)
    tok = h.permute(0, 2, 3, 4, 1).reshape(B * Fr, H * W, C)
    tok = _lin(sd, p + ".proj_in", tok)
    for blk in range(cfg.motion_num_transformer_block):
        t = f"{p}.transformer_blocks.{blk}"
        for a in range(cfg.motion_attention_blocks):
            n = layer_norm(sd, f"{t}.norms.{a}", tok)
            tok = tok + temporal_attention(sd, f"{t}.attention_blocks.{a}", n, Fr, cfg.motion_num_attention_heads)
        tok = tok + feed_forward(sd, t + ".ff", layer_norm(sd, t + ".ff_norm", tok))
    tok = _lin(sd, p + ".proj_out", tok)
    return tok.reshape(B, Fr, H, W, C).permute(0, 4, 1, 2, 3) + x


def upsample_nearest2x(x: Tensor) -> Tensor:
    """F.interpolate(scale_factor=[1,2,2], mode='nearest') (resnet.py:155)."""
    return x.repeat_interleave(2, dim=3).repeat_interleave(2, dim=4)


# --------------------------------------------------------------------------------------
# UNet3D
# --------------------------------------------------------------------------------------
def _has_motion(cfg: UNetConfig, res: int) -> bool:
    return cfg.use_motion_module and res in cfg.motion_module_resolutions


def unet3d_forward(sd: SD, cfg: UNetConfig, sample: Tensor, timestep: Tensor, ctx: Tensor,
                   fps: Optional[Tensor] = None, flow: Optional[Tensor] = None,
                   ip_tokens: Optional[Tensor] = None, taps: Optional[dict] = None,
                   use_first_frame_condition: bool = False, camera: Optional[Tensor] = None,
                   reference_images_latent: Optional[Tensor] = None) -> Tensor:
    """UNet3DConditionModel.forward (unet.py:422-672).

    sample: (B, conv_in_channels, F, h, w); timestep: scalar/(B,) int; ctx: (B, 77, D);
    fps/flow: (B,) ints; ip_tokens: (B, num_tokens, D) already projected image tokens
    (unet.py:592-594 concatenates them after the text tokens).
    ``taps``: optional dict that receives intermediate activations by name."""
    B = sample.shape[0]
    t = timestep.reshape(-1).expand(B) if timestep.dim() <= 1 else timestep
    if use_first_frame_condition:      # unet.py:523-524: one extra embedding row for timestep 0 (the clean first frame)
        t = torch.cat([t, torch.zeros(1, dtype=t.dtype)])
    emb = unet_time_embedding(sd, cfg, t, fps, flow, sample.dtype, camera=camera)
    if cfg.use_first_frame_condition_concat and reference_images_latent is not None:
        # unet.py:580-586: the clean first-frame latents, repeated over the frames, beside the sample
        sample = torch.cat((sample, reference_images_latent.unsqueeze(2).repeat(1, 1, sample.shape[2], 1, 1)), dim=1)
    if cfg.use_ip_cross_attention:
        ctx = torch.cat([ctx, ip_tokens], dim=1)
    g, eps = cfg.norm_num_groups, cfg.norm_eps

    def tap(name, val):
        if taps is not None:
            taps[name] = val

    x = conv_frames(sample, sd["conv_in.weight"], sd["conv_in.bias"])
    if cfg.use_first_frame_condition_concat:
        x = x / 2  # unet.py:589-590
    tap("conv_in", x)
    skips = [x]
    nb = len(cfg.block_out_channels)
    for i, btype in enumerate(cfg.down_block_types):
        p = f"down_blocks.{i}"
        for j in range(cfg.layers_per_block):
            x = resnet_block3d(sd, f"{p}.resnets.{j}", x, emb, g, eps)
            if btype.startswith("CrossAttn"):
                x = transformer3d(sd, f"{p}.attentions.{j}", x, ctx, cfg)
            if _has_motion(cfg, 2 ** i):
                x = motion_module(sd, f"{p}.motion_modules.{j}", x, cfg)
            skips.append(x)
        if i != nb - 1:
            x = conv_frames(x, sd[f"{p}.downsamplers.0.conv.weight"], sd[f"{p}.downsamplers.0.conv.bias"], stride=2)
            skips.append(x)
        tap(f"down{i}", x)
    # mid (unet_blocks.py:342-360); motion_module_mid_block False in the shipped config
    x = resnet_block3d(sd, "mid_block.resnets.0", x, emb, g, eps)
    x = transformer3d(sd, "mid_block.attentions.0", x, ctx, cfg)
    if cfg.use_motion_module and cfg.motion_module_mid_block:
        x = motion_module(sd, "mid_block.motion_modules.0", x, cfg)
    x = resnet_block3d(sd, "mid_block.resnets.1", x, emb, g, eps)
    tap("mid", x)
    for i, btype in enumerate(cfg.up_block_types):
        p = f"up_blocks.{i}"
        for j in range(cfg.layers_per_block + 1):
            x = torch.cat([x, skips.pop()], dim=1)  # unet_blocks.py:763, 885
            x = resnet_block3d(sd, f"{p}.resnets.{j}", x, emb, g, eps)
            if btype.startswith("CrossAttn"):
                x = transformer3d(sd, f"{p}.attentions.{j}", x, ctx, cfg)
            if _has_motion(cfg, 2 ** (nb - 1 - i)):
                x = motion_module(sd, f"{p}.motion_modules.{j}", x, cfg)
        if i != nb - 1:
            # scale_factor 2, or the next skip's size when a level is not a multiple of 2**3
            # (`upsample_size` forwarding, unet.py:466-474, 644-645; resnet.py:152-157)
            tgt = skips[-1].shape[2:]
            if tuple(tgt[1:]) == (2 * x.shape[3], 2 * x.shape[4]):
                x = upsample_nearest2x(x)
            else:
                x = F.interpolate(x, size=tuple(tgt), mode="nearest")
            x = conv_frames(x, sd[f"{p}.upsamplers.0.conv.weight"], sd[f"{p}.upsamplers.0.conv.bias"])
        tap(f"up{i}", x)
    x = F.silu(group_norm_cross_frame(x, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], g, eps))
    return conv_frames(x, sd["conv_out.weight"], sd["conv_out.bias"])


# --------------------------------------------------------------------------------------
# DDIM scheduler (patched diffusers/schedulers/scheduling_ddim.py)
# --------------------------------------------------------------------------------------
@dataclass
class DDIMConfig:
    num_train_timesteps: int = 1000
    beta_start: float = 0.00085
    beta_end: float = 0.012
    beta_schedule: str = "linear"
    steps_offset: int = 1
    clip_sample: bool = False
    set_alpha_to_one: bool = True
    prediction_type: str = "v_prediction"
    rescale_betas_zero_snr: bool = True


def ddim_alphas_cumprod(c: DDIMConfig) -> Tensor:
    """scheduling_ddim.py:183-203 and rescale_zero_terminal_snr :78-111 (fp32, as the reference)."""
    if c.beta_schedule == "linear":
        betas = torch.linspace(c.beta_start, c.beta_end, c.num_train_timesteps, dtype=torch.float32)
    elif c.beta_schedule == "scaled_linear":
        betas = torch.linspace(c.beta_start ** 0.5, c.beta_end ** 0.5, c.num_train_timesteps, dtype=torch.float32) ** 2
    else:
        raise NotImplementedError(c.beta_schedule)
    if c.rescale_betas_zero_snr:
        abar_sqrt = torch.cumprod(1.0 - betas, dim=0).sqrt()
        a0, aT = abar_sqrt[0].clone(), abar_sqrt[-1].clone()
        abar_sqrt = (abar_sqrt - aT) * (a0 / (a0 - aT))
        abar = abar_sqrt ** 2
        alphas = torch.cat([abar[0:1], abar[1:] / abar[:-1]])
        betas = 1 - alphas
    return torch.cumprod(1.0 - betas, dim=0)


def ddim_timesteps(c: DDIMConfig, n: int) -> Tensor:
    """set_timesteps (scheduling_ddim.py:238-252)."""
    ratio = c.num_train_timesteps // n
    return torch.arange(0, n, dtype=torch.int64).mul(ratio).flip(0) + c.steps_offset


def ddim_step(c: DDIMConfig, abar: Tensor, n: int, model_output: Tensor, t: int, sample: Tensor, eta: float = 0.0,
              variance_noise: Optional[Tensor] = None, use_clipped_model_output: bool = False) -> Tensor:
    """DDIMScheduler.step (scheduling_ddim.py:254-376); eta > 0 needs the caller's `variance_noise` (:346-363)."""
    prev_t = t - c.num_train_timesteps // n
    a_t = abar[t]
    a_prev = abar[prev_t] if prev_t >= 0 else (torch.tensor(1.0) if c.set_alpha_to_one else abar[0])
    b_t = 1 - a_t
    if c.prediction_type == "epsilon":
        x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
        eps = model_output
    elif c.prediction_type == "sample":
        x0, eps = model_output, model_output
    elif c.prediction_type == "v_prediction":
        x0 = (a_t ** 0.5) * sample - (b_t ** 0.5) * model_output
        eps = (a_t ** 0.5) * model_output + (b_t ** 0.5) * sample
    else:
        raise ValueError(c.prediction_type)
    if c.clip_sample:
        x0 = x0.clamp(-1, 1)
    variance = ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)          # _get_variance, :229-236
    std = eta * variance ** 0.5
    if use_clipped_model_output:                                         # :342-344
        eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
    prev = a_prev ** 0.5 * x0 + (1 - a_prev - std ** 2) ** 0.5 * eps
    if eta > 0:
        prev = prev + variance ** 0.5 * eta * variance_noise
    return prev


# --------------------------------------------------------------------------------------
# sampling loop (AnimationPipeline.__call__ :686-773, mask-concat conditioning path)
# --------------------------------------------------------------------------------------
def build_model_input(latents: Tensor, first_image_latents: Optional[Tensor], first_images_mask: Optional[Tensor],
                      partial_mask: Optional[Tensor] = None) -> Tensor:
    """9-channel input = cat(latents, mask, first_frame_block) (pipeline_animation.py:693-704);
    mask = clamp(first_images_mask[:, :, 0:1].repeat(F), 0, 1) (:632-635) or frame-0 indicator."""
    B, C, Fr, H, W = latents.shape
    block = torch.zeros_like(latents)
    block[:, :, 0] = first_image_latents
    if partial_mask is not None:
        block[:, :, 0] = block[:, :, 0] * partial_mask
    if first_images_mask is not None:
        mask = first_images_mask[:, :, 0:1].repeat(1, 1, Fr, 1, 1).clamp(0, 1).to(latents.dtype)
    else:
        mask = torch.zeros_like(latents)[:, :1]
        mask[:, :, 0] = 1
    return torch.cat([latents, mask, block], dim=1)


def denoise(sd: SD, cfg: UNetConfig, sched: DDIMConfig, latents: Tensor, text_embeddings: Tensor, num_steps: int,
            guidance_scale: float, first_image_latents: Optional[Tensor] = None,
            first_images_mask: Optional[Tensor] = None, fps: Optional[Tensor] = None,
            flow: Optional[Tensor] = None, ip_tokens: Optional[Tensor] = None,
            callback=None, video_scale: float = 0.0, use_first_frame_condition: bool = False, eta: float = 0.0,
            variance_noises=None, camera: Optional[Tensor] = None) -> Tensor:
    """The DDIM loop of AnimationPipeline.__call__ with use_first_frame_mask_condition_concat
    and classifier-free guidance: text_embeddings is cat[uncond, cond] (2B,77,D)
    (pipeline_animation.py:397, 690-773)."""
    abar = ddim_alphas_cumprod(sched)
    cfg_on = guidance_scale > 1.0
    for i, t in enumerate(ddim_timesteps(sched, num_steps).tolist()):
        if use_first_frame_condition:      # pipeline_animation.py:691-692: frame 0 is pinned to the clean first-frame latents
            latents = latents.clone()
            latents[:, :, 0] = first_image_latents
            x = latents
        elif cfg.use_first_frame_mask_condition_concat:
            x = build_model_input(latents, first_image_latents, first_images_mask)
        else:
            x = latents                  # (use_first_frame_condition_concat: the UNet concatenates reference_images_latent itself)
        if cfg_on:
            x = torch.cat([x] * 2)
        dup = (lambda v: torch.cat([v] * 2) if (cfg_on and v is not None) else v)
        pred = unet3d_forward(sd, cfg, x, torch.tensor(t), text_embeddings, dup(fps), dup(flow), ip_tokens,
                              use_first_frame_condition=use_first_frame_condition, camera=dup(camera),
                              reference_images_latent=dup(first_image_latents) if cfg.use_first_frame_condition_concat else None)
        single = None
        if video_scale > 0:
            # per-frame prediction (pipeline_animation.py:738-752): frames as one-frame clips; the text batch is
            # cat([text_embeddings] * f).chunk(2)[0] exactly as the reference builds it; no fps / flow conditioning
            b2, c_, f_, h_, w_ = x.shape
            xs = x.permute(0, 2, 1, 3, 4).reshape(b2 * f_, c_, h_, w_).unsqueeze(2).chunk(2, dim=0)[0]
            ts_ = torch.cat([text_embeddings] * f_, dim=0).chunk(2, dim=0)[0]
            ps = unet3d_forward(sd, cfg, xs, torch.tensor(t), ts_)
            single = ps.squeeze(2).reshape(b2 // 2, f_, -1, h_, w_).permute(0, 2, 1, 3, 4)
        if cfg_on:
            u, c = pred.chunk(2)
            if single is not None:
                pred = single + video_scale * (u - single) + guidance_scale * (c - u)
            else:
                pred = u + guidance_scale * (c - u)
        latents = ddim_step(sched, abar, num_steps, pred, t, latents, eta=eta,
                            variance_noise=None if variance_noises is None else variance_noises[i])
        if callback is not None:
            callback(i, t, latents)
    return latents


# --------------------------------------------------------------------------------------
# AutoencoderKL decoder (diffusers/models/vae.py:208-224, 575-610)
# --------------------------------------------------------------------------------------
def resnet_block2d(sd: SD, p: str, x: Tensor, groups: int, eps: float = 1e-6) -> Tensor:
    """ResnetBlock2D.forward without temb (diffusers/models/resnet.py:454-493)."""
    h = F.silu(F.group_norm(x, groups, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], eps))
    h = F.conv2d(h, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = F.silu(F.group_norm(h, groups, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], eps))
    h = F.conv2d(h, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if (p + ".conv_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
    return x + h


def vae_attention_block(sd: SD, p: str, x: Tensor, groups: int) -> Tensor:
    """AttentionBlock.forward, single head, softmax in fp32 (diffusers/models/attention.py:328-379)."""
    B, C, H, W = x.shape
    h = F.group_norm(x, groups, sd[p + ".group_norm.weight"], sd[p + ".group_norm.bias"], 1e-6)
    tok = h.reshape(B, C, H * W).transpose(1, 2)
    q, k, v = _lin(sd, p + ".query", tok), _lin(sd, p + ".key", tok), _lin(sd, p + ".value", tok)
    s = torch.matmul(q, k.transpose(-1, -2)) * (1.0 / math.sqrt(C))
    o = torch.matmul(s.float().softmax(dim=-1).to(s.dtype), v)
    o = _lin(sd, p + ".proj_attn", o)
    return o.transpose(1, 2).reshape(B, C, H, W) + x


def vae_decode(sd: SD, cfg: VAEConfig, z: Tensor) -> Tensor:
    """AutoencoderKL.decode(z).sample for z of shape (N, 4, h, w) (already divided by 0.18215)."""
    g = cfg.norm_num_groups
    x = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    x = F.conv2d(x, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    x = resnet_block2d(sd, "decoder.mid_block.resnets.0", x, g)
    x = vae_attention_block(sd, "decoder.mid_block.attentions.0", x, g)
    x = resnet_block2d(sd, "decoder.mid_block.resnets.1", x, g)
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        for j in range(cfg.layers_per_block + 1):
            x = resnet_block2d(sd, f"decoder.up_blocks.{i}.resnets.{j}", x, g)
        if i != nb - 1:
            x = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
            x = F.conv2d(x, sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"],
                         sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    x = F.silu(F.group_norm(x, g, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], 1e-6))
    return F.conv2d(x, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)


def decode_latents(sd: SD, cfg: VAEConfig, latents: Tensor) -> Tensor:
    """AnimationPipeline.decode_latents (pipeline_animation.py:400-413): (b,4,f,h,w) -> (b,3,f,8h,8w) in [0,1]."""
    B, C, Fr, H, W = latents.shape
    z = (latents / 0.18215).permute(0, 2, 1, 3, 4).reshape(B * Fr, C, H, W)
    frames = torch.cat([vae_decode(sd, cfg, z[i:i + 1]) for i in range(z.shape[0])])
    video = frames.reshape(B, Fr, frames.shape[1], frames.shape[2], frames.shape[3]).permute(0, 2, 1, 3, 4)
    return (video / 2 + 0.5).clamp(0, 1).float()


def vae_encode_moments(sd: SD, cfg: VAEConfig, x: Tensor) -> Tensor:
    """AutoencoderKL.encode(x) up to `moments` = quant_conv(encoder(x)) (diffusers/models/vae.py:128-144, 565-573);
    Downsample2D(padding=0) pads (0,1,0,1) then conv stride 2 (diffusers/models/resnet.py:181-190)."""
    g = cfg.norm_num_groups
    h = F.conv2d(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        for j in range(cfg.layers_per_block):
            h = resnet_block2d(sd, f"encoder.down_blocks.{i}.resnets.{j}", h, g)
        if i != nb - 1:
            k = f"encoder.down_blocks.{i}.downsamplers.0.conv"
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[k + ".weight"], sd[k + ".bias"], stride=2)
    h = resnet_block2d(sd, "encoder.mid_block.resnets.0", h, g)
    h = vae_attention_block(sd, "encoder.mid_block.attentions.0", h, g)
    h = resnet_block2d(sd, "encoder.mid_block.resnets.1", h, g)
    h = F.silu(F.group_norm(h, g, sd["encoder.conv_norm_out.weight"], sd["encoder.conv_norm_out.bias"], 1e-6))
    h = F.conv2d(h, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    return F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])
