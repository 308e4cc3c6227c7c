"""Seeded weight factory shared by the oracle, the golden-vector generator and the tests.

TEST INFRASTRUCTURE.  No pretrained checkpoints exist offline, so all
parity work uses weights drawn from a seeded CPU generator.  The schema (names and shapes)
restates what the reference modules register:

* UNet3DConditionModel: animatediff/models/unet.py:105-351, unet_blocks.py (block ctors),
  resnet.py:216-293, attention.py:140-215 & 330-456, motion_module.py:97-155 & 211-268 & 328-365,
  diffusers/models/attention.py:510-590 (CrossAttention), :733-775 (FeedForward/GEGLU),
  diffusers/models/embeddings.py:67-92.
* AutoencoderKL decoder half: diffusers/models/vae.py:147-206, 545-563,
  diffusers/models/unet_2d_blocks.py:320-396 & 1646-1697, diffusers/models/attention.py:265-288.

``tests/golden/schema_*.json`` (written by ``make_golden.py`` from the real reference's
``state_dict()``) pins these tables.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Tuple

import torch

from .functional import UNetConfig, VAEConfig, positional_encoding

Shapes = "OrderedDict[str, Tuple[int, ...]]"


def _conv(s, p, o, i, k):
    s[p + ".weight"] = (o, i, k, k)
    s[p + ".bias"] = (o,)


def _lin(s, p, o, i, bias=True):
    s[p + ".weight"] = (o, i)
    if bias:
        s[p + ".bias"] = (o,)


def _norm(s, p, c):
    s[p + ".weight"] = (c,)
    s[p + ".bias"] = (c,)


def _resnet(s, p, cin, cout, temb):
    _norm(s, p + ".norm1", cin)
    _conv(s, p + ".conv1", cout, cin, 3)
    if temb:
        _lin(s, p + ".time_emb_proj", cout, temb)
    _norm(s, p + ".norm2", cout)
    _conv(s, p + ".conv2", cout, cout, 3)
    if cin != cout:
        _conv(s, p + ".conv_shortcut", cout, cin, 1)


def _ff(s, p, c):
    _lin(s, p + ".net.0.proj", 8 * c, c)
    _lin(s, p + ".net.2", c, 4 * c)


def _xattn(s, p, c, ctx, ip=False):
    _lin(s, p + ".to_q", c, c, bias=False)
    _lin(s, p + ".to_k", c, ctx, bias=False)
    _lin(s, p + ".to_v", c, ctx, bias=False)
    _lin(s, p + ".to_out.0", c, c)
    if ip:
        _lin(s, p + ".to_k_ip", c, ctx, bias=False)
        _lin(s, p + ".to_v_ip", c, ctx, bias=False)


def _transformer(s, p, c, cfg: UNetConfig):
    _norm(s, p + ".norm", c)
    _conv(s, p + ".proj_in", c, c, 1)
    t = p + ".transformer_blocks.0"
    _xattn(s, t + ".attn1", c, c)
    _norm(s, t + ".norm1", c)
    _xattn(s, t + ".attn2", c, cfg.cross_attention_dim, ip=cfg.use_ip_cross_attention)
    _norm(s, t + ".norm2", c)
    _ff(s, t + ".ff", c)
    _norm(s, t + ".norm3", c)
    _conv(s, p + ".proj_out", c, c, 1)


def _motion(s, p, c, cfg: UNetConfig):
    p = p + ".temporal_transformer"
    _norm(s, p + ".norm", c)
    _lin(s, p + ".proj_in", c, c)
    for b in range(cfg.motion_num_transformer_block):
        t = f"{p}.transformer_blocks.{b}"
        for a in range(cfg.motion_attention_blocks):
            _xattn(s, f"{t}.attention_blocks.{a}", c, c)
            s[f"{t}.attention_blocks.{a}.pos_encoder.pe"] = (1, cfg.temporal_position_encoding_max_len, c)
        for a in range(cfg.motion_attention_blocks):
            _norm(s, f"{t}.norms.{a}", c)
        _ff(s, t + ".ff", c)
        _norm(s, t + ".ff_norm", c)
    _lin(s, p + ".proj_out", c, c)


def unet_state_shapes(cfg: UNetConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    boc = cfg.block_out_channels
    temb = cfg.time_embed_dim
    nb = len(boc)
    _conv(s, "conv_in", boc[0], cfg.conv_in_channels, 3)
    for name in (["time_embedding"] + (["camera_motion_embedding"] if cfg.use_camera_motion_condition else [])
                 + (["fps_embedding", "motion_embedding"] if cfg.use_fps_condition else [])):
        _lin(s, name + ".linear_1", temb, boc[0])
        _lin(s, name + ".linear_2", temb, temb)
    out = boc[0]
    for i, bt in enumerate(cfg.down_block_types):
        inp, out = out, boc[i]
        p = f"down_blocks.{i}"
        for j in range(cfg.layers_per_block):
            _resnet(s, f"{p}.resnets.{j}", inp if j == 0 else out, out, temb)
            if bt.startswith("CrossAttn"):
                _transformer(s, f"{p}.attentions.{j}", out, cfg)
            if cfg.use_motion_module and (2 ** i) in cfg.motion_module_resolutions:
                _motion(s, f"{p}.motion_modules.{j}", out, cfg)
        if i != nb - 1:
            _conv(s, f"{p}.downsamplers.0.conv", out, out, 3)
    c = boc[-1]
    _resnet(s, "mid_block.resnets.0", c, c, temb)
    _transformer(s, "mid_block.attentions.0", c, cfg)
    if cfg.use_motion_module and cfg.motion_module_mid_block:
        _motion(s, "mid_block.motion_modules.0", c, cfg)
    _resnet(s, "mid_block.resnets.1", c, c, temb)
    rev = list(reversed(boc))
    out = rev[0]
    for i, bt in enumerate(cfg.up_block_types):
        prev, out = out, rev[i]
        inp = rev[min(i + 1, nb - 1)]
        p = f"up_blocks.{i}"
        nl = cfg.layers_per_block + 1
        for j in range(nl):
            skip = inp if j == nl - 1 else out
            rin = prev if j == 0 else out
            _resnet(s, f"{p}.resnets.{j}", rin + skip, out, temb)
            if bt.startswith("CrossAttn"):
                _transformer(s, f"{p}.attentions.{j}", out, cfg)
            if cfg.use_motion_module and (2 ** (nb - 1 - i)) in cfg.motion_module_resolutions:
                _motion(s, f"{p}.motion_modules.{j}", out, cfg)
        if i != nb - 1:
            _conv(s, f"{p}.upsamplers.0.conv", out, out, 3)
    _norm(s, "conv_norm_out", boc[0])
    _conv(s, "conv_out", cfg.out_channels, boc[0], 3)
    return s


def vae_decoder_state_shapes(cfg: VAEConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    boc = cfg.block_out_channels
    _conv(s, "post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
    _conv(s, "decoder.conv_in", boc[-1], cfg.latent_channels, 3)
    c = boc[-1]
    _resnet(s, "decoder.mid_block.resnets.0", c, c, 0)
    a = "decoder.mid_block.attentions.0"
    _norm(s, a + ".group_norm", c)
    for n in ("query", "key", "value", "proj_attn"):
        _lin(s, f"{a}.{n}", c, c)
    _resnet(s, "decoder.mid_block.resnets.1", c, c, 0)
    rev = list(reversed(boc))
    out = rev[0]
    for i in range(len(boc)):
        prev, out = out, rev[i]
        for j in range(cfg.layers_per_block + 1):
            _resnet(s, f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else out, out, 0)
        if i != len(boc) - 1:
            _conv(s, f"decoder.up_blocks.{i}.upsamplers.0.conv", out, out, 3)
    _norm(s, "decoder.conv_norm_out", boc[0])
    _conv(s, "decoder.conv_out", cfg.out_channels, boc[0], 3)
    return s


def vae_encoder_state_shapes(cfg: VAEConfig, in_channels: int = 3) -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    boc = cfg.block_out_channels
    _conv(s, "encoder.conv_in", boc[0], in_channels, 3)
    out = boc[0]
    for i in range(len(boc)):
        inp, out = out, boc[i]
        for j in range(cfg.layers_per_block):
            _resnet(s, f"encoder.down_blocks.{i}.resnets.{j}", inp if j == 0 else out, out, 0)
        if i != len(boc) - 1:
            _conv(s, f"encoder.down_blocks.{i}.downsamplers.0.conv", out, out, 3)
    c = boc[-1]
    _resnet(s, "encoder.mid_block.resnets.0", c, c, 0)
    a = "encoder.mid_block.attentions.0"
    _norm(s, a + ".group_norm", c)
    for n in ("query", "key", "value", "proj_attn"):
        _lin(s, f"{a}.{n}", c, c)
    _resnet(s, "encoder.mid_block.resnets.1", c, c, 0)
    _norm(s, "encoder.conv_norm_out", c)
    _conv(s, "encoder.conv_out", 2 * cfg.latent_channels, c, 3)
    _conv(s, "quant_conv", 2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
    return s


def make_weights(shapes: "OrderedDict[str, Tuple[int, ...]]", seed: int, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Deterministic fp32 weights: every tensor drawn in key order from one CPU generator.

    Zero-initialised tensors of the reference (motion proj_out, fps/motion linear_2;
    motion_module.py:87-88, unet.py:141-146) are drawn non-zero too, otherwise the temporal
    path and the fps/flow conditioning would contribute exactly 0 and be untested."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = OrderedDict()
    for name, shape in shapes.items():
        if name.endswith("pos_encoder.pe"):
            sd[name] = positional_encoding(shape[2], shape[1])[None].clone()
            continue
        r = torch.randn(shape, generator=g, dtype=torch.float32)
        if name.endswith(".weight") and len(shape) == 1:
            sd[name] = 1.0 + 0.1 * r          # norm gains
        elif name.endswith(".bias"):
            sd[name] = 0.05 * r
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            sd[name] = r * (gain / math.sqrt(fan_in))
    return sd


def seeded_inputs(cfg: UNetConfig, batch: int, frames: int, h: int, w: int, seed: int, ctx_len: int = 77):
    """Seeded synthetic inputs in the shapes SURVEY.md 8d lists (CPU generator)."""
    g = torch.Generator().manual_seed(seed)
    latents = torch.randn(batch, cfg.in_channels, frames, h, w, generator=g)
    first = 0.18215 * torch.randn(batch, cfg.in_channels, h, w, generator=g) * 5.0
    mask = (torch.rand(batch, 1, 1, h, w, generator=g) > 0.5).float()
    text = torch.randn(2 * batch, ctx_len, cfg.cross_attention_dim, generator=g)
    ip = torch.randn(2 * batch, cfg.ip_num_tokens, cfg.cross_attention_dim, generator=g)
    return dict(latents=latents, first_image_latents=first, first_images_mask=mask, text=text, ip_tokens=ip)
