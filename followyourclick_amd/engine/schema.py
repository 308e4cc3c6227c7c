"""State-dict schema of the models the engine replaces, with the reference's exact key names.

The checkpoint format of the reference *is* its state-dict key set (SURVEY.md 8a''): 1 254 entries for
the shipped UNet3D config (1 214 parameters + 40 `pos_encoder.pe` buffers), plus `to_k_ip/to_v_ip` when
built with use_ip_cross_attention.  `unet_schema` / `vae_decoder_schema` enumerate name -> shape in
module registration order; tests/test_schema.py pins them against tests/golden/schema_*.json (dumped
from the reference modules).  Source of the structure: reference animatediff/models/unet.py:105-351,
unet_blocks.py, resnet.py:216-293, attention.py:140-215/330-456, motion_module.py:97-155/211-268/328-365,
diffusers/models/attention.py:510-590/733-775, embeddings.py:67-92, vae.py:147-206/545-563.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Tuple

import torch

from .config import UNet3DConfig, VAEDecoderConfig
from .weights import sinusoidal_pe

Schema = "OrderedDict[str, Tuple[int, ...]]"


class _S(OrderedDict):
    def conv(self, p, o, i, k):
        self[p + ".weight"] = (o, i, k, k)
        self[p + ".bias"] = (o,)

    def lin(self, p, o, i, bias=True):
        self[p + ".weight"] = (o, i)
        if bias:
            self[p + ".bias"] = (o,)

    def norm(self, p, c):
        self[p + ".weight"] = (c,)
        self[p + ".bias"] = (c,)

    def resnet(self, p, cin, cout, temb):
        self.norm(p + ".norm1", cin)
        self.conv(p + ".conv1", cout, cin, 3)
        if temb:
            self.lin(p + ".time_emb_proj", cout, temb)
        self.norm(p + ".norm2", cout)
        self.conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            self.conv(p + ".conv_shortcut", cout, cin, 1)

    def attn(self, p, c, ctx, ip=False):
        self.lin(p + ".to_q", c, c, False)
        self.lin(p + ".to_k", c, ctx, False)
        self.lin(p + ".to_v", c, ctx, False)
        self.lin(p + ".to_out.0", c, c)
        if ip:
            self.lin(p + ".to_k_ip", c, ctx, False)
            self.lin(p + ".to_v_ip", c, ctx, False)

    def ff(self, p, c):
        self.lin(p + ".net.0.proj", 8 * c, c)
        self.lin(p + ".net.2", c, 4 * c)

    def transformer(self, p, c, cfg: UNet3DConfig):
        self.norm(p + ".norm", c)
        self.conv(p + ".proj_in", c, c, 1)
        t = p + ".transformer_blocks.0"
        self.attn(t + ".attn1", c, c)
        self.norm(t + ".norm1", c)
        self.attn(t + ".attn2", c, cfg.cross_attention_dim, cfg.use_ip_cross_attention)
        self.norm(t + ".norm2", c)
        self.ff(t + ".ff", c)
        self.norm(t + ".norm3", c)
        self.conv(p + ".proj_out", c, c, 1)

    def motion(self, p, c, cfg: UNet3DConfig):
        p += ".temporal_transformer"
        self.norm(p + ".norm", c)
        self.lin(p + ".proj_in", c, c)
        for b in range(cfg.motion_num_transformer_block):
            t = f"{p}.transformer_blocks.{b}"
            for a in range(cfg.motion_attention_blocks):
                self.attn(f"{t}.attention_blocks.{a}", c, c)
                if cfg.temporal_position_encoding:
                    self[f"{t}.attention_blocks.{a}.pos_encoder.pe"] = (1, cfg.temporal_position_encoding_max_len, c)
            for a in range(cfg.motion_attention_blocks):
                self.norm(f"{t}.norms.{a}", c)
            self.ff(t + ".ff", c)
            self.norm(t + ".ff_norm", c)
        self.lin(p + ".proj_out", c, c)


def unet_schema(cfg: UNet3DConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    s = _S()
    boc, temb, nb = cfg.block_out_channels, cfg.time_embed_dim, len(cfg.block_out_channels)
    s.conv("conv_in", boc[0], cfg.conv_in_channels, 3)
    for name in (["time_embedding"] + (["camera_motion_embedding"] if cfg.use_camera_motion_condition else [])
                 + (["fps_embedding", "motion_embedding"] if cfg.use_fps_condition else [])):
        s.lin(name + ".linear_1", temb, boc[0])
        s.lin(name + ".linear_2", temb, temb)
    out = boc[0]
    for i, bt in enumerate(cfg.down_block_types):
        inp, out = out, boc[i]
        for j in range(cfg.layers_per_block):
            s.resnet(f"down_blocks.{i}.resnets.{j}", inp if j == 0 else out, out, temb)
            if bt.startswith("CrossAttn"):
                s.transformer(f"down_blocks.{i}.attentions.{j}", out, cfg)
            if cfg.use_motion_module and (2 ** i) in cfg.motion_module_resolutions:
                s.motion(f"down_blocks.{i}.motion_modules.{j}", out, cfg)
        if i != nb - 1:
            s.conv(f"down_blocks.{i}.downsamplers.0.conv", out, out, 3)
    c = boc[-1]
    s.resnet("mid_block.resnets.0", c, c, temb)
    s.transformer("mid_block.attentions.0", c, cfg)
    if cfg.use_motion_module and cfg.motion_module_mid_block:
        s.motion("mid_block.motion_modules.0", c, cfg)
    s.resnet("mid_block.resnets.1", c, c, temb)
    rev = list(reversed(boc))
    out = rev[0]
    for i, bt in enumerate(cfg.up_block_types):
        prev, out = out, rev[i]
        inp = rev[min(i + 1, nb - 1)]
        nl = cfg.layers_per_block + 1
        for j in range(nl):
            s.resnet(f"up_blocks.{i}.resnets.{j}", (prev if j == 0 else out) + (inp if j == nl - 1 else out), out, temb)
            if bt.startswith("CrossAttn"):
                s.transformer(f"up_blocks.{i}.attentions.{j}", out, cfg)
            if cfg.use_motion_module and (2 ** (nb - 1 - i)) in cfg.motion_module_resolutions:
                s.motion(f"up_blocks.{i}.motion_modules.{j}", out, cfg)
        if i != nb - 1:
            s.conv(f"up_blocks.{i}.upsamplers.0.conv", out, out, 3)
    s.norm("conv_norm_out", boc[0])
    s.conv("conv_out", cfg.out_channels, boc[0], 3)
    return s


def vae_decoder_schema(cfg: VAEDecoderConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    s = _S()
    boc = cfg.block_out_channels
    s.conv("post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
    s.conv("decoder.conv_in", boc[-1], cfg.latent_channels, 3)
    c = boc[-1]
    s.resnet("decoder.mid_block.resnets.0", c, c, 0)
    a = "decoder.mid_block.attentions.0"
    s.norm(a + ".group_norm", c)
    for n in ("query", "key", "value", "proj_attn"):
        s.lin(f"{a}.{n}", c, c)
    s.resnet("decoder.mid_block.resnets.1", c, c, 0)
    rev = list(reversed(boc))
    out = rev[0]
    for i in range(len(boc)):
        prev, out = out, rev[i]
        for j in range(cfg.layers_per_block + 1):
            s.resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else out, out, 0)
        if i != len(boc) - 1:
            s.conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", out, out, 3)
    s.norm("decoder.conv_norm_out", boc[0])
    s.conv("decoder.conv_out", cfg.out_channels, boc[0], 3)
    return s


def vae_encoder_schema(cfg: VAEDecoderConfig, in_channels: int = 3) -> "OrderedDict[str, Tuple[int, ...]]":
    """Encoder half + quant_conv (reference diffusers/models/vae.py:67-144, 560)."""
    s = _S()
    boc = cfg.block_out_channels
    s.conv("encoder.conv_in", boc[0], in_channels, 3)
    out = boc[0]
    for i in range(len(boc)):
        inp, out = out, boc[i]
        for j in range(cfg.layers_per_block):
            s.resnet(f"encoder.down_blocks.{i}.resnets.{j}", inp if j == 0 else out, out, 0)
        if i != len(boc) - 1:
            s.conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", out, out, 3)
    c = boc[-1]
    s.resnet("encoder.mid_block.resnets.0", c, c, 0)
    a = "encoder.mid_block.attentions.0"
    s.norm(a + ".group_norm", c)
    for n in ("query", "key", "value", "proj_attn"):
        s.lin(f"{a}.{n}", c, c)
    s.resnet("encoder.mid_block.resnets.1", c, c, 0)
    s.norm("encoder.conv_norm_out", c)
    s.conv("encoder.conv_out", 2 * cfg.latent_channels, c, 3)
    s.conv("quant_conv", 2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
    return s


def random_state_dict(schema: "OrderedDict[str, Tuple[int, ...]]", seed: int, materialize: bool = True) -> Dict[str, torch.Tensor]:
    """Random-initialised weights of the architecture (no checkpoints exist offline): N(0, 1/fan_in)
    kernels, norm gains 1 + 0.1 N, small biases; `pos_encoder.pe` buffers analytic.  With
    materialize=False tensors are left uninitialised (ranks that receive the weights by broadcast)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = OrderedDict()
    for name, shape in schema.items():
        if name.endswith("pos_encoder.pe"):
            sd[name] = sinusoidal_pe(shape[2], shape[1])[None].clone()
        elif not materialize:
            sd[name] = torch.empty(shape, dtype=torch.float32)
        else:
            r = torch.randn(shape, generator=g, dtype=torch.float32)
            if name.endswith(".weight") and len(shape) == 1:
                sd[name] = 1.0 + 0.1 * r
            elif name.endswith(".bias"):
                sd[name] = 0.05 * r
            else:
                sd[name] = r / math.sqrt(max(1, math.prod(shape[1:])))
    return sd
