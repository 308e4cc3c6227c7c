"""CPU restatement (TEST INFRASTRUCTURE, never imported by the product path) of the conditioning encoders that sit in
front of the sampling loop:

  * CLIP text encoder and CLIP vision tower - third-party code, NOT under /root/reference: the reference calls
    `transformers.CLIPTextModel` (animatediff/pipelines/pipeline_animation.py:183-186, 229-232) and
    `transformers.CLIPVisionModelWithProjection` (ip_adapter/my_ip_adapter.py:58, 132, 280-283) with no version pin.
    Restated from the published CLIP architecture (pre-LN transformer; causal mask on the text side; class token +
    learned positions + pre_layrnorm on the vision side) and pinned against transformers 5.15.0 of this image through
    tests/golden/enc_*.npz (oracle/make_golden_encoders.py).
  * ImageProjModel (ip_adapter/my_ip_adapter.py:28-45) and Resampler / PerceiverAttention / FeedForward
    (ip_adapter/resampler.py:13-20, 36-78, 81-147) - reference code, pinned against the real classes.

State-dict names are transformers' / the reference's.  A leading `text_model.` on the text keys (transformers 4
checkpoints, SD-1.5 `text_encoder/`) is accepted.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


@dataclass
class ClipTextConfig:
    vocab_size: int = 49408
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    max_position_embeddings: int = 77
    hidden_act: str = "quick_gelu"
    layer_norm_eps: float = 1e-5


@dataclass
class ClipVisionConfig:
    hidden_size: int = 1280
    intermediate_size: int = 5120
    num_hidden_layers: int = 32
    num_attention_heads: int = 16
    image_size: int = 224
    patch_size: int = 14
    projection_dim: int = 1024
    hidden_act: str = "gelu"
    layer_norm_eps: float = 1e-5


@dataclass
class ResamplerConfig:
    dim: int = 768
    depth: int = 4
    dim_head: int = 64
    heads: int = 12
    num_queries: int = 16
    embedding_dim: int = 1280
    output_dim: int = 768
    ff_mult: int = 4


def _act(x: Tensor, name: str) -> Tensor:
    if name == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)
    if name == "gelu":
        return F.gelu(x)
    raise NotImplementedError(name)


def _ln(sd: SD, p: str, x: Tensor, eps: float = 1e-5) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _lin(sd: SD, p: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def clip_encoder_layer(sd: SD, p: str, x: Tensor, heads: int, act: str, eps: float, causal: bool) -> Tensor:
    """CLIPEncoderLayer: x += attn(LN1 x); x += fc2(act(fc1(LN2 x))); attention scale d^-0.5, optional causal mask"""
    B, N, C = x.shape
    d = C // heads
    h = _ln(sd, p + ".layer_norm1", x, eps)
    q, k, v = (_lin(sd, f"{p}.self_attn.{n}_proj", h).view(B, N, heads, d).transpose(1, 2) for n in ("q", "k", "v"))
    s = (q @ k.transpose(-1, -2)) * d ** -0.5
    if causal:
        s = s.masked_fill(torch.ones(N, N, dtype=torch.bool).triu(1), float("-inf"))
    o = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, N, C)
    x = x + _lin(sd, p + ".self_attn.out_proj", o)
    h = _ln(sd, p + ".layer_norm2", x, eps)
    return x + _lin(sd, p + ".mlp.fc2", _act(_lin(sd, p + ".mlp.fc1", h), act))


def _strip(sd: SD, prefix: str) -> SD:
    return {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in sd.items()}


def clip_text_forward(sd: SD, cfg: ClipTextConfig, input_ids: Tensor) -> Tensor:
    """CLIPTextModel(input_ids)[0]: last_hidden_state after final_layer_norm, (B, 77, hidden)"""
    sd = _strip(sd, "text_model.")
    B, N = input_ids.shape
    x = sd["embeddings.token_embedding.weight"][input_ids] + sd["embeddings.position_embedding.weight"][:N][None]
    for i in range(cfg.num_hidden_layers):
        x = clip_encoder_layer(sd, f"encoder.layers.{i}", x, cfg.num_attention_heads, cfg.hidden_act, cfg.layer_norm_eps, causal=True)
    return _ln(sd, "final_layer_norm", x, cfg.layer_norm_eps)


def clip_vision_forward(sd: SD, cfg: ClipVisionConfig, pixel_values: Tensor) -> Tuple[List[Tensor], Tensor]:
    """CLIPVisionModelWithProjection(pixel_values, output_hidden_states=True) -> (hidden_states, image_embeds):
    hidden_states[0] = pre_layrnorm(embeddings), hidden_states[i] = output of layer i (no post LN);
    image_embeds = visual_projection(post_layernorm(last[:, 0]))"""
    p = "vision_model."
    B = pixel_values.shape[0]
    patches = F.conv2d(pixel_values, sd[p + "embeddings.patch_embedding.weight"], stride=cfg.patch_size).flatten(2).transpose(1, 2)
    cls = sd[p + "embeddings.class_embedding"].expand(B, 1, -1)
    x = torch.cat([cls, patches], dim=1) + sd[p + "embeddings.position_embedding.weight"][None]
    x = _ln(sd, p + "pre_layrnorm", x, cfg.layer_norm_eps)
    hs = [x]
    for i in range(cfg.num_hidden_layers):
        x = clip_encoder_layer(sd, f"{p}encoder.layers.{i}", x, cfg.num_attention_heads, cfg.hidden_act, cfg.layer_norm_eps, causal=False)
        hs.append(x)
    pooled = _ln(sd, p + "post_layernorm", x[:, 0], cfg.layer_norm_eps)
    return hs, F.linear(pooled, sd["visual_projection.weight"])


def image_proj_forward(sd: SD, image_embeds: Tensor, num_tokens: int, cross_attention_dim: int) -> Tensor:
    """ImageProjModel.forward (ip_adapter/my_ip_adapter.py:39-45): Linear -> (B, tokens, D) -> LayerNorm"""
    t = _lin(sd, "proj", image_embeds).reshape(-1, num_tokens, cross_attention_dim)
    return _ln(sd, "norm", t)


def resampler_forward(sd: SD, cfg: ResamplerConfig, x: Tensor) -> Tensor:
    """Resampler.forward (ip_adapter/resampler.py:125-147) with apply_pos_emb=False, num_latents_mean_pooled=0 (the
    MyIPAdapterPlus construction, my_ip_adapter.py:240-250).  PerceiverAttention (:55-78): q from LN2(latents), k/v from
    cat(LN1(x), LN2(latents)); q,k each scaled by d^-1/4; softmax in f32."""
    B = x.shape[0]
    H, d = cfg.heads, cfg.dim_head
    lat = sd["latents"].repeat(B, 1, 1)
    x = _lin(sd, "proj_in", x)
    for i in range(cfg.depth):
        a, f = f"layers.{i}.0", f"layers.{i}.1"
        xn, ln = _ln(sd, a + ".norm1", x), _ln(sd, a + ".norm2", lat)
        q = _lin(sd, a + ".to_q", ln)
        k, v = _lin(sd, a + ".to_kv", torch.cat([xn, ln], dim=-2)).chunk(2, dim=-1)
        q, k, v = (t.view(B, t.shape[1], H, d).transpose(1, 2) for t in (q, k, v))
        scale = 1 / math.sqrt(math.sqrt(d))
        w = torch.softmax(((q * scale) @ (k * scale).transpose(-2, -1)).float(), dim=-1).type(q.dtype)
        o = (w @ v).permute(0, 2, 1, 3).reshape(B, lat.shape[1], -1)
        lat = _lin(sd, a + ".to_out", o) + lat
        h = _ln(sd, f + ".0", lat)
        lat = _lin(sd, f + ".3", F.gelu(_lin(sd, f + ".1", h))) + lat
    return _ln(sd, "norm_out", _lin(sd, "proj_out", lat))


# ---- parameter name -> shape (for seeded weights; same role as oracle/weights.py) -------------------------------------------
def _layer_shapes(s: "OrderedDict", p: str, C: int, inter: int) -> None:
    for n in ("k", "v", "q", "out"):
        s[f"{p}.self_attn.{n}_proj.weight"], s[f"{p}.self_attn.{n}_proj.bias"] = (C, C), (C,)
    s[p + ".layer_norm1.weight"], s[p + ".layer_norm1.bias"] = (C,), (C,)
    s[p + ".mlp.fc1.weight"], s[p + ".mlp.fc1.bias"] = (inter, C), (inter,)
    s[p + ".mlp.fc2.weight"], s[p + ".mlp.fc2.bias"] = (C, inter), (C,)
    s[p + ".layer_norm2.weight"], s[p + ".layer_norm2.bias"] = (C,), (C,)


def clip_text_shapes(cfg: ClipTextConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["embeddings.token_embedding.weight"] = (cfg.vocab_size, cfg.hidden_size)
    s["embeddings.position_embedding.weight"] = (cfg.max_position_embeddings, cfg.hidden_size)
    for i in range(cfg.num_hidden_layers):
        _layer_shapes(s, f"encoder.layers.{i}", cfg.hidden_size, cfg.intermediate_size)
    s["final_layer_norm.weight"], s["final_layer_norm.bias"] = (cfg.hidden_size,), (cfg.hidden_size,)
    return s


def clip_vision_shapes(cfg: ClipVisionConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    C, p = cfg.hidden_size, "vision_model."
    s[p + "embeddings.class_embedding"] = (C,)
    s[p + "embeddings.patch_embedding.weight"] = (C, 3, cfg.patch_size, cfg.patch_size)
    s[p + "embeddings.position_embedding.weight"] = ((cfg.image_size // cfg.patch_size) ** 2 + 1, C)
    s[p + "pre_layrnorm.weight"], s[p + "pre_layrnorm.bias"] = (C,), (C,)
    for i in range(cfg.num_hidden_layers):
        _layer_shapes(s, f"{p}encoder.layers.{i}", C, cfg.intermediate_size)
    s[p + "post_layernorm.weight"], s[p + "post_layernorm.bias"] = (C,), (C,)
    s["visual_projection.weight"] = (cfg.projection_dim, C)
    return s


def image_proj_shapes(clip_dim: int, cross_dim: int, tokens: int) -> "OrderedDict[str, Tuple[int, ...]]":
    return OrderedDict([("proj.weight", (tokens * cross_dim, clip_dim)), ("proj.bias", (tokens * cross_dim,)),
                        ("norm.weight", (cross_dim,)), ("norm.bias", (cross_dim,))])


def resampler_shapes(cfg: ResamplerConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    inner, D = cfg.dim_head * cfg.heads, cfg.dim
    s["latents"] = (1, cfg.num_queries, D)
    s["proj_in.weight"], s["proj_in.bias"] = (D, cfg.embedding_dim), (D,)
    s["proj_out.weight"], s["proj_out.bias"] = (cfg.output_dim, D), (cfg.output_dim,)
    s["norm_out.weight"], s["norm_out.bias"] = (cfg.output_dim,), (cfg.output_dim,)
    for i in range(cfg.depth):
        a, f = f"layers.{i}.0", f"layers.{i}.1"
        s[a + ".norm1.weight"], s[a + ".norm1.bias"], s[a + ".norm2.weight"], s[a + ".norm2.bias"] = (D,), (D,), (D,), (D,)
        s[a + ".to_q.weight"], s[a + ".to_kv.weight"], s[a + ".to_out.weight"] = (inner, D), (2 * inner, D), (D, inner)
        s[f + ".0.weight"], s[f + ".0.bias"] = (D,), (D,)
        s[f + ".1.weight"], s[f + ".3.weight"] = (D * cfg.ff_mult, D), (D, D * cfg.ff_mult)
    return s


def make_encoder_weights(shapes, seed: int) -> SD:
    """seeded weights: fan-in scaled matrices, LayerNorm gains near 1, small biases / embeddings"""
    g = torch.Generator().manual_seed(seed)
    out: SD = OrderedDict()
    for k, shp in shapes.items():
        if len(shp) == 1 and k.endswith("weight"):
            out[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif len(shp) == 1 or "embedding" in k or k == "latents":
            out[k] = 0.3 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for n in shp[1:]:
                fan_in *= n
            out[k] = torch.randn(shp, generator=g) * (1.5 / math.sqrt(fan_in))
    return out


TINY_TEXT = ClipTextConfig(vocab_size=1000, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4)
TINY_VISION = ClipVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=4, image_size=28,
                               patch_size=14, projection_dim=64)
TINY_RESAMPLER = ResamplerConfig(dim=64, depth=2, dim_head=16, heads=4, num_queries=4, embedding_dim=128, output_dim=64, ff_mult=4)
