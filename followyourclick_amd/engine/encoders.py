"""Conditioning encoders on the HIP ops: CLIP text encoder, CLIP vision tower, IP-Adapter image
projection (ImageProjModel) and Resampler.  They run once per clip in front of the sampling loop.

What they replace in the reference:
  * `self.text_encoder(input_ids, attention_mask=None)[0]` - animatediff/pipelines/pipeline_animation.py:183-186, 229-232
    (transformers.CLIPTextModel: token+position embedding, 12 pre-LN layers with a causal mask, final LayerNorm);
  * `self.image_encoder(pixels).image_embeds` / `.hidden_states[-2]` - ip_adapter/my_ip_adapter.py:132, 280-283
    (transformers.CLIPVisionModelWithProjection: 14x14 patch embedding, class token, pre_layrnorm, pre-LN layers,
    post_layernorm + visual_projection of the class token);
  * `ImageProjModel.forward` - ip_adapter/my_ip_adapter.py:39-45;  `Resampler.forward` - ip_adapter/resampler.py:125-147.

Op schedule per transformer layer: LayerNorm kernel -> fused QKV GEMM (bias, head-split epilogue) -> per batch element the
materialised attention (batched QK^T GEMM with the scale folded in, row-softmax kernel - causal for text -, batched PV GEMM;
sequences are 77 / 257 / 273 tokens, far too short for the flash kernel to matter) -> out-proj GEMM (+bias +residual) ->
LayerNorm -> fc1 GEMM with the activation in its epilogue -> fc2 GEMM (+bias +residual).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from .. import _lib as L
from .base import EngineBase
from .weights import Packed, f32, pack_linear

Tensor = torch.Tensor
ACTS = {"quick_gelu": L.ACT_QUICK_GELU, "gelu": L.ACT_GELU}


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


def _pad8(n: int) -> int:
    return (n + 7) // 8 * 8


def _check_dims(C: int, heads: int, what: str) -> None:
    if C % 8 or C % heads or (C // heads) % 8:
        raise ValueError(f"{what}: hidden size {C} / heads {heads}: width and head dim must be multiples of 8")


# ---- weight packing -----------------------------------------------------------------------------------------------------
def _ln(sd, p, device):
    return (f32(sd[p + ".weight"], device), f32(sd[p + ".bias"], device))


def _clip_layer(sd: Dict[str, Tensor], p: str, dtype, device) -> Packed:
    qkv_w = torch.cat([sd[f"{p}.self_attn.{n}_proj.weight"] for n in ("q", "k", "v")])
    qkv_b = torch.cat([sd[f"{p}.self_attn.{n}_proj.bias"] for n in ("q", "k", "v")])
    return Packed(ln1=_ln(sd, p + ".layer_norm1", device), ln2=_ln(sd, p + ".layer_norm2", device),
                  qkv_w=pack_linear(qkv_w, dtype, device), qkv_b=f32(qkv_b, device),
                  o_w=pack_linear(sd[p + ".self_attn.out_proj.weight"], dtype, device), o_b=f32(sd[p + ".self_attn.out_proj.bias"], device),
                  fc1_w=pack_linear(sd[p + ".mlp.fc1.weight"], dtype, device), fc1_b=f32(sd[p + ".mlp.fc1.bias"], device),
                  fc2_w=pack_linear(sd[p + ".mlp.fc2.weight"], dtype, device), fc2_b=f32(sd[p + ".mlp.fc2.bias"], device))


def pack_clip_text(sd: Dict[str, Tensor], cfg: ClipTextConfig, dtype, device) -> Packed:
    """state dict of transformers.CLIPTextModel (with or without the `text_model.` prefix of transformers-4 checkpoints)"""
    sd = {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in sd.items()}
    _check_dims(cfg.hidden_size, cfg.num_attention_heads, "CLIP text encoder")
    return Packed(cfg=cfg, dtype=dtype, device=torch.device(device),
                  tok=f32(sd["embeddings.token_embedding.weight"], device), pos=f32(sd["embeddings.position_embedding.weight"], device),
                  layers=[_clip_layer(sd, f"encoder.layers.{i}", dtype, device) for i in range(cfg.num_hidden_layers)],
                  final_ln=_ln(sd, "final_layer_norm", device))


def pack_clip_vision(sd: Dict[str, Tensor], cfg: ClipVisionConfig, dtype, device) -> Packed:
    """state dict of transformers.CLIPVisionModelWithProjection (`vision_model.*`, `visual_projection.weight`)"""
    p = "vision_model."
    _check_dims(cfg.hidden_size, cfg.num_attention_heads, "CLIP vision tower")
    C = cfg.hidden_size
    pos = sd[p + "embeddings.position_embedding.weight"].float()
    k_pad = _pad8(3 * cfg.patch_size ** 2)
    has_proj = "visual_projection.weight" in sd
    return Packed(cfg=cfg, dtype=dtype, device=torch.device(device), k_pad=k_pad,
                  patch_w=pack_linear(sd[p + "embeddings.patch_embedding.weight"], dtype, device, k_pad=k_pad),
                  cls_pos0=f32((sd[p + "embeddings.class_embedding"].float() + pos[0]).reshape(1, C), device),
                  pos_rest=pos[1:].to(dtype).contiguous().to(device),
                  pre_ln=_ln(sd, p + "pre_layrnorm", device), post_ln=_ln(sd, p + "post_layernorm", device),
                  layers=[_clip_layer(sd, f"{p}encoder.layers.{i}", dtype, device) for i in range(cfg.num_hidden_layers)],
                  proj_w=pack_linear(sd["visual_projection.weight"], dtype, device) if has_proj else None)


def pack_image_proj(sd: Dict[str, Tensor], dtype, device) -> Packed:
    return Packed(dtype=dtype, device=torch.device(device), w=pack_linear(sd["proj.weight"], dtype, device), b=f32(sd["proj.bias"], device),
                  ln=_ln(sd, "norm", device), cross_dim=sd["norm.weight"].shape[0])


def pack_resampler(sd: Dict[str, Tensor], cfg: ResamplerConfig, dtype, device) -> Packed:
    _check_dims(cfg.dim_head * cfg.heads, cfg.heads, "Resampler")
    if cfg.dim % 8 or cfg.embedding_dim % 8 or cfg.output_dim % 8:
        raise ValueError("Resampler widths must be multiples of 8")
    layers = []
    for i in range(cfg.depth):
        a, f = f"layers.{i}.0", f"layers.{i}.1"
        layers.append(Packed(n1=_ln(sd, a + ".norm1", device), n2=_ln(sd, a + ".norm2", device),
                             q_w=pack_linear(sd[a + ".to_q.weight"], dtype, device), kv_w=pack_linear(sd[a + ".to_kv.weight"], dtype, device),
                             o_w=pack_linear(sd[a + ".to_out.weight"], dtype, device), ff_ln=_ln(sd, f + ".0", device),
                             ff1=pack_linear(sd[f + ".1.weight"], dtype, device), ff2=pack_linear(sd[f + ".3.weight"], dtype, device)))
    return Packed(cfg=cfg, dtype=dtype, device=torch.device(device), latents=sd["latents"].reshape(cfg.num_queries, cfg.dim).to(dtype).contiguous().to(device),
                  pin_w=pack_linear(sd["proj_in.weight"], dtype, device), pin_b=f32(sd["proj_in.bias"], device),
                  pout_w=pack_linear(sd["proj_out.weight"], dtype, device), pout_b=f32(sd["proj_out.bias"], device),
                  out_ln=_ln(sd, "norm_out", device), layers=layers)


# ---- engines -------------------------------------------------------------------------------------------------------------
class _EncoderBase(EngineBase):
    groups = 32

    def __init__(self, packed: Packed, ops=None):
        if ops is None:
            from .. import ops as _ops
            ops = _ops.get()
        self.ops, self.P = ops, packed
        self.dtype, self.device = packed.dtype, packed.device

    def ln(self, x: Tensor, gb, rows: int, C: int, eps: float = 1e-5, out: Optional[Tensor] = None) -> Tensor:
        y = out if out is not None else self.new(rows, C)
        self.ops.layernorm(x, gb[0], gb[1], y, rows=rows, C_=C, eps=eps)
        return y

    def attention(self, q: Tensor, k: Tensor, vt: Tensor, out: Tensor, *, B: int, H: int, n_q: int, n_k: int, d: int, ldk: int,
                  C: int, causal: bool = False) -> None:
        """softmax(q k^T d^-1/2 [+causal mask]) v per batch element; q (B,H,n_q,d), k (B,H,n_k,d), vt (B,H,d,ldk) -> out rows (B*n_q, C)"""
        o = self.ops
        for b in range(B):
            S = self.zeros(H, n_q, ldk)
            o.gemm(q[b], k[b], S, M=n_q, N=n_k, K=d, lda=d, ldw=d, ldo=ldk, batch=H, stride_a=n_q * d, stride_w=n_k * d,
                   stride_o=n_q * ldk, out_scale=d ** -0.5)
            o.softmax_rows(S, rows=H * n_q, cols=n_k, ld=ldk, causal_rows=n_q if causal else 0)
            o.gemm(S, vt[b], out[b * n_q:(b + 1) * n_q], M=n_q, N=d, K=ldk, lda=ldk, ldw=ldk, ldo=C, batch=H, stride_a=n_q * ldk,
                   stride_w=d * ldk, stride_o=d)

    def clip_layer(self, lw: Packed, x: Tensor, B: int, N: int, C: int, H: int, act: int, eps: float, causal: bool) -> Tensor:
        rows, d, ld, o = B * N, C // H, _pad8(N), self.ops
        h = self.ln(x, lw.ln1, rows, C, eps)
        q, k = self.new(B, H, N, d), self.new(B, H, N, d)
        vt = self.zeros(B, H, d, ld) if ld != N else self.new(B, H, d, ld)
        o.gemm(h, lw.qkv_w, None, M=rows, N=3 * C, K=C, lda=C, ldw=C, bias=lw.qkv_b, epilogue=L.EPI_HEADS,
               heads=dict(seg_cols=C, heads=H, tokens=N, outs=[q, k, vt], transposed=[0, 0, 1], ld=[0, 0, ld]))
        att = self.new(rows, C)
        self.attention(q, k, vt, att, B=B, H=H, n_q=N, n_k=N, d=d, ldk=ld, C=C, causal=causal)
        x = self.lin(att, lw.o_w, rows, bias=lw.o_b, residual=x)
        h = self.ln(x, lw.ln2, rows, C, eps)
        inter = lw.fc1_w.shape[0]
        mid = self.new(rows, inter)
        o.gemm(h, lw.fc1_w, mid, M=rows, N=inter, K=C, lda=C, ldw=C, ldo=inter, bias=lw.fc1_b, act=act)
        return self.lin(mid, lw.fc2_w, rows, bias=lw.fc2_b, residual=x)

    def to_f32(self, x: Tensor, rows: int, C: int) -> Tensor:
        y = torch.empty(rows, C, dtype=torch.float32, device=self.device)
        self.ops.cast_to_f32(x, y, rows=rows, cols=C, ld=C)
        return y


class ClipTextEngine(_EncoderBase):
    @torch.no_grad()
    def encode(self, input_ids: Tensor) -> Tensor:
        """(B, N<=77) int64 token ids -> last_hidden_state (B, N, hidden) f32"""
        P, cfg = self.P, self.P.cfg
        if input_ids.dim() != 2 or input_ids.shape[1] > cfg.max_position_embeddings:
            raise ValueError(f"input_ids must be (batch, <= {cfg.max_position_embeddings}), got {tuple(input_ids.shape)}")
        ids = input_ids.to(self.device, torch.int64).contiguous()
        if int(ids.min()) < 0 or int(ids.max()) >= cfg.vocab_size:
            raise IndexError(f"token id outside [0, {cfg.vocab_size})")
        B, N = ids.shape
        C, H = cfg.hidden_size, cfg.num_attention_heads
        x = self.new(B * N, C)
        self.ops.embed_tokens(ids, P.tok, P.pos, x, rows=B * N, seq=N, C_=C)
        for lw in P.layers:
            x = self.clip_layer(lw, x, B, N, C, H, ACTS[cfg.hidden_act], cfg.layer_norm_eps, causal=True)
        x = self.ln(x, P.final_ln, B * N, C, cfg.layer_norm_eps)
        return self.to_f32(x, B * N, C).reshape(B, N, C)


class ClipVisionEngine(_EncoderBase):
    @torch.no_grad()
    def encode(self, pixel_values: Tensor, want: Tuple[str, ...] = ("image_embeds",)) -> Dict[str, Tensor]:
        """(B,3,S,S) preprocessed pixels -> {"image_embeds": (B, proj) f32, "penultimate": hidden_states[-2] (B, 1+n, C) f32,
        "last": hidden_states[-1]} - only the entries named in `want` are produced"""
        P, cfg, o = self.P, self.P.cfg, self.ops
        B, Cin, Hh, Ww = pixel_values.shape
        if Cin != 3 or Hh != cfg.image_size or Ww != cfg.image_size:
            raise ValueError(f"pixel_values must be (B, 3, {cfg.image_size}, {cfg.image_size}), got {tuple(pixel_values.shape)}")
        img = pixel_values.to(self.device, torch.float32).contiguous()
        C, H, g = cfg.hidden_size, cfg.num_attention_heads, cfg.image_size // cfg.patch_size
        n, N = g * g, g * g + 1
        patches = self.new(B * n, P.k_pad)
        o.patchify(img, patches, B=B, Cin=3, H=Hh, W=Ww, P=cfg.patch_size, ld=P.k_pad)
        x = self.new(B * N, C)
        # patch rows 1..n of every image = patches @ W^T + position embedding (residual shared by the batch); row 0 = class + pos[0]
        o.gemm(patches, P.patch_w, x[1:], M=n, N=C, K=P.k_pad, lda=P.k_pad, ldw=P.k_pad, ldo=C, residual=P.pos_rest, ldr=C, batch=B,
               stride_a=n * P.k_pad, stride_w=0, stride_o=N * C)
        for b in range(B):
            o.cast_from_f32(P.cls_pos0, x[b * N:b * N + 1], rows=1, cols=C, ld=C)
        x = self.ln(x, P.pre_ln, B * N, C, cfg.layer_norm_eps)
        out: Dict[str, Tensor] = {}
        L_ = len(P.layers)
        for i, lw in enumerate(P.layers):
            if i == L_ - 1 and "penultimate" in want:
                out["penultimate"] = self.to_f32(x, B * N, C).reshape(B, N, C)
            if i == L_ - 1 and want == ("penultimate",):
                return out                         # the last layer only feeds image_embeds / last
            x = self.clip_layer(lw, x, B, N, C, H, ACTS[cfg.hidden_act], cfg.layer_norm_eps, causal=False)
        if "last" in want:
            out["last"] = self.to_f32(x, B * N, C).reshape(B, N, C)
        if "image_embeds" in want:
            if P.proj_w is None:
                raise ValueError("this vision tower has no visual_projection (load CLIPVisionModelWithProjection weights)")
            cls = self.new(B, C)
            for b in range(B):                     # gather the class rows (B is 1-2): LayerNorm row by row into a dense buffer
                self.ln(x[b * N:b * N + 1], P.post_ln, 1, C, cfg.layer_norm_eps, out=cls[b:b + 1])
            emb = self.lin(cls, P.proj_w, B)
            out["image_embeds"] = self.to_f32(emb, B, P.proj_w.shape[0])
        return out


class ImageProjEngine(_EncoderBase):
    @torch.no_grad()
    def project(self, image_embeds: Tensor) -> Tensor:
        """(B, clip_dim) -> (B, tokens, cross_dim) f32"""
        P = self.P
        B, K = image_embeds.shape
        if K != P.w.shape[1]:
            raise ValueError(f"image_embeds has {K} features, the projection expects {P.w.shape[1]}")
        x = self.new(B, K)
        self.ops.cast_from_f32(image_embeds.to(self.device, torch.float32).contiguous(), x, rows=B, cols=K, ld=K)
        t = self.lin(x, P.w, B, bias=P.b)
        D = P.cross_dim
        tokens = P.w.shape[0] // D
        y = self.ln(t.view(B * tokens, D), P.ln, B * tokens, D)
        return self.to_f32(y, B * tokens, D).reshape(B, tokens, D)


class ResamplerEngine(_EncoderBase):
    @torch.no_grad()
    def resample(self, feats: Tensor) -> Tensor:
        """CLIP hidden states (B, n1, embedding_dim) -> (B, num_queries, output_dim) f32"""
        P, cfg, o = self.P, self.P.cfg, self.ops
        B, n1, E = feats.shape
        if E != cfg.embedding_dim:
            raise ValueError(f"features have width {E}, the resampler expects {cfg.embedding_dim}")
        D, H, d, nq = cfg.dim, cfg.heads, cfg.dim_head, cfg.num_queries
        inner, nk = H * d, n1 + nq
        ldk = _pad8(nk)
        xin = self.new(B * n1, E)
        o.cast_from_f32(feats.to(self.device, torch.float32).reshape(B * n1, E).contiguous(), xin, rows=B * n1, cols=E, ld=E)
        x = self.lin(xin, P.pin_w, B * n1, bias=P.pin_b)
        lat = P.latents.repeat(B, 1)
        for lw in P.layers:
            # kv input = cat(LN1(x), LN2(latents)) along tokens, per batch element
            kv_in, lnl = self.new(B * nk, D), self.new(B * nq, D)
            self.ln(lat, lw.n2, B * nq, D, out=lnl)
            for b in range(B):
                self.ln(x[b * n1:(b + 1) * n1], lw.n1, n1, D, out=kv_in[b * nk:b * nk + n1])
                self.ln(lat[b * nq:(b + 1) * nq], lw.n2, nq, D, out=kv_in[b * nk + n1:(b + 1) * nk])
            q, k = self.new(B, H, nq, d), self.new(B, H, nk, d)
            vt = self.zeros(B, H, d, ldk) if ldk != nk else self.new(B, H, d, ldk)
            o.gemm(lnl, lw.q_w, None, M=B * nq, N=inner, K=D, lda=D, ldw=D, epilogue=L.EPI_HEADS,
                   heads=dict(seg_cols=inner, heads=H, tokens=nq, outs=[q], transposed=[0], ld=[0]))
            o.gemm(kv_in, lw.kv_w, None, M=B * nk, N=2 * inner, K=D, lda=D, ldw=D, epilogue=L.EPI_HEADS,
                   heads=dict(seg_cols=inner, heads=H, tokens=nk, outs=[k, vt], transposed=[0, 1], ld=[0, ldk]))
            att = self.new(B * nq, inner)
            self.attention(q, k, vt, att, B=B, H=H, n_q=nq, n_k=nk, d=d, ldk=ldk, C=inner)
            lat = self.lin(att, lw.o_w, B * nq, residual=lat)
            h = self.ln(lat, lw.ff_ln, B * nq, D)
            mid = self.new(B * nq, lw.ff1.shape[0])
            o.gemm(h, lw.ff1, mid, M=B * nq, N=lw.ff1.shape[0], K=D, lda=D, ldw=D, ldo=lw.ff1.shape[0], act=L.ACT_GELU)
            lat = self.lin(mid, lw.ff2, B * nq, residual=lat)
        y = self.lin(lat, P.pout_w, B * nq, bias=P.pout_b)
        y = self.ln(y, P.out_ln, B * nq, cfg.output_dim)
        return self.to_f32(y, B * nq, cfg.output_dim).reshape(B, nq, cfg.output_dim)
