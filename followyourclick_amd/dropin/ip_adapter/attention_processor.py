"""`ip_adapter.attention_processor` - the diffusers>=0.17 style attention *processors* of the reference
(reference ip_adapter/attention_processor.py: AttnProcessor :7-77, IPAttnProcessor :80-183, AttnProcessor2_0
:186-272, IPAttnProcessor2_0 :275-400, CNAttnProcessor(2_0) :404-554) with the attention core running on
libfyc_hip.so.

A processor is called as `proc(attn, hidden_states, encoder_hidden_states, attention_mask, temb)` with a
host attention module `attn` that owns to_q/to_k/to_v/to_out and the head count; the projections stay the
host's own modules (that is the protocol), the softmax(QK^T * scale) V core - the part the reference hands to
torch.bmm / F.scaled_dot_product_attention - is the fused MFMA flash kernel (bf16) or, for f32 tensors, the
materialised GEMM+softmax+GEMM parity path.  The bmm ("AttnProcessor") and SDPA ("2_0") flavours of the
reference compute the same function, so they share one implementation here.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from followyourclick_amd import ops as _ops


def fused_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: Optional[float] = None) -> torch.Tensor:
    """softmax(q k^T * scale) v for (B, N, C) / (B, Nk, C) tensors with channel = head*d + i
    (reference attn.head_to_batch_dim / get_attention_scores / batch_to_head_dim)."""
    if not q.is_cuda:
        raise RuntimeError("ip_adapter.attention_processor on the MI355X engine needs HIP device tensors (no CPU fallback)")
    B, N, C = q.shape
    Nk, d = k.shape[1], C // heads
    if d % 8 or d > 160:
        raise NotImplementedError(f"head dim {d}: the attention kernels need a multiple of 8 up to 160")
    scale = d ** -0.5 if scale is None else scale
    o = _ops.get()
    o.ensure_init(q.device)
    dt_in = q.dtype
    dt = dt_in if dt_in in (torch.float32, torch.float16) else torch.bfloat16      # fp16 modules (the reference's deployment) stay fp16
    ld = ((Nk + 7) // 8) * 8
    qh = q.to(dt).reshape(B, N, heads, d).permute(0, 2, 1, 3).contiguous()
    kh = k.to(dt).reshape(B, Nk, heads, d).permute(0, 2, 1, 3).contiguous()
    vt = torch.zeros(B, heads, d, ld, dtype=dt, device=q.device)
    vt[..., :Nk] = v.to(dt).reshape(B, Nk, heads, d).permute(0, 2, 3, 1)
    out = torch.empty(B * N, C, dtype=dt, device=q.device)
    if dt != torch.float32:
        o.attention(qh, kh, vt, out, batch=B, heads=heads, n_q=N, n_k=Nk, d=d, ldo=C, ldvt=ld, scale=scale)
    else:
        for b in range(B):
            S = torch.zeros(heads, N, ld, dtype=dt, device=q.device)
            o.gemm(qh[b], kh[b], S, M=N, N=Nk, K=d, lda=d, ldw=d, ldo=ld, batch=heads, stride_a=N * d, stride_w=Nk * d, stride_o=N * ld,
                   out_scale=scale)
            o.softmax_rows(S, rows=heads * N, cols=Nk, ld=ld)
            o.gemm(S, vt[b], out[b * N:(b + 1) * N], M=N, N=d, K=ld, lda=ld, ldw=ld, ldo=C, batch=heads, stride_a=N * ld, stride_w=d * ld,
                   stride_o=d)
    return out.reshape(B, N, C).to(dt_in)


def _prologue(attn, hidden_states, encoder_hidden_states, attention_mask, temb):
    if attention_mask is not None:
        raise NotImplementedError("attention_mask is not supported by the fused attention kernel (the FollowYourClick path passes None)")
    residual = hidden_states
    if getattr(attn, "spatial_norm", None) is not None:
        hidden_states = attn.spatial_norm(hidden_states, temb)
    shape4 = None
    if hidden_states.ndim == 4:
        b, c, h, w = hidden_states.shape
        shape4 = (b, c, h, w)
        hidden_states = hidden_states.view(b, c, h * w).transpose(1, 2)
    if getattr(attn, "group_norm", None) is not None:
        hidden_states = attn.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
    return residual, hidden_states, shape4


def _epilogue(attn, hidden_states, residual, shape4):
    hidden_states = attn.to_out[0](hidden_states)
    hidden_states = attn.to_out[1](hidden_states)
    if shape4 is not None:
        b, c, h, w = shape4
        hidden_states = hidden_states.transpose(-1, -2).reshape(b, c, h, w)
    if getattr(attn, "residual_connection", False):
        hidden_states = hidden_states + residual
    return hidden_states / getattr(attn, "rescale_output_factor", 1.0)


def _scale(attn):
    return getattr(attn, "scale", None)


class AttnProcessor(nn.Module):
    """plain self/cross attention (reference :7-77 and its SDPA twin :186-272)"""

    def __init__(self, hidden_size=None, cross_attention_dim=None):
        super().__init__()

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        residual, hidden_states, shape4 = _prologue(attn, hidden_states, encoder_hidden_states, attention_mask, temb)
        query = attn.to_q(hidden_states)
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        elif getattr(attn, "norm_cross", False):
            encoder_hidden_states = attn.norm_encoder_hidden_states(encoder_hidden_states)
        out = fused_attention(query, attn.to_k(encoder_hidden_states), attn.to_v(encoder_hidden_states), attn.heads, _scale(attn))
        return _epilogue(attn, out, residual, shape4)


class IPAttnProcessor(nn.Module):
    """decoupled text / image-prompt cross attention: out = attn(q, K_text, V_text) + scale * attn(q, K_ip, V_ip)
    where the last `num_tokens` context tokens are image tokens with their own to_k_ip / to_v_ip
    (reference :80-183; same math as IPCrossAttention, animatediff/models/attention.py:49-127)."""

    def __init__(self, hidden_size, cross_attention_dim=None, scale=1.0, num_tokens=4):
        super().__init__()
        self.hidden_size, self.cross_attention_dim, self.scale, self.num_tokens = hidden_size, cross_attention_dim, scale, num_tokens
        self.to_k_ip = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)
        self.to_v_ip = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        residual, hidden_states, shape4 = _prologue(attn, hidden_states, encoder_hidden_states, attention_mask, temb)
        query = attn.to_q(hidden_states)
        if encoder_hidden_states is None:
            raise ValueError("IPAttnProcessor needs encoder_hidden_states (text tokens followed by the image tokens)")
        end = encoder_hidden_states.shape[1] - self.num_tokens
        text, ip = encoder_hidden_states[:, :end, :], encoder_hidden_states[:, end:, :]
        if getattr(attn, "norm_cross", False):
            text = attn.norm_encoder_hidden_states(text)
        s = _scale(attn)
        out = fused_attention(query, attn.to_k(text), attn.to_v(text), attn.heads, s)
        out_ip = fused_attention(query, self.to_k_ip(ip), self.to_v_ip(ip), attn.heads, s)
        return _epilogue(attn, out + self.scale * out_ip, residual, shape4)


class CNAttnProcessor:
    """ControlNet variant: cross-attend to the text tokens only, dropping the trailing image tokens (reference :404-470)"""

    def __init__(self, num_tokens=4):
        self.num_tokens = num_tokens

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        residual, hidden_states, shape4 = _prologue(attn, hidden_states, encoder_hidden_states, attention_mask, temb)
        query = attn.to_q(hidden_states)
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        else:
            encoder_hidden_states = encoder_hidden_states[:, : encoder_hidden_states.shape[1] - self.num_tokens]
            if getattr(attn, "norm_cross", False):
                encoder_hidden_states = attn.norm_encoder_hidden_states(encoder_hidden_states)
        out = fused_attention(query, attn.to_k(encoder_hidden_states), attn.to_v(encoder_hidden_states), attn.heads, _scale(attn))
        return _epilogue(attn, out, residual, shape4)


# the "2_0" (F.scaled_dot_product_attention) flavours compute the same function
AttnProcessor2_0 = AttnProcessor
IPAttnProcessor2_0 = IPAttnProcessor
CNAttnProcessor2_0 = CNAttnProcessor
