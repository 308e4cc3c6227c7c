"""Weight packing: reference-layout state dicts -> kernel-friendly device buffers (done once at load).

Reference layouts: conv OIHW, linear (out, in), 1x1 conv (O, I, 1, 1); see SURVEY.md 8a'' for the key
schema (animatediff/models/unet.py, unet_blocks.py, attention.py, motion_module.py state dicts).
Packed layouts (activation dtype unless noted):
  conv3x3   [O][slab][ky][kx][c]  -> (O, 9*I_pad): K order matches the implicit-GEMM gather (gemm_kernel.h)
  linear    (O, I) row-major = the GEMM's W[n][k] operand as is
  qkv       to_q | to_k | to_v stacked on O -> one GEMM per attention block
  GEGLU     ff.net.0.proj rows interleaved in blocks of 16 (value rows 16b.., then their gate rows)
            so that the value and its gate land in the same lane of the MFMA epilogue
  biases, norm affine parameters, positional tables, time-embedding MLPs: f32
"""
from __future__ import annotations

import os

from typing import Dict, Optional

import torch

from .config import UNet3DConfig, VAEDecoderConfig

Tensor = torch.Tensor
K_ALIGN = 64  # channels are padded to this so every conv K tile (128 B) stays inside one filter tap


def pad_channels(c: int) -> int:
    return ((c + K_ALIGN - 1) // K_ALIGN) * K_ALIGN


def conv_slab(dtype) -> int:
    """channels per 128-byte K tile of the implicit-GEMM conv (64 bf16 / 32 f32)"""
    return 128 // torch.empty((), dtype=dtype).element_size()


def pack_conv3x3(w: Tensor, dtype, device) -> Tensor:
    """OIHW -> [O][slab][ky][kx][c in slab]: K order of gemm_kernel.h (the 9 taps of a slab are adjacent K tiles)"""
    O, I, kh, kw = w.shape
    assert kh == 3 and kw == 3
    Ip, sl = pad_channels(I), conv_slab(dtype)
    p = torch.zeros(O, Ip, 3, 3, dtype=torch.float32)
    p[:, :I] = w
    p = p.reshape(O, Ip // sl, sl, 3, 3).permute(0, 1, 3, 4, 2)
    return p.reshape(O, 9 * Ip).to(dtype).contiguous().to(device)


def pack_linear(w: Tensor, dtype, device, k_pad: Optional[int] = None) -> Tensor:
    w = w.reshape(w.shape[0], -1)  # also accepts (O, I, 1, 1)
    if k_pad is not None and k_pad != w.shape[1]:
        p = torch.zeros(w.shape[0], k_pad, dtype=torch.float32)
        p[:, : w.shape[1]] = w
        w = p
    return w.to(dtype).contiguous().to(device)


def pack_geglu(w: Tensor, b: Tensor, dtype, device):
    """ff.net.0.proj: rows [0,4C) = value, [4C,8C) = gate (reference diffusers/models/attention.py:819-821)."""
    O, I = w.shape
    half = O // 2
    assert half % 16 == 0
    wv, wg = w[:half].reshape(half // 16, 16, I), w[half:].reshape(half // 16, 16, I)
    bv, bg = b[:half].reshape(half // 16, 16), b[half:].reshape(half // 16, 16)
    wp = torch.stack([wv, wg], dim=1).reshape(O, I)
    bp = torch.stack([bv, bg], dim=1).reshape(O)
    return wp.to(dtype).contiguous().to(device), bp.float().contiguous().to(device)


def f32(t: Tensor, device) -> Tensor:
    return t.detach().float().contiguous().to(device)


def fold_layernorm(w: Tensor, b: Optional[Tensor], gamma: Tensor, beta: Tensor, dtype, device, geglu: bool = False):
    """LayerNorm folded into the Linear that consumes it (fyc_gemm ln_stats / ln_colsum):
        LN(x) W^T + b = rstd * (x (W*gamma)^T - mean * colsum) + (W beta + b)
    Returns (packed W*gamma, f32 bias W beta + b, f32 colsum).  colsum is taken from the ROUNDED packed weights so that the
    mean term cancels against exactly what the MFMAs accumulate."""
    w = w.reshape(w.shape[0], -1).double()
    wf = (w * gamma.double()[None, :]).float()
    bias = (w @ beta.double() + (b.double() if b is not None else 0.0)).float()
    if geglu:
        wp, bp = pack_geglu(wf, bias, dtype, device)
    else:
        wp, bp = pack_linear(wf, dtype, device), f32(bias, device)
    return wp, bp, wp.float().sum(dim=1).contiguous()

LN_FOLD = os.environ.get("FYC_LN_FOLD", "1") != "0"   # A/B switch: 0 keeps the separate LayerNorm kernel


class Packed(dict):
    """dict with attribute access; leaves are device tensors."""
    __getattr__ = dict.__getitem__


def _resnet(sd: Dict[str, Tensor], p: str, dtype, device, temb: bool = True) -> Packed:
    r = Packed(
        n1_g=f32(sd[p + ".norm1.weight"], device), n1_b=f32(sd[p + ".norm1.bias"], device),
        c1_w=pack_conv3x3(sd[p + ".conv1.weight"], dtype, device), c1_b=f32(sd[p + ".conv1.bias"], device),
        n2_g=f32(sd[p + ".norm2.weight"], device), n2_b=f32(sd[p + ".norm2.bias"], device),
        c2_w=pack_conv3x3(sd[p + ".conv2.weight"], dtype, device), c2_b=f32(sd[p + ".conv2.bias"], device),
        cin=sd[p + ".conv1.weight"].shape[1], cout=sd[p + ".conv1.weight"].shape[0], sc_w=None, sc_b=None)
    if (p + ".conv_shortcut.weight") in sd:
        r["sc_w"] = pack_linear(sd[p + ".conv_shortcut.weight"], dtype, device)
        r["sc_b"] = f32(sd[p + ".conv_shortcut.bias"], device)
    return r


def _ff(sd, p, dtype, device, proj_out_w: Tensor, proj_out_b: Tensor, ln: Optional[str] = None) -> Packed:
    """GEGLU feed-forward followed by the block's output projection.

    The reference computes tok' = tok + W2 h + b2 (FeedForward, diffusers/models/attention.py:772-775) and then
    out = x + Wp tok' + bp (Transformer3DModel.proj_out attention.py:291-304 / TemporalTransformer3DModel.proj_out
    motion_module.py:199-203); tok' is used nowhere else.  Both are linear, so they are merged at load time into ONE
    GEMM over the K-concatenated operand [tok | h]:  out = x + (bp + Wp b2) + [Wp | Wp W2] [tok ; h]
    (products in f32, then cast) - one launch and one full pass over the activations less per block."""
    cs1 = None
    if ln is not None and LN_FOLD:      # the LayerNorm in front of the FF (norm3 / ff_norm) is folded into FF1
        w1, b1, cs1 = fold_layernorm(sd[p + ".net.0.proj.weight"], sd[p + ".net.0.proj.bias"], sd[ln + ".weight"], sd[ln + ".bias"],
                                     dtype, device, geglu=True)
    else:
        w1, b1 = pack_geglu(sd[p + ".net.0.proj.weight"], sd[p + ".net.0.proj.bias"], dtype, device)
    w2, b2 = sd[p + ".net.2.weight"].double(), sd[p + ".net.2.bias"].double()
    wp, bp = proj_out_w.reshape(proj_out_w.shape[0], -1).double(), proj_out_b.double()
    merged_w = torch.cat([wp, wp @ w2], dim=1).float()
    merged_b = (bp + wp @ b2).float()
    return Packed(w1=w1, b1=b1, cs1=cs1, po_w=pack_linear(merged_w, dtype, device), po_b=f32(merged_b, device))


def _ln(sd, p, device):
    return f32(sd[p + ".weight"], device), f32(sd[p + ".bias"], device)


def _transformer(sd, p: str, cfg: UNet3DConfig, dtype, device) -> Packed:
    t = p + ".transformer_blocks.0"
    a1, a2 = t + ".attn1", t + ".attn2"
    d = Packed(
        C=sd[p + ".proj_in.weight"].shape[0],
        norm_g=f32(sd[p + ".norm.weight"], device), norm_b=f32(sd[p + ".norm.bias"], device),
        pin_w=pack_linear(sd[p + ".proj_in.weight"], dtype, device), pin_b=f32(sd[p + ".proj_in.bias"], device),
        ln1=_ln(sd, t + ".norm1", device), ln2=_ln(sd, t + ".norm2", device), ln3=_ln(sd, t + ".norm3", device),
        qkv_w=pack_linear(torch.cat([sd[a1 + ".to_q.weight"], sd[a1 + ".to_k.weight"], sd[a1 + ".to_v.weight"]], 0), dtype, device),
        o1_w=pack_linear(sd[a1 + ".to_out.0.weight"], dtype, device), o1_b=f32(sd[a1 + ".to_out.0.bias"], device),
        q2_w=pack_linear(sd[a2 + ".to_q.weight"], dtype, device),
        kv2_w=pack_linear(torch.cat([sd[a2 + ".to_k.weight"], sd[a2 + ".to_v.weight"]], 0), dtype, device),
        o2_w=pack_linear(sd[a2 + ".to_out.0.weight"], dtype, device), o2_b=f32(sd[a2 + ".to_out.0.bias"], device),
        ff=_ff(sd, t + ".ff", dtype, device, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"], ln=t + ".norm3"), kvip_w=None,
        qkv_f=None, q2_f=None)
    if LN_FOLD:   # norm1 -> fused QKV, norm2 -> to_q of the cross-attention
        qkv = torch.cat([sd[a1 + ".to_q.weight"], sd[a1 + ".to_k.weight"], sd[a1 + ".to_v.weight"]], 0)
        d["qkv_f"] = fold_layernorm(qkv, None, sd[t + ".norm1.weight"], sd[t + ".norm1.bias"], dtype, device)
        d["q2_f"] = fold_layernorm(sd[a2 + ".to_q.weight"], None, sd[t + ".norm2.weight"], sd[t + ".norm2.bias"], dtype, device)
        d["qkv_w"] = d["q2_w"] = None
    if cfg.use_ip_cross_attention:
        d["kvip_w"] = pack_linear(torch.cat([sd[a2 + ".to_k_ip.weight"], sd[a2 + ".to_v_ip.weight"]], 0), dtype, device)
    return d


def _motion(sd, p: str, cfg: UNet3DConfig, dtype, device) -> Packed:
    p = p + ".temporal_transformer"
    C = sd[p + ".proj_in.weight"].shape[0]
    blocks = []
    for b in range(cfg.motion_num_transformer_block):
        t = f"{p}.transformer_blocks.{b}"
        attns = []
        for a in range(cfg.motion_attention_blocks):
            ab = f"{t}.attention_blocks.{a}"
            pe = None
            if cfg.temporal_position_encoding:
                # analytic table (== the persistent `pos_encoder.pe` buffer, reference motion_module.py:286-304);
                # regenerated so that clips longer than a checkpoint's max_len still work
                pe = sinusoidal_pe(C, max(cfg.temporal_position_encoding_max_len, 32)).to(device)
            qkv = torch.cat([sd[ab + ".to_q.weight"], sd[ab + ".to_k.weight"], sd[ab + ".to_v.weight"]], 0)
            att = Packed(ln=_ln(sd, f"{t}.norms.{a}", device), pe=pe, qkv_w=None, qkv_f=None, pe_w=None,
                         o_w=pack_linear(sd[ab + ".to_out.0.weight"], dtype, device), o_b=f32(sd[ab + ".to_out.0.bias"], device))
            if LN_FOLD:
                # (LN(x) + pe_f) W^T = LN(x) W^T + pe_f W^T: the positional table becomes a per-frame row bias of the QKV GEMM
                att["qkv_f"] = fold_layernorm(qkv, None, sd[f"{t}.norms.{a}.weight"], sd[f"{t}.norms.{a}.bias"], dtype, device)
                if pe is not None:
                    att["pe_w"] = (pe.double().cpu() @ qkv.double().t()).float().contiguous().to(device)
            else:
                att["qkv_w"] = pack_linear(qkv, dtype, device)
            attns.append(att)
        last = b == cfg.motion_num_transformer_block - 1
        if last:   # the module's proj_out is merged into the last block's FF (see _ff)
            ff = _ff(sd, t + ".ff", dtype, device, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"], ln=t + ".ff_norm")
        else:      # inner blocks keep a plain FF2 + residual: identity "projection"
            ff = _ff(sd, t + ".ff", dtype, device, torch.eye(C), torch.zeros(C), ln=t + ".ff_norm")
        blocks.append(Packed(attns=attns, ff_ln=_ln(sd, t + ".ff_norm", device), ff=ff))
    return Packed(C=C, norm_g=f32(sd[p + ".norm.weight"], device), norm_b=f32(sd[p + ".norm.bias"], device),
                  pin_w=pack_linear(sd[p + ".proj_in.weight"], dtype, device), pin_b=f32(sd[p + ".proj_in.bias"], device),
                  blocks=blocks)


def pack_temporal_block(att: Packed, heads: int, frames: int) -> dict:
    """per-head operands of fyc_temporal_block (csrc/temporal_block.hip) from a packed temporal attention block with the
    LayerNorm folded in (`qkv_f`): w_qkv [H][128][C] (q | k | v rows of the head, zero padded), colsum / bias [H][128],
    pe_bias [F][H][128], w_out [H][C][48] (Wo[:, head's 40 columns], zero padded), b_out [C]"""
    w, b, cs = att.qkv_f
    C = w.shape[1]
    d = C // heads
    assert 3 * d <= 128 and d <= 48

    def per_head(t: Tensor) -> Tensor:          # leading axis 3C = [q | k | v] x (head, d)  ->  [H][128][...]
        out = torch.zeros(heads, 128, *t.shape[1:], dtype=t.dtype, device=t.device)
        for i in range(3):
            out[:, i * d:(i + 1) * d] = t[i * C:(i + 1) * C].reshape(heads, d, *t.shape[1:])
        return out.contiguous()

    pe = None
    if att.pe_w is not None:
        if frames > att.pe_w.shape[0]:
            raise ValueError(f"video_length {frames} exceeds the positional table ({att.pe_w.shape[0]})")
        pe = per_head(att.pe_w[:frames].t().contiguous()).permute(2, 0, 1).contiguous()       # [F][H][128]
    wo = torch.zeros(heads, C, 48, dtype=att.o_w.dtype, device=att.o_w.device)
    wo[:, :, :d] = att.o_w.reshape(C, heads, d).permute(1, 0, 2)
    ops = dict(w_qkv=per_head(w), colsum=per_head(cs), bias=per_head(b), pe_bias=pe, w_out=wo.contiguous(), b_out=att.o_b)
    if frames == 16 and C % 32 == 0:
        ops["wstream"] = pack_temporal_stream(ops["w_qkv"], ops["bias"], pe, ops["w_out"], d)
    return ops


def temporal_block_layout(C: int) -> dict:
    """piece indices of the two stages per head of the fyc_temporal_block weight stream (include/fyc.h); at C = 320: stage A =
    60 q/k fragments + 6 table pieces, stage B = 30 v fragments + 40 Wo' fragments + 3 table pieces, stride 73 pieces"""
    ks, nb = C // 32, C // 16
    lay = dict(ks=ks, nb=nb, a_tab=6 * ks, a_pieces=6 * ks + 6, b_wo=3 * ks, b_tab=3 * ks + 2 * nb, b_pieces=3 * ks + 2 * nb + 3)
    lay["stride"] = max(lay["a_pieces"], lay["b_pieces"])
    return lay


def temporal_slot_feature(t: int, k: int) -> int:
    """head feature that k-slot k (0..31) of k-step t of the attention output operand carries in csrc/temporal_block_rr.hip (a lane
    holds 4 consecutive features of each 16-feature block where an MFMA operand wants 8 consecutive k), 48 = none (zero)"""
    g, e = k // 8, k % 8
    if t == 0:
        return 4 * g + e if e < 4 else 16 + 4 * g + e - 4
    return 32 + 4 * g + e if e < 4 else 48


def pack_temporal_stream(w_qkv: Tensor, bias: Tensor, pe_bias: Optional[Tensor], w_out: Tensor, d: int) -> Tensor:
    """weight stream of the register-resident fyc_temporal_block (csrc/temporal_block_rr.hip; layout: include/fyc.h) from the
    per-head operands above: w_qkv [H][128][C] (q | k | v rows of the head), bias [H][128], pe_bias [16][H][128] or None,
    w_out [H][C][48].  Two stages per head, 1-KiB MFMA fragments; the bias tables are f32 bytes inside the stream."""
    H, _, C = w_qkv.shape
    F = 16
    assert d <= 48 and C % 32 == 0 and (pe_bias is None or pe_bias.shape[0] == F)
    lay = temporal_block_layout(C)
    ks, nb = lay["ks"], lay["nb"]
    T, dev = w_qkv.dtype, w_qkv.device
    st = torch.zeros(2 * H, lay["stride"], 512, dtype=T, device=dev)
    qkv = torch.zeros(H, 3, 48, C, dtype=T, device=dev)
    for i in range(3):
        qkv[:, i, :d] = w_qkv[:, i * d:(i + 1) * d]
    # stage A: piece s * 6 + (which * 3 + b) = rows 16 b .. +16 of q (which = 0) / k (1), columns 32 s .. +32
    blk = qkv[:, :2].reshape(H, 2, 3, 16, ks, 32).permute(0, 4, 1, 2, 3, 5).reshape(H, ks * 6, 16, 32)
    st[0::2, :6 * ks] = _mfma_fragments(blk)
    tb = bias.float()[:, None, :] + (pe_bias.float().permute(1, 0, 2) if pe_bias is not None else 0.0)      # [H][F][128]
    tb = tb.expand(H, F, 128)
    ta = torch.zeros(H, F, 96, dtype=torch.float32, device=dev)
    ta[:, :, :d] = tb[:, :, :d]
    ta[:, :, 48:48 + d] = tb[:, :, d:2 * d]
    st[0::2, lay["a_tab"]:lay["a_tab"] + 6] = ta.reshape(H, -1).contiguous().view(T).reshape(H, 6, 512)
    # stage B: v fragments, Wo' fragments in the k-slot order of the attention output, v table [48][F]
    blk = qkv[:, 2].reshape(H, 3, 16, ks, 32).permute(0, 3, 1, 2, 4).reshape(H, ks * 3, 16, 32)
    st[1::2, :3 * ks] = _mfma_fragments(blk)
    slot = torch.tensor([[temporal_slot_feature(t, k) for k in range(32)] for t in range(2)], device=dev)   # [2][32]
    wo_ext = torch.cat([w_out, torch.zeros(H, C, 1, dtype=w_out.dtype, device=dev)], dim=2)                  # column 48 = zero
    wog = wo_ext[:, :, slot.reshape(-1)].reshape(H, nb, 16, 2, 32).permute(0, 3, 1, 2, 4).reshape(H, 2 * nb, 16, 32)
    st[1::2, lay["b_wo"]:lay["b_wo"] + 2 * nb] = _mfma_fragments(wog.to(T))
    tv = torch.zeros(H, 48, F, dtype=torch.float32, device=dev)
    tv[:, :d] = tb[:, :, 2 * d:3 * d].transpose(1, 2)
    st[1::2, lay["b_tab"]:lay["b_tab"] + 3] = tv.reshape(H, -1).contiguous().view(T).reshape(H, 3, 512)
    return st.reshape(-1).contiguous()


def ff_block_layout(C: int, hidden: int):
    """(half-stages, pieces per half-stage, FF1 k-steps in half A) of the fyc_ff_block weight stream (include/fyc.h);
    (92, 32, 14) at the one shape csrc/ff_block.hip is built for (C = 320, hidden = 1280).  k-steps are 16 wide, blocks 32 rows."""
    nb, ks, chunks = C // 32, C // 16, hidden // 32
    sa = (7 * ks + 9) // 10                                      # FF1 k-steps whose W1 pieces sit in half A (the rest, and W2', in half B)
    hp = max(2 * nb, 2 * sa + 1, 2 * (ks - sa) + 2 * nb)
    return ks // 2 + 2 * chunks + 2, (hp + 3) // 4 * 4, sa


def ff_slot_unit(sg: int, k: int) -> int:
    """hidden unit (within its chunk of 32) that k-slot k (0..15) of k-step sg (0, 1) of the FF2 operand carries in csrc/ff_block.hip:
    a lane (token, k half kh) holds of a 32 x 32 result the rows 8 b + 4 kh + e (register 4 b + e); registers 8 sg .. 8 sg + 7 are
    its operand of k-step sg"""
    kh, e = k // 8, k % 8
    return 16 * sg + 8 * (e // 4) + 4 * kh + e % 4


def _mfma_fragments32(w: Tensor) -> Tensor:
    """[..., 32, 16] weight blocks -> [..., 512]: the A operand of v_mfma_f32_32x32x16_bf16 in lane order - position
    8 l + e = block[l % 32][8 (l // 32) + e] (include/fyc.h, fyc_ff_block)"""
    lead = w.shape[:-2]
    return w.reshape(*lead, 32, 2, 8).transpose(-3, -2).reshape(*lead, 512)


def _mfma_fragments(w: Tensor) -> Tensor:
    """[..., 16, 32] weight blocks -> [..., 512]: the operand fragment of v_mfma_f32_16x16x32_bf16 in lane order - position
    8 l + e = block[l & 15][8 (l >> 4) + e] (include/fyc.h, fyc_ff_block)"""
    lead = w.shape[:-2]
    return w.reshape(*lead, 16, 4, 8).transpose(-3, -2).reshape(*lead, 512)


def pack_ff_block(ff: Packed) -> Tensor:
    """weight stream of fyc_ff_block (csrc/ff_block.hip) from a packed feed-forward (`_ff`) whose LayerNorm is folded into FF1:
    half-stages x pieces x 512 elements (layout: include/fyc.h; 92 x 32 x 1 KiB at C = 320), a piece = the A operand of one
    v_mfma_f32_32x32x16_bf16 (a 32 x 16 weight block).  Half t < ks / 2: k-steps 2t, 2t + 1 of the projection of the token half
    of the merged [Wp | Wp W2] weight.  Chunk c of 32 hidden units: half ks/2 + 2c ("A") = the value / gate blocks of W1 for
    k-steps 0..sa-1 and the f32 bias of chunk c - 1; half ks/2 + 2c + 1 ("B") = the remaining W1 pieces and the W2' columns of
    chunk c - 1 in the k-slot order the kernel's GEGLU outputs have (ff_slot_unit); the last two halves hold bias / W2' of the
    last chunk."""
    w1, b1, po = ff.w1, ff.b1, ff.po_w
    assert ff.cs1 is not None, "fyc_ff_block needs the LayerNorm folded into FF1"
    C = w1.shape[1]
    hid = w1.shape[0] // 2
    assert C % 32 == 0 and hid % 32 == 0 and tuple(po.shape) == (C, C + hid) and w1.dtype == po.dtype
    nb, ks, chunks = C // 32, C // 16, hid // 32
    nh, hp, sa = ff_block_layout(C, hid)
    npj = ks // 2
    dev = w1.device
    st = torch.zeros(nh, hp, 512, dtype=w1.dtype, device=dev)
    # projection: piece s' nb + j of half t = Wp[32 j .. +32][16 (2 t + s') .. +16]
    wp = _mfma_fragments32(po[:, :C].reshape(nb, 32, npj, 2, 16).permute(2, 3, 0, 1, 4))           # [t][s'][j][512]
    st[:npj, : 2 * nb] = wp.reshape(npj, 2 * nb, 512)
    # FF1: ff.w1 is GEGLU-packed in blocks of 16 (16 value rows, their 16 gate rows, ...); the kernel wants the 32 value rows and
    # the 32 gate rows of a chunk as two 32-row blocks, units in natural order
    vg = w1.reshape(chunks, 2, 2, 16, C).permute(0, 2, 1, 3, 4).reshape(chunks, 2, 32, C)             # [c][value | gate][unit][C]
    f1 = _mfma_fragments32(vg.reshape(chunks, 2, 32, ks, 16).permute(0, 3, 1, 2, 4))                 # [c][s][v][512]
    ha = torch.arange(chunks, device=dev) * 2 + npj
    st[ha, : 2 * sa] = f1[:, :sa].reshape(chunks, 2 * sa, 512)
    st[ha + 1, : 2 * (ks - sa)] = f1[:, sa:].reshape(chunks, 2 * (ks - sa), 512)
    # bias piece (index 2 sa) of the half A that FOLLOWS the chunk: f32 [32 value | 32 gate] (beta folded in) - the kernel gates a
    # chunk one chunk after its FF1, on LayerNorm-ed tokens (bit pattern of the floats inside a bf16 stream)
    cst = torch.zeros(chunks, 512 * w1.element_size() // 4, dtype=torch.float32, device=dev)
    cst[:, :64] = b1.reshape(chunks, 2, 2, 16).permute(0, 2, 1, 3).reshape(chunks, 64)
    st[ha + 2, 2 * sa] = cst.view(w1.dtype)
    # FF2: pieces 2 (ks - sa) + sg nb + j of the half B that follows the chunk = W2'[32 j .. +32][k-slots of k-step sg of chunk c]
    slot = torch.tensor([[ff_slot_unit(sg, k) for k in range(16)] for sg in range(2)], device=dev)     # [sg][16]
    w2 = po[:, C:].reshape(nb, 32, chunks, 32)[..., slot.reshape(-1)].reshape(nb, 32, chunks, 2, 16)   # [j][32][c][sg][slot]
    f2 = _mfma_fragments32(w2.permute(2, 3, 0, 1, 4))                                        # [c][sg][j][512]
    st[ha + 3, 2 * (ks - sa): 2 * (ks - sa) + 2 * nb] = f2.reshape(chunks, 2 * nb, 512)
    return st.reshape(-1).contiguous()


def pack_panel_linear(w: Tensor) -> Tensor:
    """weight stream of fyc_panel_linear (csrc/panel_linear.hip) from a linear weight (N, K): (N / PN) column passes x (K / 64)
    stages x 2 PN / 16 pieces x 512 elements, PN = 320 columns per pass (N itself below 320: the emulated small-width tests);
    piece s * (PN / 16) + j of stage t of pass P = MFMA fragment of W[PN P + 16 j .. +16][32 (2 t + s) .. +32] (include/fyc.h)"""
    N, K = w.shape
    pn = min(N, 320)
    assert N % pn == 0 and pn % 16 == 0 and K % 64 == 0
    fr = _mfma_fragments(w.reshape(N // pn, pn // 16, 16, K // 64, 2, 32).permute(0, 3, 4, 1, 2, 5))     # [P][t][s][j][512]
    return fr.reshape(-1).contiguous()


def sinusoidal_pe(channels: int, length: int) -> Tensor:
    """pe[p, 2i] = sin(p * exp(-2i ln(1e4)/C)), pe[p, 2i+1] = cos(...) (reference motion_module.py:295-301)."""
    import math
    pos = torch.arange(length, dtype=torch.float32)[:, None]
    div = torch.exp(torch.arange(0, channels, 2, dtype=torch.float32) * (-math.log(10000.0) / channels))
    pe = torch.zeros(length, channels)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.contiguous()


def pack_unet(sd: Dict[str, Tensor], cfg: UNet3DConfig, dtype, device) -> Packed:
    """sd: reference-schema state dict (fp32, CPU).  Returns the packed parameter tree."""
    cfg.validate()
    sd = {k: v.detach().float().cpu() for k, v in sd.items()}
    nb = len(cfg.block_out_channels)
    P = Packed(cfg=cfg, dtype=dtype)
    # use_first_frame_condition_concat: the reference halves conv_in's output (`sample = sample / 2`, unet.py:589-590) - folded into
    # the packed weight and bias (a power of two: exact in every storage type)
    half = 0.5 if cfg.use_first_frame_condition_concat else 1.0
    P["conv_in_w"] = pack_conv3x3(sd["conv_in.weight"] * half, dtype, device)
    P["conv_in_b"] = f32(sd["conv_in.bias"] * half, device)
    # time / fps / flow embedding MLPs stay f32 (M = batch rows only; precision matters, FLOPs do not)
    emb = {}
    for name in (["time_embedding"] + (["camera_motion_embedding"] if cfg.use_camera_motion_condition else [])
                 + (["fps_embedding", "motion_embedding"] if cfg.use_fps_condition else [])):
        emb[name] = Packed(w1=f32(sd[name + ".linear_1.weight"], device), b1=f32(sd[name + ".linear_1.bias"], device),
                           w2=f32(sd[name + ".linear_2.weight"], device), b2=f32(sd[name + ".linear_2.bias"], device))
    P["emb"] = emb
    resnets, temb_w, temb_b = [], [], []

    def add_resnet(p):
        r = _resnet(sd, p, dtype, device)
        r["temb_off"] = sum(w.shape[0] for w in temb_w)
        temb_w.append(sd[p + ".time_emb_proj.weight"])
        temb_b.append(sd[p + ".time_emb_proj.bias"])
        resnets.append(r)
        return r

    down = []
    for i, bt in enumerate(cfg.down_block_types):
        p = f"down_blocks.{i}"
        layers = []
        for j in range(cfg.layers_per_block):
            layers.append(Packed(
                resnet=add_resnet(f"{p}.resnets.{j}"),
                attn=_transformer(sd, f"{p}.attentions.{j}", cfg, dtype, device) if bt.startswith("CrossAttn") else None,
                motion=_motion(sd, f"{p}.motion_modules.{j}", cfg, dtype, device)
                if cfg.use_motion_module and (2 ** i) in cfg.motion_module_resolutions else None))
        ds = None
        if i != nb - 1:
            ds = Packed(w=pack_conv3x3(sd[f"{p}.downsamplers.0.conv.weight"], dtype, device), b=f32(sd[f"{p}.downsamplers.0.conv.bias"], device))
        down.append(Packed(layers=layers, down=ds))
    P["down"] = down
    P["mid"] = Packed(r0=add_resnet("mid_block.resnets.0"), attn=_transformer(sd, "mid_block.attentions.0", cfg, dtype, device),
                      motion=_motion(sd, "mid_block.motion_modules.0", cfg, dtype, device)
                      if cfg.use_motion_module and cfg.motion_module_mid_block else None,
                      r1=add_resnet("mid_block.resnets.1"))
    up = []
    for i, bt in enumerate(cfg.up_block_types):
        p = f"up_blocks.{i}"
        layers = []
        for j in range(cfg.layers_per_block + 1):
            layers.append(Packed(
                resnet=add_resnet(f"{p}.resnets.{j}"),
                attn=_transformer(sd, f"{p}.attentions.{j}", cfg, dtype, device) if bt.startswith("CrossAttn") else None,
                motion=_motion(sd, f"{p}.motion_modules.{j}", cfg, dtype, device)
                if cfg.use_motion_module and (2 ** (nb - 1 - i)) in cfg.motion_module_resolutions else None))
        us = None
        if i != nb - 1:
            us = Packed(w=pack_conv3x3(sd[f"{p}.upsamplers.0.conv.weight"], dtype, device), b=f32(sd[f"{p}.upsamplers.0.conv.bias"], device))
        up.append(Packed(layers=layers, up=us))
    P["up"] = up
    P["out_g"], P["out_b"] = f32(sd["conv_norm_out.weight"], device), f32(sd["conv_norm_out.bias"], device)
    P["conv_out_w"] = pack_conv3x3(sd["conv_out.weight"], dtype, device)
    P["conv_out_b"] = f32(sd["conv_out.bias"], device)
    # all ResnetBlock3D.time_emb_proj stacked: one GEMM gives every block's time-embedding row
    P["temb_w"] = f32(torch.cat(temb_w, 0), device)
    P["temb_b"] = f32(torch.cat(temb_b, 0), device)
    P["temb_total"] = P["temb_w"].shape[0]
    return P


def pack_vae_decoder(sd: Dict[str, Tensor], cfg: VAEDecoderConfig, dtype, device) -> Packed:
    sd = {k: v.detach().float().cpu() for k, v in sd.items()}
    P = Packed(cfg=cfg, dtype=dtype)
    # post_quant_conv (1x1, 4->4) is folded into decoder.conv_in (3x3, 4->C): both linear, no bias
    # interaction except through the zero padding, so it is kept as its own tiny GEMM instead.
    P["pq_w"] = pack_linear(sd["post_quant_conv.weight"], dtype, device, k_pad=pad_channels(cfg.latent_channels))
    P["pq_b"] = f32(sd["post_quant_conv.bias"], device)
    P["conv_in_w"] = pack_conv3x3(sd["decoder.conv_in.weight"], dtype, device)
    P["conv_in_b"] = f32(sd["decoder.conv_in.bias"], device)
    P["mid_r0"] = _resnet(sd, "decoder.mid_block.resnets.0", dtype, device)
    a = "decoder.mid_block.attentions.0"
    P["attn"] = Packed(g=f32(sd[a + ".group_norm.weight"], device), b=f32(sd[a + ".group_norm.bias"], device),
                       qkv_w=pack_linear(torch.cat([sd[a + ".query.weight"], sd[a + ".key.weight"], sd[a + ".value.weight"]], 0), dtype, device),
                       qkv_b=f32(torch.cat([sd[a + ".query.bias"], sd[a + ".key.bias"], sd[a + ".value.bias"]], 0), device),
                       o_w=pack_linear(sd[a + ".proj_attn.weight"], dtype, device), o_b=f32(sd[a + ".proj_attn.bias"], device))
    P["mid_r1"] = _resnet(sd, "decoder.mid_block.resnets.1", dtype, device)
    ups = []
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        rs = [_resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", dtype, device) for j in range(cfg.layers_per_block + 1)]
        us = None
        if i != nb - 1:
            k = f"decoder.up_blocks.{i}.upsamplers.0.conv"
            us = Packed(w=pack_conv3x3(sd[k + ".weight"], dtype, device), b=f32(sd[k + ".bias"], device))
        ups.append(Packed(resnets=rs, up=us))
    P["ups"] = ups
    P["out_g"], P["out_b"] = f32(sd["decoder.conv_norm_out.weight"], device), f32(sd["decoder.conv_norm_out.bias"], device)
    P["conv_out_w"] = pack_conv3x3(sd["decoder.conv_out.weight"], dtype, device)
    P["conv_out_b"] = f32(sd["decoder.conv_out.bias"], device)
    return P


def _vae_attn(sd, a: str, dtype, device) -> Packed:
    return Packed(g=f32(sd[a + ".group_norm.weight"], device), b=f32(sd[a + ".group_norm.bias"], device),
                  qkv_w=pack_linear(torch.cat([sd[a + ".query.weight"], sd[a + ".key.weight"], sd[a + ".value.weight"]], 0), dtype, device),
                  qkv_b=f32(torch.cat([sd[a + ".query.bias"], sd[a + ".key.bias"], sd[a + ".value.bias"]], 0), device),
                  o_w=pack_linear(sd[a + ".proj_attn.weight"], dtype, device), o_b=f32(sd[a + ".proj_attn.bias"], device))


def pack_vae_encoder(sd: Dict[str, Tensor], cfg: VAEDecoderConfig, dtype, device) -> Packed:
    """Encoder half of AutoencoderKL (reference diffusers/models/vae.py:67-144) + quant_conv."""
    sd = {k: v.detach().float().cpu() for k, v in sd.items()}
    P = Packed(cfg=cfg, dtype=dtype)
    P["conv_in_w"] = pack_conv3x3(sd["encoder.conv_in.weight"], dtype, device)
    P["conv_in_b"] = f32(sd["encoder.conv_in.bias"], device)
    downs = []
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        rs = [_resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}", dtype, device) for j in range(cfg.layers_per_block)]
        ds = None
        if i != nb - 1:
            k = f"encoder.down_blocks.{i}.downsamplers.0.conv"
            ds = Packed(w=pack_conv3x3(sd[k + ".weight"], dtype, device), b=f32(sd[k + ".bias"], device))
        downs.append(Packed(resnets=rs, down=ds))
    P["downs"] = downs
    P["mid_r0"] = _resnet(sd, "encoder.mid_block.resnets.0", dtype, device)
    P["attn"] = _vae_attn(sd, "encoder.mid_block.attentions.0", dtype, device)
    P["mid_r1"] = _resnet(sd, "encoder.mid_block.resnets.1", dtype, device)
    P["out_g"], P["out_b"] = f32(sd["encoder.conv_norm_out.weight"], device), f32(sd["encoder.conv_norm_out.bias"], device)
    P["conv_out_w"] = pack_conv3x3(sd["encoder.conv_out.weight"], dtype, device)
    P["conv_out_b"] = f32(sd["encoder.conv_out.bias"], device)
    C2 = 2 * cfg.latent_channels
    P["q_w"] = pack_linear(sd["quant_conv.weight"], dtype, device, k_pad=pad_channels(C2))
    P["q_b"] = f32(sd["quant_conv.bias"], device)
    return P
