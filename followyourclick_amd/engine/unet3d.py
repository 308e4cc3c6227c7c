"""UNet3D denoiser: the op schedule of one forward pass over channels-last activations.

Restates what UNet3DConditionModel.forward computes (reference animatediff/models/unet.py:422-672,
unet_blocks.py:342-360/482-529/604-632/749-809/880-905, resnet.py:296-342, attention.py:217-308 &
489-564, motion_module.py:157-208 & 270-283 & 371-464) as a flat sequence of libfyc_hip.so calls:

  * activations live as [B*F*H*W][C] (channels-last == token-major), so every `rearrange`/`permute`
    of the reference is free; only the 9-channel model input and the 4-channel prediction cross the
    (b,c,f,h,w) boundary (ops.unet_input / ops.cfg_ddim_step);
  * text (and IP) keys/values do not depend on the timestep or the frame: they are projected once
    per clip (`prepare_context`) instead of F*steps times (reference attention.py:264 repeats them);
  * all time-embedding work (3 sinusoid MLPs + the 22 ResnetBlock3D.time_emb_proj) is done for every
    DDIM step up front in f32 (`prepare_time_embeddings`) and enters the convs as an epilogue row.

The only torch calls are allocation (`torch.empty/zeros`) and host->device copies of tables.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional, Sequence, Tuple, Union

import torch

from .. import _lib as L
from .. import ops as ops_mod
from .base import EngineBase
from .config import UNet3DConfig
from .weights import Packed, pack_ff_block, pack_panel_linear, pack_temporal_block

Tensor = torch.Tensor

# Statistics of an activation are accumulated by the epilogue of the GEMM / conv that writes it (fyc_gemm chan_stats / row_parts)
# instead of by a read pass per GroupNorm / LayerNorm.  FYC_FUSE_STATS=0 restores the separate passes (A/B measurements).
FUSE_STATS = os.environ.get("FYC_FUSE_STATS", "1") != "0"
FUSE_TEMPORAL = os.environ.get("FYC_FUSE_TEMPORAL", "1") != "0"   # fyc_temporal_block: one kernel per temporal attention sub-block (C = 320 level)
TEMPORAL_RR = os.environ.get("FYC_TEMPORAL_RR", "1") != "0"       # ... in its register-resident form (csrc/temporal_block_rr.hip, pre-packed weight stream)
FUSE_FF = os.environ.get("FYC_FUSE_FF", "1") != "0"               # fyc_ff_block: LayerNorm + FF1 + GEGLU + FF2 + output projection in one kernel (C = 320 level)
FUSE_PANEL = os.environ.get("FYC_FUSE_PANEL", "1") != "0"         # fyc_panel_linear for GroupNorm -> proj_in: the norm is applied to the operand registers (no apply pass)
# the plain / residual K <= 640 projections through fyc_panel_linear as well: measured equal to fyc_gemm (98 vs 91 us at the 64x64
# level, 60 vs 61 at 32x32: one workgroup per CU leaves its memory phases exposed, profiles/r03_panel_linear_probe.txt) - off
PANEL_ALL = os.environ.get("FYC_PANEL_ALL", "0") != "0"
FUSE_ROWS = os.environ.get("FYC_FUSE_ROWS", "0") != "0"      # the LayerNorm half (row_parts): measured slower than the separate fyc_row_stats pass (profiles/r02_stats_fusion_ab.txt), off by default


class Act:
    """an activation [rows][C] plus what its producer already knows about it: `cs` = per-(frame, channel) {sum, sum of
    squares} ([rows / cs_rows][C][2] f64), `rp` = per-row partial {sum, sum sq} ([rows][rp_n][2] f32)"""
    __slots__ = ("t", "C", "cs", "cs_rows", "rp", "rp_n")

    def __init__(self, t: Tensor, C: int, cs: Optional[Tensor] = None, cs_rows: int = 0, rp: Optional[Tensor] = None, rp_n: int = 0):
        self.t, self.C, self.cs, self.cs_rows, self.rp, self.rp_n = t, C, cs, cs_rows, rp, rp_n


def sinusoid_host(values: Sequence[float], dim: int) -> Tensor:
    """Timesteps(dim, flip_sin_to_cos=True, freq_shift=0) on the host: [cos | sin]
    (reference diffusers/models/embeddings.py:21-64; unet.py:129).  f32 like the reference."""
    half = dim // 2
    v = torch.tensor(list(values), dtype=torch.float32)
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    ang = v[:, None] * freqs[None, :]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1).contiguous()


class UNet3DEngine(EngineBase):
    def __init__(self, packed: Packed, ops=None):
        self.P = packed
        self.cfg: UNet3DConfig = packed.cfg
        self.dtype = packed.dtype
        self.ops = ops if ops is not None else ops_mod.get()
        self.device = packed.conv_in_w.device
        self.heads = self.cfg.attention_head_dim
        self.groups = self.cfg.norm_num_groups
        self.mat_attn = self.dtype == torch.float32  # f32 parity mode: materialised attention through the GEMM (bf16 / f16: fyc_attention)
        self.transformers: List[Packed] = []
        for blk in self.P.down:
            self.transformers += [l.attn for l in blk.layers if l.attn is not None]
        self.transformers.append(self.P.mid.attn)
        for blk in self.P.up:
            self.transformers += [l.attn for l in blk.layers if l.attn is not None]
        for i, t in enumerate(self.transformers):
            t["idx"] = i
        self.ctx_cache = None
        self.fuse_stats = FUSE_STATS
        self.fuse_rows = FUSE_STATS and FUSE_ROWS
        self.ops.ensure_init(self.device)

    # ---- once per clip ---------------------------------------------------------------------
    def prepare_time_embeddings(self, timesteps: Sequence[int], fps: Optional[Sequence[float]], flow: Optional[Sequence[float]],
                                batch: int, camera: Optional[Sequence[float]] = None):
        """Returns (emb [S*batch, 1280] f32, temb [S*batch, temb_total] f32); row = step*batch + b.
        emb = time_embedding(sin t) + fps_embedding(sin fps_b) + motion_embedding(sin flow_b)
        (reference unet.py:526-558); temb = time_emb_proj(SiLU(emb)) of all ResNets (resnet.py:306-307)."""
        cfg, P, o = self.cfg, self.P, self.ops
        c0, S = cfg.block_out_channels[0], len(timesteps)
        rows = S * batch

        def mlp(m: Packed, sin: Tensor, residual=None) -> Tensor:
            h = self.lin(sin.to(self.device), m.w1, rows, bias=m.b1)
            a = self.new(rows, h.shape[1], dtype=torch.float32)
            o.silu_f32(h, a)
            return self.lin(a, m.w2, rows, bias=m.b2, residual=residual)

        def per_row(v):        # one value per batch element, repeated over the steps - or already one per row (per-sample timesteps: S = rows, batch = 1)
            s = sinusoid_host(v, c0)
            return s if s.shape[0] == rows else s.repeat(S, 1)

        emb = mlp(P.emb["time_embedding"], sinusoid_host(timesteps, c0).repeat_interleave(batch, dim=0))
        if cfg.use_camera_motion_condition and camera is not None:       # reference unet.py:538-544: added before the fps / motion embeddings
            emb = mlp(P.emb["camera_motion_embedding"], per_row(camera), residual=emb)
        if cfg.use_fps_condition and fps is not None:
            emb = mlp(P.emb["fps_embedding"], per_row(fps), residual=emb)
            emb = mlp(P.emb["motion_embedding"], per_row(flow), residual=emb)
        act = self.new(rows, emb.shape[1], dtype=torch.float32)
        o.silu_f32(emb, act)
        temb = self.lin(act, P.temb_w, rows, bias=P.temb_b)
        return emb, temb

    def prepare_context(self, ctx: Tensor, ip_tokens: Optional[Tensor] = None) -> None:
        """ctx: (B_eff, 77, D) text states; ip_tokens: (B_eff, n, D) projected image tokens.  Projects the
        cross-attention K / V^T of all 16 transformers once (they are constant over steps and frames)."""
        cfg, o = self.cfg, self.ops
        Bn, Nk, D = ctx.shape
        H = self.heads

        def to_T(t: Tensor) -> Tensor:
            t = t.to(device=self.device, dtype=torch.float32).contiguous()
            y = self.new(t.shape[0] * t.shape[1], D)
            o.cast_from_f32(t, y, rows=t.shape[0] * t.shape[1], cols=D, ld=D)
            return y

        def project(x: Tensor, w: Tensor, n_tok: int, C: int):
            d = C // H
            ld = ((n_tok + 7) // 8) * 8
            k = self.new(Bn, H, n_tok, d)
            vt = self.zeros(Bn, H, d, ld)
            o.gemm(x, w, None, M=Bn * n_tok, N=2 * C, K=D, lda=D, ldw=D, epilogue=L.EPI_HEADS,
                   heads=dict(seg_cols=C, heads=H, tokens=n_tok, outs=[k, vt], transposed=[0, 1], ld=[0, ld]))
            return k, vt, ld

        if cfg.use_ip_cross_attention and ip_tokens is None:
            # IPCrossAttention.forward always treats the LAST num_tokens tokens of encoder_hidden_states as image tokens
            # (reference animatediff/models/attention.py:52-53, 104-120) - also when the caller passed plain text states
            n = cfg.ip_num_tokens
            if Nk <= n:
                raise ValueError(f"encoder_hidden_states has {Nk} tokens, the IP cross-attention splits off the last {n}")
            ctx, ip_tokens = ctx[:, :Nk - n], ctx[:, Nk - n:]
            Nk -= n
        xt = to_T(ctx)
        xi = to_T(ip_tokens) if (cfg.use_ip_cross_attention and ip_tokens is not None) else None
        cache = []
        for t in self.transformers:
            e = dict(text=project(xt, t.kv2_w, Nk, t.C), n_text=Nk, ip=None)
            if xi is not None:
                e["ip"] = project(xi, t.kvip_w, ip_tokens.shape[1], t.C)
                e["n_ip"] = ip_tokens.shape[1]
            cache.append(e)
        self.ctx_cache = cache

    # ---- attention cores -----------------------------------------------------------------------
    def _attend(self, q: Tensor, k: Tensor, vt: Tensor, out: Tensor, *, batch: int, n_q: int, n_k: int, d: int, ldvt: int,
                C: int, kv_div: int, accumulate: bool = False, o_scale: float = 1.0) -> None:
        H, o = self.heads, self.ops
        scale = d ** -0.5
        if not self.mat_attn:
            o.attention(q, k, vt, out, batch=batch, heads=H, n_q=n_q, n_k=n_k, d=d, ldo=C, ldvt=ldvt, scale=scale,
                        kv_batch_div=kv_div, accumulate=accumulate, o_scale=o_scale)
            return
        # f32 parity mode: softmax(q k^T * scale) v materialised per batch element through the GEMM
        # (reference CrossAttention._attention, diffusers/models/attention.py:649-678)
        ldS = ((n_k + 7) // 8) * 8
        for b in range(batch):
            kb = b // kv_div
            S = self.zeros(H, n_q, ldS)
            o.gemm(q[b], k[kb], S, M=n_q, N=n_k, K=d, lda=d, ldw=d, ldo=ldS, batch=H, stride_a=n_q * d, stride_w=n_k * d,
                   stride_o=n_q * ldS, out_scale=scale)
            o.softmax_rows(S, rows=H * n_q, cols=n_k, ld=ldS)
            dst = out[b * n_q:(b + 1) * n_q]
            if accumulate:
                tmp = self.new(n_q, C)
                o.gemm(S, vt[kb], tmp, M=n_q, N=d, K=ldS, lda=ldS, ldw=ldvt, ldo=C, batch=H, stride_a=n_q * ldS, stride_w=d * ldvt,
                       stride_o=d)
                self._axpy(tmp, dst, n_q, C, o_scale)
            else:
                o.gemm(S, vt[kb], dst, M=n_q, N=d, K=ldS, lda=ldS, ldw=ldvt, ldo=C, batch=H, stride_a=n_q * ldS, stride_w=d * ldvt,
                       stride_o=d)

    def row_stats(self, x: Tensor, rows: int, C: int, eps: float = 1e-5) -> Tensor:
        """{mean, rstd} per token row: the statistics half of a LayerNorm whose affine half lives in the next GEMM"""
        st = self.new(rows, 2, dtype=torch.float32)
        self.ops.row_stats(x, st, rows=rows, C_=C, eps=eps)
        return st

    # ---- fused statistics: _cs_plan / _cs_finish live in EngineBase (the VAE engines use them too) ----------------------------------------
    def _gn(self, x: Union[Act, Tuple[Act, Act]], gamma: Tensor, beta: Tensor, rows: int, rows_per_sample: int, eps: float,
            silu: bool) -> Tuple[Tensor, Optional[Tensor]]:
        """GroupNorm (+SiLU) of an activation or of the channel concat of two (up blocks: cat([hidden, skip], dim=1),
        reference unet_blocks.py:763,885).  Returns (normalised tensor, materialised concat or None)."""
        o = self.ops
        if isinstance(x, Act):
            if x.cs is not None and rows_per_sample % x.cs_rows == 0:
                y = self.new(rows, x.C)
                o.gn_apply_cs(x.t, x.cs, gamma, beta, y, rows=rows, C1=x.C, groups=self.groups, rows_per_sample=rows_per_sample,
                              eps=eps, silu=silu, cs_rows=x.cs_rows)
                return y, None
            return self.group_norm(x.t, gamma, beta, rows, x.C, rows_per_sample, eps, silu), None
        a, b = x
        if a.cs is not None and b.cs is not None and a.cs_rows == b.cs_rows and rows_per_sample % a.cs_rows == 0:
            y = self.new(rows, a.C + b.C)
            o.gn_apply_cs(a.t, a.cs, gamma, beta, y, rows=rows, C1=a.C, groups=self.groups, rows_per_sample=rows_per_sample,
                          eps=eps, silu=silu, x2=b.t, cs2=b.cs, C2=b.C, cs_rows=a.cs_rows)
            return y, None
        cat = self.new(rows, a.C + b.C)
        o.concat_channels(a.t, b.t, cat, rows=rows, c1=a.C, c2=b.C)
        return self.group_norm(cat, gamma, beta, rows, a.C + b.C, rows_per_sample, eps, silu), cat

    def _lin_rp(self, x: Tensor, w: Tensor, rows: int, bias=None, residual=None) -> Act:
        """Linear whose output feeds a (folded) LayerNorm: the epilogue also writes the per-row partial sums"""
        N, K = w.shape
        out = self.new(rows, N, dtype=x.dtype)
        if FUSE_PANEL and PANEL_ALL and not self.fuse_rows and self.ops.panel_linear_supported(x.dtype, rows=rows, N=N, K=K):
            self.ops.panel_linear(x, out, wstream=self._panel_stream(w), rows=rows, N=N, K=K, bias=bias, residual=residual)
            return Act(out, N)
        rp, n = None, 0
        if self.fuse_rows:
            n = self.ops.gemm_row_parts(x.dtype, M=rows, N=N, K=K)
            rp = self.new(rows, n, 2, dtype=torch.float32)
        self.ops.gemm(x, w, out, M=rows, N=N, K=K, lda=K, ldw=K, ldo=N, bias=bias, residual=residual, ldr=N, row_parts=rp, row_nparts=n)
        return Act(out, N, rp=rp, rp_n=n)

    def _panel_stream(self, w: Tensor) -> Tensor:
        """fragment-ordered copy of a linear weight for fyc_panel_linear, packed once per weight tensor"""
        cache = self.__dict__.setdefault("_panel_ws", {})
        key = w.data_ptr()
        hit = cache.get(key)
        if hit is None or hit[0] is not w:                      # the entry holds the weight tensor: an address alone can be recycled
            hit = cache[key] = (w, pack_panel_linear(w))
        return hit[1]

    def _norm_proj_in(self, x: Act, node: Packed, rows: int, rows_per_sample: int) -> Act:
        """GroupNorm -> proj_in of a transformer / motion module (reference attention.py:269-270, motion_module.py:188-191).  With the
        producer's channel sums at hand the norm is applied to fyc_panel_linear's operand registers: no normalised tensor in HBM."""
        C = node.C
        if (FUSE_PANEL and isinstance(x, Act) and x.cs is not None and not self.fuse_rows and rows_per_sample % x.cs_rows == 0
                and self.ops.panel_linear_supported(self.dtype, rows=rows, N=C, K=C, gn_rows_per_sample=rows_per_sample, gn_groups=self.groups)):
            out = self.new(rows, C)
            self.ops.panel_linear(x.t, out, wstream=self._panel_stream(node.pin_w), rows=rows, N=C, K=C, bias=node.pin_b, gn_cs=x.cs,
                                  gn_gamma=node.norm_g, gn_beta=node.norm_b, gn_rows_per_sample=rows_per_sample,
                                  gn_stat_samples=rows_per_sample // x.cs_rows, gn_groups=self.groups, gn_eps=1e-6)
            return Act(out, C)
        h, _ = self._gn(x, node.norm_g, node.norm_b, rows, rows_per_sample, 1e-6, False)
        return self._lin_rp(h, node.pin_w, rows, bias=node.pin_b)

    def _ln_args(self, tok: Act, rows: int, C: int) -> dict:
        """ln_stats arguments of the GEMM that consumes LayerNorm(tok): the producer's partial sums, or a statistics pass"""
        if tok.rp is not None:
            return dict(ln_stats=tok.rp, ln_nparts=tok.rp_n, ln_eps=1e-5)
        return dict(ln_stats=self.row_stats(tok.t, rows, C))

    def _pe_rowbias(self, a: Packed, B: int, F: int) -> Tensor:
        """pe_f W_qkv^T for token rows ordered [(b f)][pixel]: one row per (b, f)"""
        key = (B, F)
        cache = a.setdefault("_pe_rb", {})
        if key not in cache:
            if F > a.pe_w.shape[0]:
                raise ValueError(f"video_length {F} exceeds the positional table ({a.pe_w.shape[0]})")
            cache[key] = a.pe_w[:F].repeat(B, 1).contiguous()
        return cache[key]

    def _axpy(self, x: Tensor, y: Tensor, rows: int, C: int, alpha: float) -> None:
        """y = y + alpha * x via the GEMM epilogue (parity mode only): (x @ (alpha*I)) + y."""
        key = ("axpy", C, alpha)
        if not hasattr(self, "_eyes"):
            self._eyes = {}
        if key not in self._eyes:
            self._eyes[key] = (torch.eye(C, dtype=torch.float32) * alpha).to(self.dtype).to(self.device)
        self.ops.gemm(x, self._eyes[key], y, M=rows, N=C, K=C, lda=C, ldw=C, ldo=C, residual=y, ldr=C)

    # ---- blocks --------------------------------------------------------------------------------
    # `nxt` = (rows per statistics sample of the producing epilogue = one frame, rows per sample of the GroupNorm that consumes
    # the block's output: one frame for the per-frame norm of a transformer / motion module, F frames for the cross-frame norm
    # of a ResNet / conv_norm_out); fyc_chan_stats_reduce folds the row-tile partials to the consumer's granularity.
    def _conv_act(self, x: Tensor, w: Tensor, b: Tensor, frames: int, Hin: int, Win: int, nxt: Tuple[int, int], **kw) -> Act:
        Cout, K = w.shape
        if kw.get("up2"):
            Ho, Wo = kw["up_size"]
        else:
            st = kw.get("stride", 1)
            Ho, Wo = (Hin - 1) // st + 1, (Win - 1) // st + 1
        rows = frames * Ho * Wo
        plan = self._cs_plan(rows, nxt[0], Cout, K, L.GEMM_CONV3X3_UP2 if kw.get("up2") else L.GEMM_CONV3X3) if nxt else None
        out = self.conv(x, w, b, frames, Hin, Win, chan_parts=None if plan is None else plan[0], cs_rows=nxt[0] if plan is not None else 0, **kw)
        return Act(out, Cout, None if plan is None else self._cs_finish(plan, rows, nxt[0], Cout, nxt[1]), nxt[1] if nxt else 0)

    def resnet(self, r: Packed, x: Union[Act, Tuple[Act, Act]], temb: Tensor, g: dict, nxt: Tuple[int, int]) -> Act:
        """ResnetBlock3D (reference resnet.py:296-342): cross-frame GroupNorm statistics.  x may be the (hidden, skip) pair of an
        up block: the concat only ever exists normalised (norm1's output); the 1x1 shortcut reads both sources (dual-K GEMM)."""
        rows, rpb = g["rows"], g["F"] * g["H"] * g["W"]
        frames = g["B"] * g["F"]
        h, cat = self._gn(x, r.n1_g, r.n1_b, rows, rpb, self.cfg.norm_eps, True)
        tb = temb[:, r.temb_off:]  # view: row pitch stays temb_total (ldrb)
        # time-embedding row per clip, or per (clip, frame) when frame 0 carries the timestep-0 embedding (use_first_frame_condition)
        h1 = self._conv_act(h, r.c1_w, r.c1_b, frames, g["H"], g["W"], (g["H"] * g["W"], rpb), rowbias=tb,
                            rpb=g["H"] * g["W"] if g.get("temb_per_frame") else rpb, ldrb=temb.shape[1])
        h2, _ = self._gn(h1, r.n2_g, r.n2_b, rows, rpb, self.cfg.norm_eps, True)
        if r.sc_w is None:
            sc = x.t
        elif isinstance(x, Act) or cat is not None:
            sc = self.lin(cat if cat is not None else x.t, r.sc_w, rows, bias=r.sc_b)
        else:                       # conv_shortcut over cat([hidden, skip]) without materialising the concat
            a, b = x
            sc = self.new(rows, r.cout)
            self.ops.gemm(a.t, r.sc_w, sc, M=rows, N=r.cout, K=a.C + b.C, lda=a.C, ldw=a.C + b.C, ldo=r.cout, bias=r.sc_b,
                          a2=b.t, k_split=a.C, lda2=b.C)
        return self._conv_act(h2, r.c2_w, r.c2_b, frames, g["H"], g["W"], nxt, residual=sc)

    def feed_forward_out(self, ff: Packed, ln, tok: Act, residual: Optional[Tensor], rows: int, C: int, nxt=None) -> Act:
        """LN -> GEGLU FF -> (+tok) -> output projection (+residual) with FF2 and the projection merged into one GEMM
        over [tok | h] (see weights._ff): returns residual + Wp (tok + W2 h + b2) + bp."""
        hidden = ff.w1.shape[0] // 2
        cs_rows = nxt[0] if (nxt and self.fuse_stats) else 0
        fused = FUSE_FF and ff.cs1 is not None
        if fused and cs_rows and not self.ops.ff_block_supported(self.dtype, rows=rows, C_=C, hidden=hidden, cs_rows=cs_rows):
            cs_rows = 0             # frames that are not whole 128-row tiles: the consuming GroupNorm runs its own statistics pass
        if fused and self.ops.ff_block_supported(self.dtype, rows=rows, C_=C, hidden=hidden, cs_rows=cs_rows):
            # one kernel: the 4C-wide hidden activation never reaches HBM, the LayerNorm statistics come from the token registers
            if "_wstream" not in ff:
                ff["_wstream"] = pack_ff_block(ff)
            out = self.new(rows, C)
            parts = self.new(rows // 128 * C * 2, dtype=torch.float32) if cs_rows else None
            self.ops.ff_block(tok.t, residual, out, wstream=ff["_wstream"], b_out=ff.po_b, rows=rows, C_=C, hidden=hidden,
                              chan_parts=parts, cs_rows=cs_rows)
            cs = self._cs_finish((parts, 128, 1), rows, nxt[0], C, nxt[1]) if cs_rows else None
            return Act(out, C, cs, nxt[1] if cs_rows else 0)
        if ff.cs1 is not None:      # LayerNorm folded into FF1: statistics from the producer of tok (or one statistics pass)
            N1 = ff.w1.shape[0]
            hmid = self.new(rows, N1 // 2)
            self.ops.gemm(tok.t, ff.w1, hmid, M=rows, N=N1, K=C, lda=C, ldw=C, ldo=N1 // 2, bias=ff.b1, epilogue=L.EPI_GEGLU,
                          ln_colsum=ff.cs1, **self._ln_args(tok, rows, C))
        else:
            n = self.layer_norm(tok.t, ln, rows, C)
            hmid = self.lin(n, ff.w1, rows, bias=ff.b1, geglu=True)
        out = self.new(rows, C)
        K = ff.po_w.shape[1]
        plan = self._cs_plan(rows, nxt[0], C, K, L.GEMM_PLAIN) if nxt else None
        rp, rp_n = None, 0
        if residual is None and self.fuse_rows:      # inner motion blocks: the output is the next block's token stream (LayerNorm input)
            rp_n = self.ops.gemm_row_parts(tok.t.dtype, M=rows, N=C, K=K)
            rp = self.new(rows, rp_n, 2, dtype=torch.float32)
        self.ops.gemm(tok.t, ff.po_w, out, M=rows, N=C, K=K, lda=C, ldw=K, ldo=C, bias=ff.po_b, residual=residual, ldr=C,
                      a2=hmid, k_split=C, lda2=K - C, chan_parts=None if plan is None else plan[0], cs_rows=nxt[0] if plan is not None else 0,
                      row_parts=rp, row_nparts=rp_n)
        return Act(out, C, None if plan is None else self._cs_finish(plan, rows, nxt[0], C, nxt[1]), nxt[1] if nxt else 0, rp, rp_n)

    def transformer(self, t: Packed, x: Act, g: dict, nxt: Tuple[int, int]) -> Act:
        """Transformer3DModel + BasicTransformerBlock (reference attention.py:217-308, 489-564)."""
        rows, C, H, o = g["rows"], t.C, self.heads, self.ops
        BF, N = g["B"] * g["F"], g["H"] * g["W"]
        d = C // H
        tok = self._norm_proj_in(x, t, rows, N)
        # --- attn1: spatial self-attention
        ld = ((N + 7) // 8) * 8
        q, k, vt = self.new(BF, H, N, d), self.new(BF, H, N, d), (self.zeros(BF, H, d, ld) if ld != N else self.new(BF, H, d, ld))
        hd = dict(seg_cols=C, heads=H, tokens=N, outs=[q, k, vt], transposed=[0, 0, 1], ld=[0, 0, ld])
        if t.qkv_f is not None:     # LayerNorm (norm1) folded into the fused QKV projection
            w, b, cs = t.qkv_f
            o.gemm(tok.t, w, None, M=rows, N=3 * C, K=C, lda=C, ldw=C, bias=b, epilogue=L.EPI_HEADS, heads=hd, ln_colsum=cs,
                   **self._ln_args(tok, rows, C))
        else:
            n1 = self.layer_norm(tok.t, t.ln1, rows, C)
            o.gemm(n1, t.qkv_w, None, M=rows, N=3 * C, K=C, lda=C, ldw=C, epilogue=L.EPI_HEADS, heads=hd)
        att = self.new(rows, C)
        self._attend(q, k, vt, att, batch=BF, n_q=N, n_k=N, d=d, ldvt=ld, C=C, kv_div=1)
        tok = self._lin_rp(att, t.o1_w, rows, bias=t.o1_b, residual=tok.t)
        # --- attn2: cross-attention on the cached text (and IP) K/V
        q2 = self.new(BF, H, N, d)
        hd2 = dict(seg_cols=C, heads=H, tokens=N, outs=[q2], transposed=[0], ld=[0])
        if t.q2_f is not None:
            w, b, cs = t.q2_f
            o.gemm(tok.t, w, None, M=rows, N=C, K=C, lda=C, ldw=C, bias=b, epilogue=L.EPI_HEADS, heads=hd2, ln_colsum=cs,
                   **self._ln_args(tok, rows, C))
        else:
            n2 = self.layer_norm(tok.t, t.ln2, rows, C)
            o.gemm(n2, t.q2_w, None, M=rows, N=C, K=C, lda=C, ldw=C, epilogue=L.EPI_HEADS, heads=hd2)
        cache = self.ctx_cache[t.idx]
        kt, vtt, ldt = cache["text"]
        att2 = self.new(rows, C)
        self._attend(q2, kt, vtt, att2, batch=BF, n_q=N, n_k=cache["n_text"], d=d, ldvt=ldt, C=C, kv_div=g["F"])
        if cache["ip"] is not None:
            ki, vti, ldi = cache["ip"]
            self._attend(q2, ki, vti, att2, batch=BF, n_q=N, n_k=cache["n_ip"], d=d, ldvt=ldi, C=C, kv_div=g["F"],
                         accumulate=True, o_scale=self.cfg.ip_scale)
        tok = self._lin_rp(att2, t.o2_w, rows, bias=t.o2_b, residual=tok.t)
        return self.feed_forward_out(t.ff, t.ln3, tok, x.t, rows, C, nxt)

    def motion(self, m: Packed, x: Act, g: dict, nxt: Tuple[int, int]) -> Act:
        """VanillaTemporalModule (reference motion_module.py:157-208, 270-283, 371-464)."""
        rows, C, o = g["rows"], m.C, self.ops
        N, Hm = g["H"] * g["W"], self.cfg.motion_num_attention_heads
        d = C // Hm
        tok = self._norm_proj_in(x, m, rows, N)
        for bi, blk in enumerate(m.blocks):
            for a in blk.attns:
                if FUSE_TEMPORAL and a.qkv_f is not None and o.temporal_block_supported(self.dtype, clips=g["B"], frames=g["F"], pixels=N, heads=Hm, d=d):
                    cache = a.setdefault("_tblock", {})      # per-head operands, packed once per clip length
                    if g["F"] not in cache:
                        cache[g["F"]] = pack_temporal_block(a, Hm, g["F"])
                    out = self.new(rows, C)
                    kw = cache[g["F"]]
                    if not TEMPORAL_RR and "wstream" in kw:
                        kw = {k: v for k, v in kw.items() if k != "wstream"}
                    o.temporal_block(tok.t, out, clips=g["B"], frames=g["F"], pixels=N, heads=Hm, d=d, scale=d ** -0.5, **kw)
                    tok = Act(out, C)
                    continue
                if a.qkv_f is not None:     # LayerNorm folded into the QKV projection, positional table as a per-frame row bias
                    w, b, cs = a.qkv_f
                    qkv = self.new(rows, 3 * C)
                    rb = self._pe_rowbias(a, g["B"], g["F"]) if a.pe_w is not None else None
                    o.gemm(tok.t, w, qkv, M=rows, N=3 * C, K=C, lda=C, ldw=C, ldo=3 * C, bias=b, rowbias=rb, rows_per_batch=N,
                           ln_colsum=cs, **self._ln_args(tok, rows, C))
                else:
                    n = self.layer_norm(tok.t, a.ln, rows, C, pe=a.pe, pe_div=N, pe_rows=g["F"])
                    qkv = self.lin(n, a.qkv_w, rows)
                att = self.new(rows, C)
                o.temporal_attention(qkv, att, clips=g["B"], frames=g["F"], pixels=N, heads=Hm, d=d, scale=d ** -0.5)
                tok = self._lin_rp(att, a.o_w, rows, bias=a.o_b, residual=tok.t)
            last = bi == len(m.blocks) - 1
            tok = self.feed_forward_out(blk.ff, blk.ff_ln, tok, x.t if last else None, rows, C, nxt if last else None)   # inner blocks: identity projection
        return tok

    # ---- forward -------------------------------------------------------------------------------
    def forward(self, x: Tensor, temb: Tensor, B: int, F: int, H: int, W: int, temb_first: Optional[Tensor] = None) -> Tensor:
        """x: channels-last model input [B*F*H*W][pad64(conv_in_channels)] (B already includes the CFG
        duplicate); temb: [B, temb_total] f32 rows of the current step.  Returns [B*F*H*W][out_channels].
        temb_first ([1, temb_total], optional): time-embedding row of timestep 0 that every ResNet adds to FRAME 0 instead of
        the clip's own row (reference `use_first_frame_condition`, unet.py:523-524, resnet.py:310-317)."""
        assert self.ctx_cache is not None, "call prepare_context() first"
        cfg, P = self.cfg, self.P
        g = dict(B=B, F=F, H=H, W=W, rows=B * F * H * W)
        if temb_first is not None:
            temb = temb.repeat_interleave(F, dim=0)            # one row per (clip, frame)
            temb[0::F] = temb_first
            g["temb_per_frame"] = True
        # (use_first_frame_condition_concat, reference unet.py:580-590: the 8-channel input is built by fyc_unet_input mode 1 and the
        # `sample / 2` behind conv_in is folded into its packed weights, engine/weights.py)
        frames = B * F

        def hw(gg):         # rows of one frame = one statistics sample of every producer
            return gg["H"] * gg["W"]

        def frame(gg):      # the consumer is a per-frame GroupNorm (transformer / motion module)
            return hw(gg), hw(gg)

        def clip(gg):       # the consumer is a cross-frame GroupNorm (ResNet norm1 / conv_norm_out)
            return hw(gg), F * hw(gg)

        def layer(l, xin, gg):
            """resnet [-> transformer] [-> motion module]; each stage tells its producer which norm consumes its output"""
            y = self.resnet(l.resnet, xin, temb, gg, frame(gg) if (l.attn is not None or l.motion is not None) else clip(gg))
            if l.attn is not None:
                y = self.transformer(l.attn, y, gg, frame(gg) if l.motion is not None else clip(gg))
            if l.motion is not None:
                y = self.motion(l.motion, y, gg, clip(gg))
            return y

        x = self._conv_act(x, P.conv_in_w, P.conv_in_b, frames, H, W, clip(g))
        skips = [x]
        sizes = [(H, W)]                       # spatial size per resolution level (odd sizes: ceil-halving on the way down)
        for blk in P.down:
            for l in blk.layers:
                x = layer(l, x, g)
                skips.append(x)
            if blk.down is not None:
                g2 = dict(g, H=(g["H"] - 1) // 2 + 1, W=(g["W"] - 1) // 2 + 1)
                g2["rows"] = B * F * g2["H"] * g2["W"]
                x = self._conv_act(x.t, blk.down.w, blk.down.b, frames, g["H"], g["W"], clip(g2), stride=2)
                g = g2
                sizes.append((g["H"], g["W"]))
                skips.append(x)
        x = self.resnet(P.mid.r0, x, temb, g, frame(g))
        x = self.transformer(P.mid.attn, x, g, frame(g) if P.mid.motion is not None else clip(g))
        if P.mid.motion is not None:
            x = self.motion(P.mid.motion, x, g, clip(g))
        x = self.resnet(P.mid.r1, x, temb, g, clip(g))
        for blk in P.up:
            for l in blk.layers:
                x = layer(l, (x, skips.pop()), g)      # cat([hidden, skip], dim=1) folded into norm1 / the shortcut
            if blk.up is not None:
                # Upsample3D: nearest to the size of the next skip connection (exactly 2x unless a level was odd:
                # the reference forwards `upsample_size`, unet.py:466-474, 644-645)
                sizes.pop()
                Hn, Wn = sizes[-1]
                g2 = dict(g, H=Hn, W=Wn)
                g2["rows"] = B * F * Hn * Wn
                x = self._conv_act(x.t, blk.up.w, blk.up.b, frames, g["H"], g["W"], clip(g2), up2=True, up_size=(Hn, Wn))
                g = g2
        h, _ = self._gn(x, P.out_g, P.out_b, g["rows"], F * hw(g), cfg.norm_eps, True)
        return self.conv(h, P.conv_out_w, P.conv_out_b, frames, g["H"], g["W"])
