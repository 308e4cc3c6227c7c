"""AutoencoderKL decoder as an op schedule over channels-last activations.

Restates AutoencoderKL.decode / Decoder.forward (reference diffusers/models/vae.py:208-224, 575-610),
ResnetBlock2D without temb (diffusers/models/resnet.py:454-493), Upsample2D (:108-143, nearest x2 folded
into the conv gather), the single-head mid AttentionBlock (diffusers/models/attention.py:328-379) and
AnimationPipeline.decode_latents (animatediff/pipelines/pipeline_animation.py:400-413).  The reference
decodes one frame per call; here all frames of a chunk go through the same launches.
"""
from __future__ import annotations

import os

import torch

from .. import _lib as L
from .. import ops as ops_mod
from .base import EngineBase
from .config import VAEDecoderConfig
from .weights import Packed, pad_channels

FUSE_STATS = os.environ.get("FYC_VAE_FUSE_STATS", "1") != "0"      # GroupNorm statistics from the producing convolution's epilogue (A/B switch)

Tensor = torch.Tensor


class VAEDecoderEngine(EngineBase):
    def __init__(self, packed: Packed, ops=None):
        self.P = packed
        self.cfg: VAEDecoderConfig = packed.cfg
        self.dtype = packed.dtype
        self.ops = ops if ops is not None else ops_mod.get()
        self.device = packed.conv_in_w.device
        self.groups = self.cfg.norm_num_groups
        self.ops.ensure_init(self.device)

    # ---- GroupNorm statistics from the producing convolution's epilogue (round 6: the decode used to run a statistics pass over every
    # GroupNorm input - 30 passes, 3.3 of 49 ms per 16-frame chunk at 512^2; the UNet engine has taken them from the epilogues since round 2) ----
    def conv_cs(self, x: Tensor, w: Tensor, b: Tensor, frames: int, Hin: int, Win: int, **kw):
        """a 3x3 convolution whose output feeds a per-frame GroupNorm -> (output, per-(frame, channel) sums or None)"""
        Cout, K = w.shape
        up2, stride, pad = kw.get("up2", False), kw.get("stride", 1), kw.get("pad", 1)
        Ho, Wo = (2 * Hin, 2 * Win) if up2 else ((Hin + pad - 2) // stride + 1, (Win + pad - 2) // stride + 1)
        rows, hw = frames * Ho * Wo, Ho * Wo
        plan = self._cs_plan(rows, hw, Cout, K, L.GEMM_CONV3X3_UP2 if up2 else L.GEMM_CONV3X3) if FUSE_STATS else None
        out = self.conv(x, w, b, frames, Hin, Win, chan_parts=None if plan is None else plan[0], cs_rows=hw if plan is not None else 0, **kw)
        return out, (None if plan is None else self._cs_finish(plan, rows, hw, Cout, hw))

    def gn(self, x: Tensor, cs, g: Tensor, b: Tensor, rows: int, C: int, hw: int, silu: bool) -> Tensor:
        if cs is None:
            return self.group_norm(x, g, b, rows, C, hw, 1e-6, silu)
        y = self.new(rows, C)
        self.ops.gn_apply_cs(x, cs, g, b, y, rows=rows, C1=C, groups=self.groups, rows_per_sample=hw, eps=1e-6, silu=silu, cs_rows=hw)
        return y

    def resnet(self, r: Packed, x: Tensor, frames: int, H: int, W: int, cs=None):
        """-> (output, its per-(frame, channel) sums or None); `cs`: the sums of x when its producer wrote them"""
        rows, hw = frames * H * W, H * W
        h = self.gn(x, cs, r.n1_g, r.n1_b, rows, r.cin, hw, True)
        h, cs1 = self.conv_cs(h, r.c1_w, r.c1_b, frames, H, W)
        h = self.gn(h, cs1, r.n2_g, r.n2_b, rows, r.cout, hw, True)
        sc = self.lin(x, r.sc_w, rows, bias=r.sc_b) if r.sc_w is not None else x
        return self.conv_cs(h, r.c2_w, r.c2_b, frames, H, W, residual=sc)

    def attention(self, a: Packed, x: Tensor, frames: int, H: int, W: int, cs=None) -> Tensor:
        """single head, d = C (512): scores materialised through the batched GEMM, softmax in f32"""
        o = self.ops
        rows, N = frames * H * W, H * W
        C = a.o_w.shape[0]
        h = self.gn(x, cs, a.g, a.b, rows, C, N, False)
        ld = ((N + 7) // 8) * 8
        q, k = self.new(frames, 1, N, C), self.new(frames, 1, N, C)
        vt = self.zeros(frames, 1, C, ld) if ld != N else self.new(frames, 1, C, ld)
        o.gemm(h, a.qkv_w, None, M=rows, N=3 * C, K=C, lda=C, ldw=C, bias=a.qkv_b, epilogue=L.EPI_HEADS,
               heads=dict(seg_cols=C, heads=1, tokens=N, outs=[q, k, vt], transposed=[0, 0, 1], ld=[0, 0, ld]))
        S = self.zeros(frames, N, ld) if ld != N else self.new(frames, N, ld)
        o.gemm(q, k, S, M=N, N=N, K=C, lda=C, ldw=C, ldo=ld, batch=frames, stride_a=N * C, stride_w=N * C, stride_o=N * ld,
               out_scale=C ** -0.5)
        o.softmax_rows(S, rows=frames * N, cols=N, ld=ld)
        att = self.new(rows, C)
        o.gemm(S, vt, att, M=N, N=C, K=ld, lda=ld, ldw=ld, ldo=C, batch=frames, stride_a=N * ld, stride_w=C * ld, stride_o=N * C)
        return self.lin(att, a.o_w, rows, bias=a.o_b, residual=x)

    def decode(self, z: Tensor, raw: bool = False) -> Tensor:
        """z: (N, 4, h, w) f32 latents *already in model space* (not yet divided by the scaling factor);
        returns (N, 3, 8h, 8w) f32 = clamp(decode(z / 0.18215) / 2 + 0.5, 0, 1), or with raw=True the decoder's own
        unclamped sample (what AutoencoderKL.decode returns, reference diffusers/models/vae.py:575-589)."""
        P, cfg, o = self.P, self.cfg, self.ops
        N, Cz, H, W = z.shape
        z = z.to(device=self.device, dtype=torch.float32).contiguous()
        cp = pad_channels(Cz)
        x = self.new(N * H * W, cp)
        o.nchw_to_nhwc(z, x, N=N, C_=Cz, HW=H * W, c_pad=cp, scale=1.0 / cfg.scaling_factor)
        y = self.zeros(N * H * W, cp)
        o.gemm(x, P.pq_w, y, M=N * H * W, N=Cz, K=cp, lda=cp, ldw=cp, ldo=cp, bias=P.pq_b)   # post_quant_conv (1x1)
        x, cs = self.conv_cs(y, P.conv_in_w, P.conv_in_b, N, H, W)
        x, cs = self.resnet(P.mid_r0, x, N, H, W, cs)
        x = self.attention(P.attn, x, N, H, W, cs)          # (its output projection writes no sums: mid_r1's first norm runs the statistics pass)
        x, cs = self.resnet(P.mid_r1, x, N, H, W)
        for blk in P.ups:
            for r in blk.resnets:
                x, cs = self.resnet(r, x, N, H, W, cs)
            if blk.up is not None:
                x, cs = self.conv_cs(x, blk.up.w, blk.up.b, N, H, W, up2=True)
                H, W = 2 * H, 2 * W
        c0 = cfg.block_out_channels[0]
        h = self.gn(x, cs, P.out_g, P.out_b, N * H * W, c0, H * W, True)
        Co = cfg.out_channels
        ldo = ((Co + 3) // 4) * 4
        img = self.new(N * H * W, ldo)
        Kc = P.conv_out_w.shape[1]
        o.gemm(h, P.conv_out_w, img, M=N * H * W, N=Co, K=Kc, lda=Kc // 9, ldw=Kc, ldo=ldo, bias=P.conv_out_b, mode=L.GEMM_CONV3X3,
               conv=dict(Hout=H, Wout=W, Hin=H, Win=W, Cin=Kc // 9, stride=1))
        out = self.new(N, Co, H, W, dtype=torch.float32)
        if raw:
            o.nhwc_to_nchw(img, out, N=N, C_=Co, HW=H * W, ld=ldo)
        else:
            o.nhwc_to_nchw(img, out, N=N, C_=Co, HW=H * W, ld=ldo, mul=0.5, add=0.5, lo=0.0, hi=1.0)
        return out

    def decode_video(self, latents: Tensor, chunk: int = 16) -> Tensor:
        """(b, 4, f, h, w) -> (b, 3, f, 8h, 8w) in [0, 1]  (decode_latents of the reference pipeline)."""
        B, C, F, H, W = latents.shape
        z = latents.to(self.device, torch.float32).permute(0, 2, 1, 3, 4).reshape(B * F, C, H, W).contiguous()
        outs = [self.decode(z[i:i + chunk]) for i in range(0, B * F, chunk)]
        vid = torch.cat(outs, 0) if len(outs) > 1 else outs[0]
        return vid.reshape(B, F, vid.shape[1], vid.shape[2], vid.shape[3]).permute(0, 2, 1, 3, 4)


class VAEEncoderEngine(VAEDecoderEngine):
    """AutoencoderKL.encode up to the Gaussian moments (reference diffusers/models/vae.py:128-144, 565-573;
    DownEncoderBlock2D unet_2d_blocks.py:870-930; Downsample2D with padding=0 -> F.pad (0,1,0,1),
    resnet.py:181-190).  The conditioning front-end row of SURVEY.md 8f.1: run once per clip on the first frame."""

    def encode_moments(self, x: Tensor) -> Tensor:
        """x: (N, 3, H, W) f32 image in [-1, 1]  ->  (N, 2*latent, H/8, W/8) f32 = [mean | logvar]"""
        P, cfg, o = self.P, self.cfg, self.ops
        N, Ci, H, W = x.shape
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        cp = pad_channels(Ci)
        h = self.new(N * H * W, cp)
        o.nchw_to_nhwc(x, h, N=N, C_=Ci, HW=H * W, c_pad=cp, scale=1.0)
        h, cs = self.conv_cs(h, P.conv_in_w, P.conv_in_b, N, H, W)
        for blk in P.downs:
            for r in blk.resnets:
                h, cs = self.resnet(r, h, N, H, W, cs)
            if blk.down is not None:
                h, cs = self.conv_cs(h, blk.down.w, blk.down.b, N, H, W, stride=2, pad=0)
                H, W = H // 2, W // 2
        h, cs = self.resnet(P.mid_r0, h, N, H, W, cs)
        h = self.attention(P.attn, h, N, H, W, cs)
        h, cs = self.resnet(P.mid_r1, h, N, H, W)
        c = cfg.block_out_channels[-1]
        h = self.gn(h, cs, P.out_g, P.out_b, N * H * W, c, H * W, True)
        C2 = 2 * cfg.latent_channels
        cp2 = pad_channels(C2)
        z = self.zeros(N * H * W, cp2)                      # conv_out into a zero-padded row so quant_conv (1x1) can read 128-B K tiles
        Kc = P.conv_out_w.shape[1]
        o.gemm(h, P.conv_out_w, z, M=N * H * W, N=C2, K=Kc, lda=Kc // 9, ldw=Kc, ldo=cp2, bias=P.conv_out_b, mode=L.GEMM_CONV3X3,
               conv=dict(Hout=H, Wout=W, Hin=H, Win=W, Cin=Kc // 9, stride=1, pad=1))
        m = self.new(N * H * W, 8 if C2 <= 8 else cp2)
        o.gemm(z, P.q_w, m, M=N * H * W, N=C2, K=cp2, lda=cp2, ldw=cp2, ldo=m.shape[1], bias=P.q_b)
        out = self.new(N, C2, H, W, dtype=torch.float32)
        o.nhwc_to_nchw(m, out, N=N, C_=C2, HW=H * W, ld=m.shape[1])
        return out
