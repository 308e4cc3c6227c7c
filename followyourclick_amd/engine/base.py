"""Shared host helpers of the UNet / VAE engines: buffer allocation and thin op wrappers."""
from __future__ import annotations

import torch

from .. import _lib as L

Tensor = torch.Tensor

# Debug aid: ALLOC_FILL = a byte value makes every buffer the engines allocate start as that byte pattern instead of whatever the
# caching allocator hands back (0xFF = NaN in bf16 / f16 / f32 / f64).  A kernel that reads memory nobody wrote - rows past M, partial-sum
# slots the producer skipped - then shows up as a result that depends on the fill: tests/test_uninit_gpu.py runs the forward under two
# fills and demands identical bits.  FYC_ALLOC_FILL=255 sets it from the environment.
import os as _os
ALLOC_FILL = int(_os.environ["FYC_ALLOC_FILL"]) if _os.environ.get("FYC_ALLOC_FILL") else None


def empty(*shape, dtype, device) -> Tensor:
    t = torch.empty(*shape, dtype=dtype, device=device)
    if ALLOC_FILL is not None and t.numel():
        t.view(torch.uint8).fill_(ALLOC_FILL)
    return t


class EngineBase:
    """Expects self.ops, self.dtype, self.device and self.groups (GroupNorm group count)."""

    # ---- buffers -----------------------------------------------------------------------------
    def new(self, *shape, dtype=None) -> Tensor:
        return empty(*shape, dtype=dtype or self.dtype, device=self.device)

    def zeros(self, *shape, dtype=None) -> Tensor:
        return torch.zeros(*shape, dtype=dtype or self.dtype, device=self.device)

    # ---- fused statistics ------------------------------------------------------------------------
    def _cs_plan(self, rows: int, rows_per_sample: int, N: int, K: int, mode: int):
        """row-tile partial buffer for the producer's epilogue: (parts, tile_rows, slots), or None when the shape cannot be fused
        (rows per frame not a multiple of 16, or a row tile would touch more than 4 frames)"""
        if not getattr(self, "fuse_stats", True) or rows_per_sample % 16 or rows % rows_per_sample:
            return None
        # small M, long K - the 8x8 latents - run split-K: since round 6 (ABI 301) its finish kernel writes the same partial sums and
        # fyc_gemm_stat_layout answers for it; an older A/B library (FYC_LIB_PATH) keeps the separate statistics pass
        if getattr(self.ops, "abi_version", 301) < 301 and self.ops.gemm_split_bytes(self.dtype, M=rows, N=N, K=K, mode=mode) > 0:
            return None
        nt, tile_rows, slots = self.ops.gemm_stat_layout(self.dtype, M=rows, N=N, K=K, cs_rows=rows_per_sample, mode=mode)
        if not 1 <= slots <= 4:
            return None
        return self.new(nt * slots * N * 2, dtype=torch.float32), tile_rows, slots

    def _cs_finish(self, plan, rows: int, rows_per_sample: int, N: int, out_rows: int) -> Tensor:
        """per-(GroupNorm sample, channel) f64 sums from the epilogue's row-tile partials; out_rows = rows of the consuming
        norm's sample (a frame, or the F frames of a clip)"""
        parts, tile_rows, slots = plan
        cs = self.new(rows // out_rows, N, 2, dtype=torch.float64)
        self.ops.chan_stats_reduce(parts, cs, rows=rows, N=N, cs_rows=rows_per_sample, tile_rows=tile_rows, slots=slots, out_rows=out_rows)
        return cs

    # ---- primitive helpers -------------------------------------------------------------------
    def lin(self, x: Tensor, w: Tensor, rows: int, bias=None, residual=None, geglu=False, out=None) -> Tensor:
        N, K = w.shape
        cols = N // 2 if geglu else N
        out = out if out is not None else self.new(rows, cols, dtype=x.dtype)
        self.ops.gemm(x, w, out, M=rows, N=N, K=K, lda=K, ldw=K, ldo=cols, bias=bias, residual=residual, ldr=cols,
                      epilogue=L.EPI_GEGLU if geglu else L.EPI_LINEAR)
        return out

    def conv(self, x: Tensor, w: Tensor, b: Tensor, frames: int, Hin: int, Win: int, stride: int = 1, up2: bool = False,
             rowbias=None, rpb: int = 1, residual=None, ldrb: int = 0, up_size=None, pad: int = 1, chan_parts=None, cs_rows: int = 0) -> Tensor:
        Cout, K = w.shape
        Cin = K // 9
        if up2:
            Ho, Wo = up_size if up_size is not None else (2 * Hin, 2 * Win)
        else:
            Ho, Wo = (Hin + pad - 2) // stride + 1, (Win + pad - 2) // stride + 1
        M = frames * Ho * Wo
        out = self.new(M, Cout)
        self.ops.gemm(x, w, out, M=M, N=Cout, K=K, lda=Cin, ldw=K, ldo=Cout, bias=b, rowbias=rowbias, rows_per_batch=rpb, ldrb=ldrb,
                      residual=residual, ldr=Cout, mode=L.GEMM_CONV3X3_UP2 if up2 else L.GEMM_CONV3X3,
                      conv=dict(Hout=Ho, Wout=Wo, Hin=Hin, Win=Win, Cin=Cin, stride=stride, pad=pad), chan_parts=chan_parts, cs_rows=cs_rows)
        return out

    def group_norm(self, x: Tensor, g: Tensor, b: Tensor, rows: int, C: int, rows_per_sample: int, eps: float, silu: bool) -> Tensor:
        stats = self.new(rows // rows_per_sample, self.groups, 2, dtype=torch.float64)
        self.ops.gn_stats(x, stats, rows=rows, C_=C, groups=self.groups, rows_per_sample=rows_per_sample)
        y = self.new(rows, C)
        self.ops.gn_apply(x, stats, g, b, y, rows=rows, C_=C, groups=self.groups, rows_per_sample=rows_per_sample,
                          eps=eps, silu=silu)
        return y

    def layer_norm(self, x: Tensor, gb, rows: int, C: int, pe=None, pe_div: int = 1, pe_rows: int = 1) -> Tensor:
        y = self.new(rows, C)
        self.ops.layernorm(x, gb[0], gb[1], y, rows=rows, C_=C, eps=1e-5, pe=pe, pe_div=pe_div, pe_rows=pe_rows)
        return y

