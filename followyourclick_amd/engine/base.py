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

