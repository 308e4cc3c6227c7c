"""Tensor-level wrappers over the C ABI (include/fyc.h).

torch is plumbing here: it owns device memory and the stream; every op below only forwards raw
``data_ptr()``s and sizes to libfyc_hip.so.  There is no CPU implementation in the product: a tensor
that is not on a HIP device raises.  (The engine reaches these functions through the module-level
``impl`` object so that the CPU test-suite can exercise the host orchestration with an op emulator
that lives under tests/ - never in the product path.)
"""
from __future__ import annotations

import os

import ctypes as C
from typing import Optional

import torch

from . import _lib as L

Tensor = torch.Tensor


_DTYPE_CODES = {torch.bfloat16: L.FYC_BF16, torch.float16: L.FYC_F16, torch.float32: L.FYC_F32}


def _dtc(dtype: torch.dtype) -> int:
    """torch dtype of the activations / weights -> fyc_dtype (include/fyc.h): bf16 production, f16 (the reference's deployed
    fp16-autocast precision class), f32 parity mode"""
    try:
        return _DTYPE_CODES[dtype]
    except KeyError:
        raise TypeError(f"unsupported activation dtype {dtype}") from None


def _dt(t: Tensor) -> int:
    return _dtc(t.dtype)


def _p(t: Optional[Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise L.FycError("followyourclick_amd ops need HIP device tensors; there is no CPU fallback")
    return t.data_ptr()


def _f32(t: Optional[Tensor], what: str) -> Optional[int]:
    if t is not None and t.dtype != torch.float32:
        raise TypeError(f"{what} must be float32")
    return _p(t)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda idx: torch.cuda.current_stream(idx).cuda_stream)


class HipOps:
    """The product backend: libfyc_hip.so on the current torch HIP stream."""

    name = "hip"

    def __init__(self):
        self.lib = L.load()
        self._zero = None
        self._ws = None              # scratch for split-K GEMMs: one buffer, grown on demand (all work is on one stream)
        self._ws_need = {}           # (shape, flags) -> split-K scratch bytes (fyc_gemm_workspace_bytes), see gemm()
        self._q_cache = {}           # host-side layout queries (row parts, split bytes, statistics layout) per shape
        self._inited_dev = None

    # -- library ---------------------------------------------------------------------------
    def ensure_init(self, device: torch.device) -> None:
        """One GPU per process (the library's zero page and tuning table are process-wide; the multi-GPU path is one process
        per GPU): the first device used is bound, any other raises."""
        if self._inited_dev is not None and getattr(device, "index", None) == self._inited_dev:
            return                                             # the ~900 calls of a DDIM step take this exit
        if not torch.cuda.is_available():
            raise L.FycError("no HIP device visible: followyourclick_amd has no CPU fallback")
        device = torch.device(device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if self._inited_dev == idx:
            return
        if self._inited_dev is not None:
            raise L.FycError(f"followyourclick_amd is bound to cuda:{self._inited_dev}; use one process per GPU (asked for cuda:{idx})")
        self._zero = torch.zeros(4096, dtype=torch.uint8, device=torch.device("cuda", idx))
        L.check(self.lib.fyc_init(self._zero.data_ptr()), "fyc_init")
        self.abi_version = int(self.lib.fyc_version())        # (differs from _lib.FYC_VERSION only for an A/B library, FYC_LIB_PATH)
        self._inited_dev = idx
        for kv in filter(None, os.environ.get("FYC_TUNING", "").split(",")):   # A/B runs: FYC_TUNING="5=1,4=8" (fyc_set_tuning keys)
            k, v = kv.split("=")
            self.set_tuning(int(k), int(v))

    def device_caps(self):
        caps = (L.i64 * 8)()
        L.check(self.lib.fyc_device_caps(caps), "fyc_device_caps")
        return list(caps)

    def set_tuning(self, key: int, value: int) -> None:
        L.check(self.lib.fyc_set_tuning(key, value), "fyc_set_tuning")
        self._ws_need.clear()        # tile / split-K decisions depend on the tuning table
        self._q_cache.clear()

    def _stream(self) -> int:
        """raw hipStream_t of torch's CURRENT stream on the bound device (a `with torch.cuda.stream(...)` block is honoured);
        the raw getter costs ~0.3 us, torch.cuda.current_stream().cuda_stream ~8 us - a third of the host time of a launch"""
        return _raw_stream(self._inited_dev if self._inited_dev is not None else torch.cuda.current_device())

    def _call(self, name: str, args) -> None:
        L.check(getattr(self.lib, name)(C.byref(args), self._stream()), name)

    # -- GEMM family -------------------------------------------------------------------------
    def gemm(self, a: Tensor, w: Tensor, out: Optional[Tensor], *, M: int, N: int, K: int, lda: int, ldw: int,
             ldo: int = 0, bias: Optional[Tensor] = None, rowbias: Optional[Tensor] = None, rows_per_batch: int = 1,
             residual: Optional[Tensor] = None, ldr: int = 0, ldrb: int = 0, out_scale: float = 1.0, epilogue: int = L.EPI_LINEAR,
             mode: int = L.GEMM_PLAIN, conv: Optional[dict] = None, batch: int = 1, stride_a: int = 0,
             stride_w: int = 0, stride_o: int = 0, heads: Optional[dict] = None, tile: int = 0,
             a2: Optional[Tensor] = None, k_split: int = 0, lda2: int = 0, act: int = L.ACT_NONE,
             ln_stats: Optional[Tensor] = None, ln_colsum: Optional[Tensor] = None, ln_nparts: int = 0, ln_eps: float = 1e-5,
             chan_parts: Optional[Tensor] = None, cs_rows: int = 0, row_parts: Optional[Tensor] = None, row_nparts: int = 0) -> None:
        self.ensure_init(a.device)
        g = self._gemm_args(a, w, out, M=M, N=N, K=K, lda=lda, ldw=ldw, ldo=ldo, bias=bias, rowbias=rowbias, rows_per_batch=rows_per_batch,
                            residual=residual, ldr=ldr, ldrb=ldrb, out_scale=out_scale, epilogue=epilogue, mode=mode, conv=conv, batch=batch,
                            stride_a=stride_a, stride_w=stride_w, stride_o=stride_o, heads=heads, tile=tile, a2=a2, k_split=k_split,
                            lda2=lda2, act=act, ln_stats=ln_stats, ln_colsum=ln_colsum, ln_nparts=ln_nparts, ln_eps=ln_eps,
                            chan_parts=chan_parts, cs_rows=cs_rows, row_parts=row_parts, row_nparts=row_nparts)
        # split-K scratch (small M, long K): the decision depends on the shape and the flags below only - cached, so that the ~300
        # GEMM launches of a DDIM step do not each pay a second ctypes call.  One grown-on-demand buffer: all work of a process is on
        # ONE stream (the engine's), a second stream would need its own HipOps.
        key = (M, N, K, mode, epilogue, act, batch, tile, g.dtype, ln_stats is not None, chan_parts is not None, row_parts is not None)
        need = self._ws_need.get(key)
        if need is None:
            need = int(self.lib.fyc_gemm_workspace_bytes(C.byref(g))) if hasattr(self.lib, "fyc_gemm_workspace_bytes") else 0
            self._ws_need[key] = need
        if need > 0:
            if self._ws is None or self._ws.numel() < need or self._ws.device != a.device:
                self._ws = torch.empty(need, dtype=torch.uint8, device=a.device)
            g.workspace, g.workspace_bytes = self._ws.data_ptr(), self._ws.numel()
        self._call("fyc_gemm", g)

    def gemm_row_parts(self, dtype: torch.dtype, *, M: int, N: int, K: int, mode: int = L.GEMM_PLAIN, batch: int = 1, tile: int = 0) -> int:
        """column tiles fyc_gemm will use for this problem = row_nparts of its `row_parts` output (fyc_gemm_row_parts)"""
        key = ("rp", dtype, M, N, K, mode, batch, tile)
        if key not in self._q_cache:
            g = L.GemmArgs()
            g.M, g.N, g.K, g.mode, g.batch, g.tile = M, N, K, mode, batch, tile
            g.dtype = _dtc(dtype)
            self._q_cache[key] = int(self.lib.fyc_gemm_row_parts(C.byref(g))) if hasattr(self.lib, "fyc_gemm_row_parts") else 0
        return self._q_cache[key]

    def gemm_split_bytes(self, dtype: torch.dtype, *, M: int, N: int, K: int, mode: int = L.GEMM_PLAIN) -> int:
        """scratch bytes a plain LINEAR-epilogue problem of this shape would use for split-K (0: it runs unsplit)"""
        key = ("sb", dtype, M, N, K, mode)
        if key not in self._q_cache:
            g = L.GemmArgs()
            g.M, g.N, g.K, g.mode, g.batch, g.epilogue = M, N, K, mode, 1, L.EPI_LINEAR
            g.dtype = _dtc(dtype)
            self._q_cache[key] = int(self.lib.fyc_gemm_workspace_bytes(C.byref(g))) if hasattr(self.lib, "fyc_gemm_workspace_bytes") else 0
        return self._q_cache[key]

    def gemm_stat_layout(self, dtype: torch.dtype, *, M: int, N: int, K: int, cs_rows: int, mode: int = L.GEMM_PLAIN, batch: int = 1,
                         tile: int = 0):
        """(row tiles, rows per tile, sample slots per tile) of the `chan_parts` output of this problem (fyc_gemm_stat_layout)"""
        key = ("sl", dtype, M, N, K, cs_rows, mode, batch, tile)
        if key not in self._q_cache:
            if not hasattr(self.lib, "fyc_gemm_stat_layout"):       # an older A/B library (FYC_LIB_PATH): no fused statistics
                self._q_cache[key] = (0, 0, 0)
            else:
                g = L.GemmArgs()
                g.M, g.N, g.K, g.mode, g.batch, g.tile, g.cs_rows = M, N, K, mode, batch, tile, cs_rows
                g.dtype = _dtc(dtype)
                tr, sl = L.i32(0), L.i32(0)
                n = int(self.lib.fyc_gemm_stat_layout(C.byref(g), C.byref(tr), C.byref(sl)))
                self._q_cache[key] = (n, int(tr.value), int(sl.value))
        return self._q_cache[key]

    def chan_stats_reduce(self, parts: Tensor, cs: Tensor, *, rows: int, N: int, cs_rows: int, tile_rows: int, slots: int,
                          out_rows: int = 0) -> None:
        if cs.dtype != torch.float64:
            raise TypeError("channel statistics must be float64")
        a = L.ChanStatsReduceArgs()
        a.parts, a.cs, a.rows, a.N, a.cs_rows, a.tile_rows, a.slots = _f32(parts, "parts"), _p(cs), rows, N, cs_rows, tile_rows, slots
        a.out_rows = out_rows
        self._call("fyc_chan_stats_reduce", a)

    @staticmethod
    def _gemm_args(a, w, out, *, M, N, K, lda, ldw, ldo, bias, rowbias, rows_per_batch, residual, ldr, ldrb, out_scale, epilogue, mode,
                   conv, batch, stride_a, stride_w, stride_o, heads, tile, a2, k_split, lda2, act, ln_stats, ln_colsum, ln_nparts,
                   ln_eps, chan_parts, cs_rows, row_parts, row_nparts):
        g = L.GemmArgs()
        g.a, g.w, g.bias, g.rowbias = _p(a), _p(w), _f32(bias, "bias"), _f32(rowbias, "rowbias")
        g.residual, g.out = _p(residual), _p(out)
        g.M, g.N, g.K, g.lda, g.ldw, g.ldo, g.ldr, g.ldrb = M, N, K, lda, ldw, ldo, ldr, ldrb
        g.stride_a, g.stride_w, g.stride_o, g.batch = stride_a, stride_w, stride_o, batch
        g.mode, g.epilogue, g.rows_per_batch, g.out_scale, g.dtype = mode, epilogue, rows_per_batch, out_scale, _dt(a)
        g.tile, g.act = tile, act
        g.ln_stats, g.ln_colsum = _f32(ln_stats, "ln_stats"), _f32(ln_colsum, "ln_colsum")
        g.a2, g.k_split, g.lda2 = _p(a2), k_split, lda2
        if a.dtype != w.dtype:
            raise TypeError(f"gemm: activation {a.dtype} vs weight {w.dtype}")
        if conv is not None:
            g.Hout, g.Wout, g.Hin, g.Win, g.Cin, g.conv_stride = (conv["Hout"], conv["Wout"], conv["Hin"], conv["Win"],
                                                                  conv["Cin"], conv.get("stride", 1))
            g.conv_pad = conv.get("pad", 1)
        if heads is not None:
            g.seg_cols, g.heads, g.tokens = heads["seg_cols"], heads["heads"], heads["tokens"]
            for i, (t, tr, ld) in enumerate(zip(heads["outs"], heads["transposed"], heads["ld"])):
                g.seg_out[i], g.seg_transposed[i], g.seg_ld[i] = _p(t), int(tr), int(ld)
        g.ln_nparts, g.ln_eps = ln_nparts, ln_eps
        g.chan_parts, g.cs_rows = _f32(chan_parts, "chan_parts"), cs_rows
        g.row_parts, g.row_nparts = _f32(row_parts, "row_parts"), row_nparts
        return g

    # -- attention ---------------------------------------------------------------------------
    def attention(self, q: Tensor, k: Tensor, vt: Tensor, o: Tensor, *, batch: int, heads: int, n_q: int, n_k: int,
                  d: int, ldo: int, ldvt: int, scale: float, kv_batch_div: int = 1, accumulate: bool = False,
                  o_scale: float = 1.0) -> None:
        self.ensure_init(q.device)
        a = L.AttnArgs()
        a.q, a.k, a.vt, a.o = _p(q), _p(k), _p(vt), _p(o)
        a.batch, a.heads, a.n_q, a.n_k, a.d, a.ldo, a.ldvt = batch, heads, n_q, n_k, d, ldo, ldvt
        a.kv_batch_div, a.o_accumulate, a.scale, a.o_scale, a.dtype = kv_batch_div, int(accumulate), scale, o_scale, _dt(q)
        self._call("fyc_attention", a)

    def temporal_attention(self, qkv: Tensor, o: Tensor, *, clips: int, frames: int, pixels: int, heads: int, d: int,
                           scale: float) -> None:
        self.ensure_init(qkv.device)
        a = L.TAttnArgs()
        a.qkv, a.o, a.clips, a.frames, a.pixels, a.heads, a.d, a.scale, a.dtype = (_p(qkv), _p(o), clips, frames, pixels,
                                                                                  heads, d, scale, _dt(qkv))
        self._call("fyc_temporal_attention", a)

    def _tblock_args(self, x, out, w_qkv, colsum, bias, pe_bias, w_out, b_out, clips, frames, pixels, heads, d, scale, eps):
        a = L.TemporalBlockArgs()
        a.x, a.out, a.w_qkv, a.w_out = _p(x), _p(out), _p(w_qkv), _p(w_out)
        a.colsum, a.bias, a.pe_bias, a.b_out = _f32(colsum, "colsum"), _f32(bias, "bias"), _f32(pe_bias, "pe_bias"), _f32(b_out, "b_out")
        a.clips, a.frames, a.pixels, a.heads, a.d, a.C = clips, frames, pixels, heads, d, heads * d
        a.scale, a.eps, a.dtype = scale, eps, _dt(x)
        return a

    def temporal_block_supported(self, dtype: torch.dtype, *, clips: int, frames: int, pixels: int, heads: int, d: int) -> bool:
        """does fyc_temporal_block (the fused temporal sub-block) cover this shape?  (no launch)"""
        if not hasattr(self.lib, "fyc_temporal_block_supported"):
            return False
        a = L.TemporalBlockArgs()
        a.clips, a.frames, a.pixels, a.heads, a.d, a.C = clips, frames, pixels, heads, d, heads * d
        a.dtype = _dtc(dtype)
        return bool(self.lib.fyc_temporal_block_supported(C.byref(a)))

    def temporal_block(self, x: Tensor, out: Tensor, *, w_qkv: Tensor, colsum: Tensor, bias: Tensor, pe_bias: Optional[Tensor],
                       w_out: Tensor, b_out: Tensor, clips: int, frames: int, pixels: int, heads: int, d: int, scale: float,
                       eps: float = 1e-5, wstream: Optional[Tensor] = None) -> None:
        """out = x + Attn_F(LayerNorm(x) + pe) Wo^T + bo in one kernel (csrc/temporal_block.hip); with `wstream`
        (engine/weights.py::pack_temporal_stream) the register-resident kernel (csrc/temporal_block_rr.hip) runs instead"""
        self.ensure_init(x.device)
        a = self._tblock_args(x, out, w_qkv, colsum, bias, pe_bias, w_out, b_out, clips, frames, pixels, heads, d, scale, eps)
        if wstream is not None and hasattr(self.lib, "fyc_temporal_block_wstream_bytes"):
            if "tbw" not in self._q_cache:
                self._q_cache["tbw"] = int(self.lib.fyc_temporal_block_wstream_bytes())
            if wstream.numel() * wstream.element_size() != self._q_cache["tbw"]:
                raise ValueError(f"temporal_block: wstream has {wstream.numel() * wstream.element_size()} bytes, expected {self._q_cache['tbw']}")
            a.wstream = _p(wstream)
        self._call("fyc_temporal_block", a)

    def ff_block_supported(self, dtype: torch.dtype, *, rows: int, C_: int, hidden: int, cs_rows: int = 0) -> bool:
        """does fyc_ff_block (the fused GEGLU feed-forward block) cover this shape?  (no launch); cs_rows > 0: with the fused
        output statistics for a GroupNorm whose statistics sample has cs_rows rows"""
        if not hasattr(self.lib, "fyc_ff_block_supported"):
            return False
        key = ("ffs", dtype, rows, C_, hidden, cs_rows)
        if key in self._q_cache:
            return self._q_cache[key]
        a = L.FFBlockArgs()
        a.rows, a.C, a.hidden, a.cs_rows = rows, C_, hidden, cs_rows
        a.chan_parts = 16 if cs_rows > 0 else None         # only tested for null / alignment by the query
        a.dtype = _dtc(dtype)
        self._q_cache[key] = bool(self.lib.fyc_ff_block_supported(C.byref(a)))
        return self._q_cache[key]

    def ff_block_wstream_bytes(self) -> int:
        if "ffw" not in self._q_cache:
            self._q_cache["ffw"] = int(self.lib.fyc_ff_block_wstream_bytes())
        return self._q_cache["ffw"]

    def ff_block(self, x: Tensor, residual: Optional[Tensor], out: Tensor, *, wstream: Tensor, b_out: Tensor, rows: int, C_: int,
                 hidden: int, eps: float = 1e-5, chan_parts: Optional[Tensor] = None, cs_rows: int = 0) -> None:
        """out = residual + b_out + [x | GEGLU(LN(x) W1^T + b1)] [Wp | Wp W2]^T in one kernel (csrc/ff_block.hip); `wstream`
        from engine/weights.py::pack_ff_block"""
        self.ensure_init(x.device)
        if wstream.numel() * wstream.element_size() != self.ff_block_wstream_bytes():
            raise ValueError(f"ff_block: wstream has {wstream.numel() * wstream.element_size()} bytes, expected {self.ff_block_wstream_bytes()}")
        a = L.FFBlockArgs()
        a.x, a.residual, a.out, a.wstream, a.b_out = _p(x), _p(residual), _p(out), _p(wstream), _f32(b_out, "b_out")
        a.chan_parts, a.cs_rows = _f32(chan_parts, "chan_parts"), cs_rows
        a.rows, a.C, a.hidden, a.eps, a.dtype = rows, C_, hidden, eps, _dt(x)
        self._call("fyc_ff_block", a)

    def panel_linear_supported(self, dtype: torch.dtype, *, rows: int, N: int, K: int, gn_rows_per_sample: int = 0, gn_groups: int = 32) -> bool:
        """does fyc_panel_linear (row-panel linear, optionally with the input's GroupNorm on the operand registers) cover this shape?"""
        if not hasattr(self.lib, "fyc_panel_linear_supported"):
            return False
        key = ("pls", dtype, rows, N, K, gn_rows_per_sample, gn_groups)
        if key not in self._q_cache:
            a = L.PanelLinearArgs()
            a.rows, a.N, a.K = rows, N, K
            a.dtype = _dtc(dtype)
            if gn_rows_per_sample > 0:
                a.gn_cs, a.gn_rows_per_sample, a.gn_stat_samples, a.gn_groups = 16, gn_rows_per_sample, 1, gn_groups     # pointer only tested for null
            self._q_cache[key] = bool(self.lib.fyc_panel_linear_supported(C.byref(a)))
        return self._q_cache[key]

    def panel_linear(self, x: Tensor, out: Tensor, *, wstream: Tensor, rows: int, N: int, K: int, bias: Optional[Tensor] = None,
                     residual: Optional[Tensor] = None, gn_cs: Optional[Tensor] = None, gn_gamma: Optional[Tensor] = None,
                     gn_beta: Optional[Tensor] = None, gn_rows_per_sample: int = 0, gn_stat_samples: int = 1, gn_groups: int = 32,
                     gn_eps: float = 1e-6) -> None:
        """out = [GroupNorm](x) W^T + bias (+ residual) in one kernel (csrc/panel_linear.hip); `wstream` from
        engine/weights.py::pack_panel_linear; gn_cs = the f64 per-(statistics sample, channel) sums of x"""
        self.ensure_init(x.device)
        key = ("plw", N, K)
        if key not in self._q_cache:
            self._q_cache[key] = int(self.lib.fyc_panel_linear_wstream_bytes(N, K))
        need = self._q_cache[key]
        if wstream.numel() * wstream.element_size() != need:
            raise ValueError(f"panel_linear: wstream has {wstream.numel() * wstream.element_size()} bytes, expected {need}")
        if gn_cs is not None and gn_cs.dtype != torch.float64:
            raise TypeError("panel_linear: gn_cs must be float64")
        a = L.PanelLinearArgs()
        a.x, a.residual, a.out, a.wstream, a.bias = _p(x), _p(residual), _p(out), _p(wstream), _f32(bias, "bias")
        a.gn_cs, a.gn_gamma, a.gn_beta = _p(gn_cs), _f32(gn_gamma, "gn_gamma"), _f32(gn_beta, "gn_beta")
        a.gn_rows_per_sample, a.gn_stat_samples, a.gn_groups, a.gn_eps = gn_rows_per_sample, gn_stat_samples, gn_groups, gn_eps
        a.rows, a.N, a.K, a.dtype = rows, N, K, _dt(x)
        self._call("fyc_panel_linear", a)

    # -- normalisation -----------------------------------------------------------------------
    def gn_stats(self, x: Tensor, stats: Tensor, *, rows: int, C_: int, groups: int, rows_per_sample: int) -> None:
        a = L.GnStatsArgs()
        if stats.dtype != torch.float64:
            raise TypeError("gn stats buffer must be float64")
        a.x, a.stats, a.rows, a.C, a.groups, a.rows_per_sample, a.dtype = _p(x), _p(stats), rows, C_, groups, rows_per_sample, _dt(x)
        key = ("gnws", rows, groups, rows_per_sample)
        need = self._q_cache.get(key)
        if need is None:
            # (an older A/B library - FYC_LIB_PATH - may predate the entry point: it then reduces without a workspace)
            need = self._q_cache[key] = int(self.lib.fyc_gn_stats_workspace(C.byref(a))) if hasattr(self.lib, "fyc_gn_stats_workspace") else 0
        if need > 0:     # chunk partial sums of the ordered (atomic-free) reduction: the split-K scratch buffer, same stream
            if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
                self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
            a.workspace, a.workspace_bytes = self._ws.data_ptr(), self._ws.numel()
        self._call("fyc_gn_stats", a)

    def gn_apply(self, x: Tensor, stats: Tensor, gamma: Tensor, beta: Tensor, y: Tensor, *, rows: int, C_: int,
                 groups: int, rows_per_sample: int, eps: float, silu: bool) -> None:
        a = L.GnApplyArgs()
        a.x, a.stats, a.gamma, a.beta, a.y = _p(x), _p(stats), _f32(gamma, "gamma"), _f32(beta, "beta"), _p(y)
        a.rows, a.C, a.groups, a.rows_per_sample, a.eps, a.silu, a.dtype = rows, C_, groups, rows_per_sample, eps, int(silu), _dt(x)
        self._call("fyc_gn_apply", a)

    def gn_apply_cs(self, x1: Tensor, cs1: Tensor, gamma: Tensor, beta: Tensor, y: Tensor, *, rows: int, C1: int, groups: int,
                    rows_per_sample: int, eps: float, silu: bool, x2: Optional[Tensor] = None, cs2: Optional[Tensor] = None,
                    C2: int = 0, cs_rows: int = 0) -> None:
        a = L.GnApplyCsArgs()
        for t in (cs1, cs2):
            if t is not None and t.dtype != torch.float64:
                raise TypeError("channel statistics must be float64")
        a.x1, a.cs1, a.x2, a.cs2 = _p(x1), _p(cs1), _p(x2), _p(cs2)
        a.gamma, a.beta, a.y = _f32(gamma, "gamma"), _f32(beta, "beta"), _p(y)
        a.C1, a.C2, a.rows, a.groups, a.rows_per_sample, a.eps, a.silu, a.dtype = C1, C2, rows, groups, rows_per_sample, eps, int(silu), _dt(x1)
        a.cs_rows = cs_rows
        self._call("fyc_gn_apply_cs", a)

    def layernorm(self, x: Tensor, gamma: Tensor, beta: Tensor, y: Tensor, *, rows: int, C_: int, eps: float = 1e-5,
                  pe: Optional[Tensor] = None, pe_div: int = 1, pe_rows: int = 1) -> None:
        a = L.LayerNormArgs()
        a.x, a.gamma, a.beta, a.pe, a.y = _p(x), _f32(gamma, "gamma"), _f32(beta, "beta"), _f32(pe, "pe"), _p(y)
        a.rows, a.C, a.eps, a.pe_div, a.pe_rows, a.dtype = rows, C_, eps, pe_div, pe_rows, _dt(x)
        self._call("fyc_layernorm", a)

    def row_stats(self, x: Tensor, stats: Tensor, *, rows: int, C_: int, eps: float = 1e-5) -> None:
        a = L.RowStatsArgs()
        a.x, a.stats, a.rows, a.C, a.eps, a.dtype = _p(x), _f32(stats, "stats"), rows, C_, eps, _dt(x)
        self._call("fyc_row_stats", a)

    def softmax_rows(self, x: Tensor, *, rows: int, cols: int, ld: int, causal_rows: int = 0) -> None:
        a = L.SoftmaxArgs()
        a.x, a.rows, a.cols, a.ld, a.dtype, a.causal_rows = _p(x), rows, cols, ld, _dt(x), causal_rows
        self._call("fyc_softmax_rows", a)

    # -- elementwise / layout ------------------------------------------------------------------
    def concat_channels(self, a_: Tensor, b_: Tensor, y: Tensor, *, rows: int, c1: int, c2: int) -> None:
        a = L.ConcatArgs()
        a.a, a.b, a.y, a.rows, a.c1, a.c2, a.dtype = _p(a_), _p(b_), _p(y), rows, c1, c2, _dt(a_)
        self._call("fyc_concat_channels", a)

    def embed_tokens(self, ids: Tensor, table: Tensor, pos: Tensor, out: Tensor, *, rows: int, seq: int, C_: int) -> None:
        if ids.dtype != torch.int64 or table.dtype != torch.float32 or pos.dtype != torch.float32:
            raise TypeError("embed_tokens: ids must be int64, tables float32")
        a = L.EmbedArgs()
        a.ids, a.table, a.pos, a.out = _p(ids), _p(table), _p(pos), _p(out)
        a.rows, a.seq, a.C, a.vocab, a.dtype = rows, seq, C_, table.shape[0], _dt(out)
        self._call("fyc_embed_tokens", a)

    def patchify(self, image: Tensor, out: Tensor, *, B: int, Cin: int, H: int, W: int, P: int, ld: int) -> None:
        if image.dtype != torch.float32:
            raise TypeError("patchify: image must be float32 NCHW")
        a = L.PatchifyArgs()
        a.image, a.out, a.B, a.Cin, a.H, a.W, a.P, a.ld, a.dtype = _p(image), _p(out), B, Cin, H, W, P, ld, _dt(out)
        self._call("fyc_patchify", a)

    def pack_conv3x3(self, w: Tensor, out: Tensor) -> None:
        """(O, I, 3, 3) f32 on the device -> the K order of the implicit-GEMM conv (engine/weights.py::pack_conv3x3)"""
        a = L.PackConv3x3Args()
        a.w, a.out, a.O, a.I, a.dtype = _f32(w, "w"), _p(out), w.shape[0], w.shape[1], _dt(out)
        self._call("fyc_pack_conv3x3", a)

    def pack_geglu(self, w: Tensor, b: Optional[Tensor], w_out: Tensor, b_out: Optional[Tensor]) -> None:
        """ff.net.0.proj rows -> 16 value rows, their 16 gate rows, ... (engine/weights.py::pack_geglu)"""
        a = L.PackGegluArgs()
        a.w, a.b, a.w_out, a.b_out = _f32(w, "w"), _f32(b, "b"), _p(w_out), _f32(b_out, "b_out")
        a.O, a.I, a.dtype = w.shape[0], w.shape[1], _dt(w_out)
        self._call("fyc_pack_geglu", a)

    def silu_f32(self, x: Tensor, y: Tensor) -> None:
        a = L.SiluArgs()
        a.x, a.y, a.n = _f32(x, "x"), _f32(y, "y"), x.numel()
        self._call("fyc_silu_f32", a)

    def cast_from_f32(self, x: Tensor, y: Tensor, *, rows: int, cols: int, ld: int) -> None:
        a = L.CastArgs()
        a.x, a.y, a.rows, a.cols, a.ld, a.dtype = _f32(x, "x"), _p(y), rows, cols, ld, _dt(y)
        self._call("fyc_cast_from_f32", a)

    def cast_to_f32(self, x: Tensor, y: Tensor, *, rows: int, cols: int, ld: int) -> None:
        a = L.CastArgs()
        a.x, a.y, a.rows, a.cols, a.ld, a.dtype = _p(x), _f32(y, "y"), rows, cols, ld, _dt(x)
        self._call("fyc_cast_to_f32", a)

    def unet_input(self, latents: Tensor, mask: Optional[Tensor], first: Optional[Tensor], x: Tensor, *, B: int, F: int,
                   HW: int, c_latent: int, c_pad: int, cfg_dup: int, mask_frames: int = 1, mode: int = 0) -> None:
        self.ensure_init(latents.device)
        a = L.UnetInputArgs()
        a.latents, a.mask, a.first, a.x = _f32(latents, "latents"), _f32(mask, "mask"), _f32(first, "first"), _p(x)
        a.B, a.F, a.HW, a.c_latent, a.c_pad, a.cfg_dup, a.mask_frames, a.dtype, a.mode = B, F, HW, c_latent, c_pad, cfg_dup, mask_frames, _dt(x), mode
        self._call("fyc_unet_input", a)

    def cfg_ddim_step(self, pred: Tensor, latents: Tensor, coef: Tensor, *, B: int, F: int, HW: int, c_latent: int,
                      ld: int, cfg: bool, guidance: float, pred_type: int, clip_sample: bool,
                      pred_single: Optional[Tensor] = None, video_scale: float = 0.0, variance_noise: Optional[Tensor] = None,
                      sigma: float = 0.0, clipped_model_output: bool = False) -> None:
        a = L.CfgDdimArgs()
        if variance_noise is not None and variance_noise.numel() != latents.numel():
            raise ValueError("cfg_ddim_step: variance_noise must have the latents' shape")
        a.variance_noise, a.sigma, a.clipped_model_output = _f32(variance_noise, "variance_noise"), float(sigma), int(clipped_model_output)
        if pred_single is not None and pred_single.dtype != pred.dtype:
            raise TypeError("cfg_ddim_step: pred_single dtype differs from pred")
        a.pred_single, a.video_scale = _p(pred_single), video_scale
        a.pred, a.latents, a.coef = _p(pred), _f32(latents, "latents"), _f32(coef, "coef")
        a.B, a.F, a.HW, a.c_latent, a.ld, a.cfg, a.guidance = B, F, HW, c_latent, ld, int(cfg), guidance
        a.pred_type, a.clip_sample, a.dtype = pred_type, int(clip_sample), _dt(pred)
        self._call("fyc_cfg_ddim_step", a)

    def nchw_to_nhwc(self, z: Tensor, x: Tensor, *, N: int, C_: int, HW: int, c_pad: int, scale: float) -> None:
        self.ensure_init(z.device)
        a = L.NchwInArgs()
        a.z, a.x, a.N, a.C, a.HW, a.c_pad, a.scale, a.dtype = _f32(z, "z"), _p(x), N, C_, HW, c_pad, scale, _dt(x)
        self._call("fyc_nchw_to_nhwc", a)

    def nhwc_to_nchw(self, x: Tensor, y: Tensor, *, N: int, C_: int, HW: int, ld: int, mul: float = 1.0, add: float = 0.0,
                     lo: float = -3.0e38, hi: float = 3.0e38) -> None:
        a = L.NhwcOutArgs()
        a.x, a.y, a.N, a.C, a.HW, a.ld, a.mul, a.add, a.lo, a.hi, a.dtype = _p(x), _f32(y, "y"), N, C_, HW, ld, mul, add, lo, hi, _dt(x)
        self._call("fyc_nhwc_to_nchw", a)


impl = None  # type: Optional[HipOps]


def get() -> HipOps:
    """The active backend; created on first use.  Raises if libfyc_hip.so is missing."""
    global impl
    if impl is None:
        impl = HipOps()
    return impl
