"""Per-kernel timing with HIP events + algorithmic FLOP / byte accounting (bench.py's roofline leg).

`TimedOps` wraps the ops backend: every C-ABI call is bracketed by two events recorded on the stream the
kernel is launched on (torch's current stream == the stream handed to libfyc_hip.so), so after a
synchronise each launch has its own duration.  FLOPs are *algorithmic* (2*M*N*K of the contraction the
reference performs), never the padded work the kernel issues.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Dict

import torch


def _esize(t) -> int:
    return t.element_size()


class TimedOps:
    def __init__(self, inner):
        self.inner = inner
        self.name = inner.name
        self.records = []  # (family, start_evt, end_evt, flops, bytes)
        self.enabled = True

    # pass-through for non-kernel attributes
    def ensure_init(self, device):
        return self.inner.ensure_init(device)

    def set_tuning(self, key, value):
        return self.inner.set_tuning(key, value)

    def device_caps(self):
        return self.inner.device_caps()

    def _timed(self, family: str, flops: float, nbytes: float, fn, *a, _key=None, **kw):
        if not self.enabled:
            return fn(*a, **kw)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn(*a, **kw)
        e.record()
        self.records.append((family, s, e, flops, nbytes, _key))
        return r

    def gemm(self, a, w, out, **kw):
        M, N, K, batch = kw["M"], kw["N"], kw["K"], kw.get("batch", 1)
        flops = 2.0 * M * N * K * batch
        es = _esize(a)
        mode = kw.get("mode", 0)
        if mode == 0:
            in_bytes = M * K * es * batch
            fam = "gemm"
        else:
            c = kw["conv"]
            frames = M // (c["Hout"] * c["Wout"])
            in_bytes = frames * c["Hin"] * c["Win"] * c["Cin"] * es
            fam = "conv3x3"
        out_cols = N // 2 if kw.get("epilogue", 0) == 1 else N
        nbytes = in_bytes + N * K * es * batch + M * out_cols * es * batch
        if kw.get("residual") is not None:
            nbytes += M * N * es
        key = f"{fam} M={M} N={N} K={K} b={batch} epi={kw.get('epilogue', 0)} res={int(kw.get('residual') is not None)}"
        return self._timed(fam, flops, nbytes, self.inner.gemm, a, w, out, _key=key, **kw)

    def attention(self, q, k, vt, o, **kw):
        B, H, nq, nk, d = kw["batch"], kw["heads"], kw["n_q"], kw["n_k"], kw["d"]
        flops = 4.0 * B * H * nq * nk * d
        kvb = (B + kw.get("kv_batch_div", 1) - 1) // kw.get("kv_batch_div", 1)
        nbytes = (2 * B * H * nq * d + 2 * kvb * H * nk * d) * 2
        fam = "attn_self" if nq == nk else "attn_cross"
        return self._timed(fam, flops, nbytes, self.inner.attention, q, k, vt, o, _key=f"{fam} B={B} H={H} nq={nq} nk={nk} d={d}", **kw)

    def temporal_attention(self, qkv, o, **kw):
        n = kw["clips"] * kw["frames"] * kw["pixels"] * kw["heads"] * kw["d"]
        flops = 4.0 * kw["clips"] * kw["pixels"] * kw["heads"] * kw["frames"] ** 2 * kw["d"]
        return self._timed("attn_temporal", flops, 4.0 * n * _esize(qkv), self.inner.temporal_attention, qkv, o, **kw)

    def gn_stats(self, x, stats, **kw):
        return self._timed("gn_stats", 0.0, kw["rows"] * kw["C_"] * _esize(x), self.inner.gn_stats, x, stats, **kw)

    def gn_apply(self, x, stats, g, b, y, **kw):
        return self._timed("gn_apply", 0.0, 2.0 * kw["rows"] * kw["C_"] * _esize(x), self.inner.gn_apply, x, stats, g, b, y, **kw)

    def gn_apply_cs(self, x1, cs1, g, b, y, **kw):
        return self._timed("gn_apply", 0.0, 2.0 * kw["rows"] * (kw["C1"] + kw.get("C2", 0)) * _esize(x1), self.inner.gn_apply_cs, x1, cs1, g, b, y, **kw)

    def gemm_row_parts(self, dtype, **kw):      # host-side queries, no kernel
        return self.inner.gemm_row_parts(dtype, **kw)

    def gemm_stat_layout(self, dtype, **kw):
        return self.inner.gemm_stat_layout(dtype, **kw)

    def gemm_split_bytes(self, dtype, **kw):
        return self.inner.gemm_split_bytes(dtype, **kw)

    def temporal_block_supported(self, dtype, **kw):
        return self.inner.temporal_block_supported(dtype, **kw)

    def temporal_block(self, x, out, **kw):
        rows, C = kw["clips"] * kw["frames"] * kw["pixels"], kw["heads"] * kw["d"]
        return self._timed("temporal_block", 2.0 * rows * C * 4 * C + 4.0 * rows * kw["frames"] * C, 2.0 * rows * C * _esize(x),
                           self.inner.temporal_block, x, out, **kw)

    def ff_block_supported(self, dtype, **kw):
        return self.inner.ff_block_supported(dtype, **kw)

    def ff_block(self, x, residual, out, **kw):
        rows, C, hid = kw["rows"], kw["C_"], kw["hidden"]
        flops = 2.0 * rows * (C * 2 * hid + (C + hid) * C)
        nbytes = (2 + (residual is not None)) * rows * C * _esize(x) + (C * 2 * hid + (C + hid) * C) * _esize(x)
        return self._timed("ff_block", flops, nbytes, self.inner.ff_block, x, residual, out, **kw)

    def panel_linear_supported(self, dtype, **kw):
        return self.inner.panel_linear_supported(dtype, **kw)

    def panel_linear(self, x, out, **kw):
        rows, N, K = kw["rows"], kw["N"], kw["K"]
        nbytes = (rows * K + rows * N * (2 if kw.get("residual") is not None else 1) + N * K) * _esize(x)
        key = f"panel_linear M={rows} N={N} K={K} gn={int(kw.get('gn_cs') is not None)} res={int(kw.get('residual') is not None)}"
        return self._timed("panel_linear", 2.0 * rows * N * K, nbytes, self.inner.panel_linear, x, out, _key=key, **kw)

    def chan_stats_reduce(self, parts, cs, **kw):
        return self._timed("gn_stats", 0.0, 0.0, self.inner.chan_stats_reduce, parts, cs, **kw)

    def layernorm(self, x, g, b, y, **kw):
        return self._timed("layernorm", 0.0, 2.0 * kw["rows"] * kw["C_"] * _esize(x), self.inner.layernorm, x, g, b, y, **kw)

    def softmax_rows(self, x, **kw):
        return self._timed("softmax", 0.0, 2.0 * kw["rows"] * kw["cols"] * _esize(x), self.inner.softmax_rows, x, **kw)

    def concat_channels(self, a, b, y, **kw):
        return self._timed("concat", 0.0, 2.0 * kw["rows"] * (kw["c1"] + kw["c2"]) * _esize(a), self.inner.concat_channels, a, b, y, **kw)

    def __getattr__(self, name):  # remaining small ops: timed under their own name, bytes unknown
        fn = getattr(self.inner, name)
        if not callable(fn):
            return fn

        def wrapper(*a, **kw):
            return self._timed(name, 0.0, 0.0, fn, *a, **kw)
        return wrapper

    def reset(self):
        self.records = []

    def by_shape(self):
        """per distinct call signature: (key, launches, total ms, TFLOP/s) sorted by time"""
        out = defaultdict(lambda: [0, 0.0, 0.0])
        for fam, s, e, fl, nb, key in self.records:
            d = out[key or fam]
            d[0] += 1
            d[1] += s.elapsed_time(e)
            d[2] += fl
        rows = [(k, v[0], v[1], (v[2] / (v[1] * 1e-3) / 1e12) if v[1] > 0 and v[2] > 0 else 0.0) for k, v in out.items()]
        return sorted(rows, key=lambda r: -r[2])

    def summary(self) -> Dict[str, dict]:
        """call after torch.cuda.synchronize()"""
        out = defaultdict(lambda: dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
        for fam, s, e, fl, nb, _ in self.records:
            d = out[fam]
            d["launches"] += 1
            d["ms"] += s.elapsed_time(e)
            d["flops"] += fl
            d["bytes"] += nb
        return dict(out)
