"""fyc_ff_block (one kernel) against the three launches it replaces (fyc_row_stats, fyc_gemm GEGLU with the folded LayerNorm,
fyc_gemm dual-K + residual + statistics), cold operands (rotating buffer sets), both instruction schedules of the kernel.
usage (GPU box): python tools/ff_probe.py > gpurun_out/ff_probe.txt"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from followyourclick_amd import _lib as L
from followyourclick_amd import ops
from followyourclick_amd.engine.weights import Packed, pack_ff_block

T = torch.bfloat16
DEV = torch.device("cuda:0")
C, HID = 320, 1280


def timed(fn, n):
    fn(0)
    fn(1)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main():
    h = ops.get()
    h.ensure_init(DEV)
    w1 = (torch.randn(2 * HID, C, device=DEV) * C ** -0.5).to(T)
    ff = Packed(w1=w1, b1=torch.randn(2 * HID, device=DEV) * 0.1, cs1=w1.float().sum(dim=1).contiguous(),
                po_w=(torch.randn(C, C + HID, device=DEV) * (C + HID) ** -0.5).to(T), po_b=torch.randn(C, device=DEV) * 0.1)
    ws = pack_ff_block(ff)
    for rows in (131072, 32768):
        nb = 8 if rows >= 32768 else 16
        sets = [(torch.randn(rows, C, device=DEV).to(T), torch.randn(rows, C, device=DEV).to(T), torch.empty(rows, C, dtype=T, device=DEV),
                 torch.empty(rows, HID, dtype=T, device=DEV)) for _ in range(nb)]
        st = torch.empty(rows, 2, dtype=torch.float32, device=DEV)
        parts = torch.empty(rows // 128 * C * 2, dtype=torch.float32, device=DEV)
        cs_rows = 4096 if rows % 4096 == 0 else 128
        n, tr, sl = h.gemm_stat_layout(T, M=rows, N=C, K=C + HID, cs_rows=cs_rows)
        gparts = torch.empty(n * sl * C * 2, dtype=torch.float32, device=DEV)
        flops = 2.0 * rows * (C * 2 * HID + (C + HID) * C)

        def fused(i):
            x, r, o, _ = sets[i % nb]
            h.ff_block(x, r, o, wstream=ws, b_out=ff.po_b, rows=rows, C_=C, hidden=HID, chan_parts=parts, cs_rows=cs_rows)

        def unfused(i):
            x, r, o, hm = sets[i % nb]
            h.row_stats(x, st, rows=rows, C_=C)
            h.gemm(x, ff.w1, hm, M=rows, N=2 * HID, K=C, lda=C, ldw=C, ldo=HID, bias=ff.b1, epilogue=L.EPI_GEGLU, ln_colsum=ff.cs1, ln_stats=st)
            h.gemm(x, ff.po_w, o, M=rows, N=C, K=C + HID, lda=C, ldw=C + HID, ldo=C, bias=ff.po_b, residual=r, ldr=C, a2=hm, k_split=C, lda2=HID,
                   chan_parts=gparts, cs_rows=cs_rows)

        res = {}
        for rnd in range(3):                    # interleaved rounds: variants see the same clocks
            for name, fn in (("unfused (3 launches)", unfused), ("ff_block", fused)):     # (the round-3 scheduling bits behind tuning keys 8 / 9 are gone: those keys belong to the GEMM variants now)
                res.setdefault(name, []).append(timed(fn, 2 * nb))
        for name, v in res.items():
            us = min(v)
            print(f"rows={rows:7d}  {name:40s} {us:8.1f} us (min of {['%.1f' % t for t in v]})  {flops / us / 1e6:7.0f} TFLOP/s", flush=True)
    return h, ws, ff


if __name__ == "__main__":
    h, ws, ff = main()

if hasattr(h.lib, "fyc_ff_timing"):                           # FF_TIMING build: phase timestamps (shader clock) of wave 0 of every workgroup
    import ctypes
    import numpy as np
    nb, rows = 1, 131072
    x, r, o = (torch.randn(rows, C, device=DEV).to(T) for _ in range(3))
    h.ff_block(x, r, o, wstream=ws, b_out=ff.po_b, rows=rows, C_=C, hidden=HID); torch.cuda.synchronize()
    n = 1024 * 16
    buf = (ctypes.c_ulonglong * n)()
    assert h.lib.fyc_ff_timing(buf, n) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16).astype(np.int64)
    names = ["start", "x requested, halves 0/1 requested, LN statistics", "projection done (10 halves)", "tokens normalised", "chunk 20: before barrier A", "after barrier A",
             "half A done (FF1 k 0-13 + gate of chunk 19)", "after barrier B", "FF1 k 14-19 done", "FF2 of chunk 19 done", "before chunk 39", "last FF2 done", "residual landed", "end"]
    for sel, lab in ((slice(0, 256), "workgroups 0..255"), (slice(512, 768), "workgroups 512..767")):
        d = t[sel]
        print(lab)
        for k in range(1, 14):
            print(f"   {names[k]:52s} +{np.median(d[:, k] - d[:, 0]):9.0f} cycles   (step {np.median(d[:, k] - d[:, k - 1]):7.0f})")
