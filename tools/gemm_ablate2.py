"""Epilogue/main-loop ablation of the GEMM on small-K shapes (needs the experimental build: FYC_LIB_PATH=tools/exp/libfyc_exp.so).
tuning key 4 bits: 1 no DMA, 2 no MFMA, 8 no stores, 16 no GELU, 32 no residual read."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from followyourclick_amd import ops
h = ops.get(); dev = torch.device("cuda:0"); h.ensure_init(dev)
T = torch.bfloat16
CASES = [("geglu L64", 131072, 2560, 320, 1, 0, 6), ("N320K320+res", 131072, 320, 320, 0, 1, 5), ("N320K320+res c2", 131072, 320, 320, 0, 1, 2),
         ("N960K320", 131072, 960, 320, 0, 0, 5), ("geglu L32", 32768, 5120, 640, 1, 0, 6), ("big conv-like", 32768, 640, 5760, 0, 1, 5)]
for name, M, N, K, epi, res, cfg in CASES:
    a = torch.randn(M, K, device=dev).to(T); w = (torch.randn(N, K, device=dev) / K ** 0.5).to(T)
    ocols = N // 2 if epi else N
    out = torch.empty(M, ocols, dtype=T, device=dev); bias = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev).to(T) if res else None
    h.set_tuning(1, cfg); h.set_tuning(2, 2)
    row = []
    for dbg in (0, 8, 16, 32, 8 | 16 | 32, 2, 1, 3, 3 | 8 | 16 | 32):
        h.set_tuning(4, dbg)
        kw = dict(M=M, N=N, K=K, lda=K, ldw=K, ldo=ocols, ldr=N, epilogue=epi, bias=bias, residual=r)
        for _ in range(2): h.gemm(a, w, out, **kw)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(6): h.gemm(a, w, out, **kw)
        e.record(); torch.cuda.synchronize()
        row.append(s.elapsed_time(e) / 6 * 1e3)
    h.set_tuning(4, 0)
    print(f"{name:18s} cfg{cfg}: full {row[0]:.0f}us | nostore {row[1]:.0f} | nogelu {row[2]:.0f} | nores {row[3]:.0f} | bare-epi {row[4]:.0f} | noMFMA {row[5]:.0f} | noDMA {row[6]:.0f} | noloop {row[7]:.0f} | nothing {row[8]:.0f}")
