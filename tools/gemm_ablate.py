"""Ablation of the GEMM main loop (tuning key 4: 1 = no DMA after prologue, 2 = no MFMA, 4 = no barrier)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from followyourclick_amd import ops
h = ops.get(); dev = torch.device("cuda:0"); h.ensure_init(dev)
T = torch.bfloat16
for (M, N, K) in [(32768, 640, 5760), (131072, 320, 320), (8192, 10240, 1280), (32768, 5120, 5120)]:
    a = torch.randn(M, K, device=dev).to(T); w = (torch.randn(N, K, device=dev) / K ** 0.5).to(T); out = torch.empty(M, N, dtype=T, device=dev)
    for cfg, ns in [(1, 2), (2, 2), (3, 2)]:
        h.set_tuning(1, cfg); h.set_tuning(2, ns)
        res = []
        for dbg in (0, 1, 2, 3, 4, 5):
            h.set_tuning(4, dbg)
            for _ in range(2): h.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldw=K, ldo=N)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5): h.gemm(a, w, out, M=M, N=N, K=K, lda=K, ldw=K, ldo=N)
            e.record(); torch.cuda.synchronize()
            res.append(s.elapsed_time(e) / 5 * 1e3)
        h.set_tuning(4, 0)
        fl = 2.0 * M * N * K
        print(f"M={M} N={N} K={K} cfg={cfg} ns={ns}: full {res[0]:.0f}us ({fl/res[0]/1e6:.0f} TF) | noDMA {res[1]:.0f} ({fl/res[1]/1e6:.0f} TF) | noMFMA {res[2]:.0f} | neither {res[3]:.0f} | nobarrier {res[4]:.0f} | nobarrier+noDMA {res[5]:.0f} ({fl/res[5]/1e6:.0f} TF)")
