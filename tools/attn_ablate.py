"""Attention kernel ablation (tuning key 5: 1 = no DMA after first tile, 2 = no softmax VALU, 4 = no MFMA)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from followyourclick_amd import ops
h = ops.get(); dev = torch.device("cuda:0"); h.ensure_init(dev)
T = torch.bfloat16
for (B, H, n, d) in [(32, 8, 4096, 40), (32, 8, 1024, 80), (32, 8, 256, 160)]:
    q = torch.randn(B * H, n, d, device=dev).to(T); k = torch.randn(B * H, n, d, device=dev).to(T); vt = torch.randn(B * H, d, n, device=dev).to(T)
    o = torch.empty(B * n, H * d, dtype=T, device=dev)
    kw = dict(batch=B, heads=H, n_q=n, n_k=n, d=d, ldo=H * d, ldvt=n, scale=d ** -0.5)
    res = []
    for dbg in (0, 1, 2, 4, 3, 5, 6, 7):
        h.set_tuning(5, dbg)
        for _ in range(2): h.attention(q, k, vt, o, **kw)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3): h.attention(q, k, vt, o, **kw)
        e.record(); torch.cuda.synchronize()
        res.append(s.elapsed_time(e) / 3 * 1e3)
    h.set_tuning(5, 0)
    fl = 4.0 * B * H * n * n * d
    names = ["full", "noDMA", "noSM", "noMFMA", "noDMA+noSM", "noDMA+noMFMA", "noSM+noMFMA", "none"]
    print(f"B={B} H={H} n={n} d={d}: " + " | ".join(f"{nm} {t:.0f}us" for nm, t in zip(names, res)) + f"  (full = {fl/res[0]/1e6:.0f} TF)")
