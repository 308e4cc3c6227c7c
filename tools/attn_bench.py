"""Attention micro-benchmark on the UNet3D cfg2 self-attention shapes: QT=2 vs QT=4 (tuning key 3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from followyourclick_amd import ops
h = ops.get(); dev = torch.device("cuda:0"); h.ensure_init(dev)
T = torch.bfloat16
for (B, H, n, nk, d) in [(32, 8, 4096, 4096, 40), (32, 8, 1024, 1024, 80), (32, 8, 256, 256, 160), (32, 8, 4096, 77, 40)]:
    q = torch.randn(B * H, n, d, device=dev).to(T); k = torch.randn(B * H, nk, d, device=dev).to(T)
    ld = (nk + 7) // 8 * 8
    vt = torch.zeros(B * H, d, ld, device=dev, dtype=T); vt[..., :nk] = torch.randn(B * H, d, nk, device=dev).to(T)
    o = torch.empty(B * n, H * d, dtype=T, device=dev)
    kw = dict(batch=B, heads=H, n_q=n, n_k=nk, d=d, ldo=H * d, ldvt=ld, scale=d ** -0.5)
    res = []
    for var in (2, 3, 4):
        h.set_tuning(3, var)
        for _ in range(2): h.attention(q, k, vt, o, **kw)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): h.attention(q, k, vt, o, **kw)
        e.record(); torch.cuda.synchronize()
        res.append(s.elapsed_time(e) / 5 * 1e3)
    h.set_tuning(3, 0)
    fl = 4.0 * B * H * n * nk * d
    print(f"B={B} H={H} nq={n} nk={nk} d={d}: " + " | ".join(f"QT{v} {r:.0f}us ({fl/r/1e6:.0f} TF)" for v, r in zip((2, 3, 4), res)))
