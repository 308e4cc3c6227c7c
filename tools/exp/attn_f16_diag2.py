"""diagnostic 2 (GPU): where the f16 fused attention turns non-finite as a function of the spike's height and position"""
import math
import sys

import torch

sys.path.insert(0, ".")
from followyourclick_amd import ops  # noqa: E402

hip = ops.get()
hip.ensure_init(torch.device("cuda:0"))
T = torch.float16


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def run(d, key, scale_, qsel=17):
    B, H, n = 1, 4, 448
    q, k = rnd((B * H, n, d), T, 1), rnd((B * H, n, d), T, 2)
    qf, kf = q.float(), k.float()
    kf[:, key] = kf[:, key] + qf[:, qsel] * scale_
    q, k, vt = qf.to(T), kf.to(T), rnd((B * H, d, n), T, 3)
    o = torch.zeros(n, H * d, dtype=T, device="cuda")
    hip.attention(q.cuda(), k.cuda(), vt.cuda(), o, batch=1, heads=H, n_q=n, n_k=n, d=d, ldo=H * d, ldvt=n, scale=d ** -0.5)
    torch.cuda.synchronize()
    o = o.float().cpu().reshape(n, H, d)
    S = torch.einsum("hqd,hkd->hqk", q.float(), k.float()) * d ** -0.5
    ref = torch.einsum("hqk,hdk->qhd", S.softmax(-1), vt.float())
    bad = (~torch.isfinite(o)).any(dim=2)
    S2 = S * math.log2(math.e)
    rows = []
    for h in range(H):
        row = S2[h, qsel]
        top = row.topk(2).values
        prev = row[:key - key % 32].max().item() if key >= 32 else float("nan")
        rows.append(f"h{h}: max {top[0]:.1f} 2nd {top[1]:.1f} prevblocks {prev:.1f} {'BAD' if bad[qsel, h] else 'ok'}")
    err = ((o - ref)[~bad].norm() / ref[~bad].norm()).item()
    return int(bad.sum()), rows, err


for d in (40, 160):
    for key in (440, 447, 416, 200, 100, 40, 33, 31, 2):
        for sc in (1.0, 2.0, 3.0, 5.0):
            nb, rows, err = run(d, key, sc * (40.0 / d))
            print(f"d{d} spike at key {key:3d} x{sc}: {nb:3d} bad rows; finite rows rel err {err:.2e}; q17: " + " | ".join(rows), flush=True)
