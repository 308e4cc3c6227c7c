"""diagnostic (GPU): which rows of the f16 fused attention come out non-finite on the late-spike / early-spike / ramp data of
tests/test_kernels_f16_gpu.py::test_attention_score_offsets_and_late_spikes_f16, and which ingredient of the data it takes"""
import math
import sys

import torch

sys.path.insert(0, ".")
from followyourclick_amd import ops  # noqa: E402

hip = ops.get()
hip.ensure_init(torch.device("cuda:0"))


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def data(T, d, late, early, ramp, q1, q0, early_scale=7.0, late_scale=5.0):
    B, H, n = 1, 4, 448
    q, k = rnd((B * H, n, d), T, 1), rnd((B * H, n, d), T, 2)
    qf, kf = q.float(), k.float()
    if q0:
        qf[..., 0] = 1.0
        kf[..., 0] = 0.0
    if late:
        kf[:, 440] = kf[:, 440] + qf[:, 17] * late_scale
    if early:
        kf[:, 2] = kf[:, 2] + qf[:, 90] * early_scale
    if ramp:
        kf[..., 1] = torch.arange(n).float()[None, :] * 0.004
    if q1:
        qf[..., 1] = 2.0
    return qf.to(T), kf.to(T), rnd((B * H, d, n), T, 3)


def run(T, d, qt=0, **kw):
    q, k, vt = data(T, d, **kw)
    n, H = 448, 4
    o = torch.zeros(n, H * d, dtype=T, device="cuda")
    hip.set_tuning(3, qt)
    try:
        hip.attention(q.cuda(), k.cuda(), vt.cuda(), o, batch=1, heads=H, n_q=n, n_k=n, d=d, ldo=H * d, ldvt=n, scale=d ** -0.5)
        torch.cuda.synchronize()
    finally:
        hip.set_tuning(3, 0)
    o = o.float().cpu().reshape(n, H, d)
    bad = (~torch.isfinite(o)).any(dim=2)
    idx = bad.nonzero().tolist()
    S = torch.einsum("hqd,hkd->hqk", q.float(), k.float()) * d ** -0.5 * math.log2(math.e)
    info = []
    for qi, h in idx[:12]:
        row = S[h, qi]
        info.append((qi, h, "nan" if torch.isnan(o[qi, h]).any() else "inf", round(row.max().item(), 1), int(row.argmax()), round(row.topk(2).values[1].item(), 1)))
    return len(idx), info


import os
SHORT = os.environ.get("FYC_DIAG_SHORT") == "1"
for T in (torch.float16,) if SHORT else (torch.float16, torch.bfloat16):
    for d in (40, 160):
        for name, kw in (("all", dict(late=1, early=1, ramp=1, q1=1, q0=1)), ("late only", dict(late=1, early=0, ramp=0, q1=0, q0=0)),
                         ("early only", dict(late=0, early=1, ramp=0, q1=0, q0=0)), ("ramp+q1", dict(late=0, early=0, ramp=1, q1=1, q0=0)),
                         ("plain", dict(late=0, early=0, ramp=0, q1=0, q0=0)), ("early x3", dict(late=0, early=1, ramp=0, q1=0, q0=0, early_scale=3.0)),
                         ("late x2", dict(late=1, early=0, ramp=0, q1=0, q0=0, late_scale=2.0)))[:2 if SHORT else 9]:
            nbad, info = run(T, d, **kw)
            print(f"{str(T)[6:]:9s} d{d:3d} {name:10s}: {nbad:3d} bad (query, head) rows", info[:10], flush=True)
        for qt in () if SHORT else (2, 3, 4):
            if d == 160 and qt > 2:
                continue
            nbad, info = run(T, d, qt=qt, late=1, early=1, ramp=1, q1=1, q0=1)
            print(f"{str(T)[6:]:9s} d{d:3d} all qt={qt}: {nbad} bad", flush=True)
