"""diagnosis: fyc_gemm GEGLU / dual-K at M = 8192, N = 2560 / 320, K = 320 / 1600 (shapes the engine never issues) against torch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from followyourclick_amd import ops, _lib as L
from emu_ops import EmuOps
h = ops.get(); h.ensure_init(torch.device("cuda:0")); emu = EmuOps(acc=torch.float64)
T = torch.bfloat16
g = torch.Generator().manual_seed(0)
for M in (8192, 16384, 4096):
    C, hid = 320, 1280
    x = torch.randn(M, C, generator=g).to(T); w1 = (torch.randn(2 * hid, C, generator=g) * C ** -0.5).to(T); b1 = torch.randn(2 * hid, generator=g) * 0.1
    st = torch.empty(M, 2, device="cuda"); h.row_stats(x.cuda(), st, rows=M, C_=C)
    cs = w1.float().sum(1)
    for tile in (0, 5, 6, 1):
        hm = torch.full((M, hid), float("nan"), dtype=T, device="cuda")
        try:
            h.gemm(x.cuda(), w1.cuda(), hm, M=M, N=2 * hid, K=C, lda=C, ldw=C, ldo=hid, bias=b1.cuda(), epilogue=L.EPI_GEGLU, ln_colsum=cs.cuda(), ln_stats=st, tile=tile)
            torch.cuda.synchronize()
            n = 256
            he = torch.zeros(n, hid, dtype=T)
            emu.gemm(x[:n], w1, he, M=n, N=2 * hid, K=C, lda=C, ldw=C, ldo=hid, bias=b1, epilogue=1, ln_colsum=cs, ln_stats=st[:n].cpu())
            bad = (~torch.isfinite(hm.float())).sum().item()
            err = ((hm[:n].cpu().double() - he.double()).norm() / he.double().norm()).item()
            print(f"GEGLU M={M} tile={tile}: nonfinite={bad} rel(first {n} rows)={err:.3e}", flush=True)
        except Exception as e:
            print(f"GEGLU M={M} tile={tile}: {str(e)[:100]}")
    hmid = torch.randn(M, hid, generator=g).to(T); po = (torch.randn(C, C + hid, generator=g) * 0.03).to(T); res = torch.randn(M, C, generator=g).to(T)
    for tile in (0, 5, 6, 1):
        o = torch.full((M, C), float("nan"), dtype=T, device="cuda")
        try:
            h.gemm(x.cuda(), po.cuda(), o, M=M, N=C, K=C + hid, lda=C, ldw=C + hid, ldo=C, residual=res.cuda(), ldr=C, a2=hmid.cuda(), k_split=C, lda2=hid, tile=tile)
            torch.cuda.synchronize()
            n = 256
            oe = torch.zeros(n, C, dtype=T)
            emu.gemm(x[:n], po, oe, M=n, N=C, K=C + hid, lda=C, ldw=C + hid, ldo=C, residual=res[:n], ldr=C, a2=hmid[:n], k_split=C, lda2=hid)
            bad = (~torch.isfinite(o.float())).sum().item()
            err = ((o[:n].cpu().double() - oe.double()).norm() / oe.double().norm()).item()
            print(f"dualK M={M} tile={tile}: nonfinite={bad} rel={err:.3e}", flush=True)
        except Exception as e:
            print(f"dualK M={M} tile={tile}: {str(e)[:100]}")
