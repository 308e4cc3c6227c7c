"""fyc_panel_linear against fyc_gemm (+ fyc_gn_apply_cs where the GroupNorm is taken along) on the K <= 640 projection shapes of
the UNet, cold operands.  usage (GPU box): python tools/panel_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from followyourclick_amd import ops
from followyourclick_amd.engine.weights import pack_panel_linear

T, DEV = torch.bfloat16, torch.device("cuda:0")


def timed(fn, n):
    fn(0); fn(1)
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
    for rows, C, rps in ((131072, 320, 4096), (32768, 640, 1024)):
        nb = 8
        w = (torch.randn(C, C, device=DEV) * C ** -0.5).to(T)
        ws, bias = pack_panel_linear(w), torch.randn(C, device=DEV) * 0.1
        sets = [(torch.randn(rows, C, device=DEV).to(T), torch.randn(rows, C, device=DEV).to(T), torch.empty(rows, C, dtype=T, device=DEV),
                 torch.empty(rows, C, dtype=T, device=DEV)) for _ in range(nb)]
        cs = torch.rand(rows // rps, C, 2, dtype=torch.float64, device=DEV) * rps
        cs[..., 1] += cs[..., 0] ** 2 / rps
        gam, bet = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        flops = 2.0 * rows * C * C

        def gemm_res(i):
            x, r, o, _ = sets[i % nb]
            h.gemm(x, w, o, M=rows, N=C, K=C, lda=C, ldw=C, ldo=C, bias=bias, residual=r, ldr=C)

        def panel_res(i):
            x, r, o, _ = sets[i % nb]
            h.panel_linear(x, o, wstream=ws, rows=rows, N=C, K=C, bias=bias, residual=r)

        def gemm_gn(i):
            x, r, o, t = sets[i % nb]
            h.gn_apply_cs(x, cs, gam, bet, t, rows=rows, C1=C, groups=32, rows_per_sample=rps, eps=1e-6, silu=False, cs_rows=rps)
            h.gemm(t, w, o, M=rows, N=C, K=C, lda=C, ldw=C, ldo=C, bias=bias)

        def panel_gn(i):
            x, r, o, _ = sets[i % nb]
            h.panel_linear(x, o, wstream=ws, rows=rows, N=C, K=C, bias=bias, gn_cs=cs, gn_gamma=gam, gn_beta=bet, gn_rows_per_sample=rps)

        res = {}
        for rnd in range(3):
            for name, fn in (("fyc_gemm + residual", gemm_res), ("fyc_panel_linear + residual", panel_res),
                             ("fyc_gn_apply_cs + fyc_gemm (proj_in)", gemm_gn), ("fyc_panel_linear with GroupNorm (proj_in)", panel_gn)):
                res.setdefault(name, []).append(timed(fn, 2 * nb))
        for name, v in res.items():
            us = min(v)
            print(f"rows={rows:7d} C={C}  {name:44s} {us:8.1f} us  {flops / us / 1e6:7.0f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
