"""Where do the small-K linears lose time inside the pipeline?  Times the engine's K=320 / 640 GEMM calls as the UNet issues
them (LayerNorm fold, per-frame row bias, fused output statistics, residual) against the bare GEMM, with the operands either
re-used every launch (Infinity-Cache hot, what tools/gemm_bench.py measures) or rotated over enough buffers to come from HBM.
usage (GPU box): python tools/gemm_probe.py > gpurun_out/gemm_probe.txt"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from followyourclick_amd import ops

T = torch.bfloat16
DEV = torch.device("cuda:0")


def run(h, M, N, K, *, epi=0, res=False, ln=False, rowbias=False, stats=False, nb=1, tile=0, reps=6, conv=None, k2=0, segs=0):
    ocols = N // 2 if epi == 1 else N
    sets = []
    for _ in range(nb):
        a = torch.randn(M, conv[2] if conv else K - k2, device=DEV).to(T)
        out = torch.empty(M, ocols, dtype=T, device=DEV)
        r = torch.randn(M, N, device=DEV).to(T) if res else None
        sets.append((a, out, r))
    w = (torch.randn(N, K, device=DEV) / K ** 0.5).to(T)
    bias = torch.randn(N, device=DEV)
    kw = dict(M=M, N=N, K=K, lda=K - k2, ldw=K, ldo=ocols, ldr=N, epilogue=epi, bias=bias, tile=tile)
    if conv:
        Hh, Ww, Cin = conv
        kw.update(lda=Cin, mode=1, conv=dict(Hout=Hh, Wout=Ww, Hin=Hh, Win=Ww, Cin=Cin, stride=1))
    if k2:      # merged FF2 | proj_out: A = [hidden (K - k2) | tokens (k2)]
        kw.update(a2=torch.randn(M, k2, device=DEV).to(T), k_split=K - k2, lda2=k2)
    if os.environ.get("PROBE_LDA0") and not conv:      # every A row aliases row 0: the A operand is cache-resident (what does the K loop cost without its HBM latency?)
        kw["lda"] = 0
        if k2:
            kw["lda2"] = 0
    if ln:
        kw["ln_stats"] = torch.rand(M, 2, device=DEV) + 0.5
        kw["ln_colsum"] = torch.randn(N, device=DEV)
    if rowbias:
        kw["rowbias"] = torch.randn(32, N, device=DEV)
        kw["rows_per_batch"] = M // 32
    if epi == 2:      # head-split epilogue: `segs` segments of C = N / segs columns (q | k | V^T, or q alone), 8 heads, 32 frames of M / 32 tokens
        Cc, tok = N // segs, M // 32
        d = Cc // 8
        outs = [torch.empty(32, 8, tok, d, dtype=T, device=DEV) for _ in range(min(segs, 2))] + ([torch.empty(32, 8, d, tok, dtype=T, device=DEV)] if segs == 3 else [])
        kw.update(heads=dict(seg_cols=Cc, heads=8, tokens=tok, outs=outs, transposed=[0, 0, 1][:segs], ld=[0, 0, tok][:segs]))
        for i in range(len(sets)):
            sets[i] = (sets[i][0], None, None)
    if stats:
        n, tr, sl = h.gemm_stat_layout(T, M=M, N=N, K=K, cs_rows=M // 32, mode=1 if conv else 0, tile=tile)
        kw["chan_parts"] = torch.empty(n * sl * N * 2, device=DEV)
        kw["cs_rows"] = M // 32
    for i in range(2 * nb):
        a, out, r = sets[i % nb]
        h.gemm(a, w, out, residual=r, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(reps * nb):
        a, out, r = sets[i % nb]
        h.gemm(a, w, out, residual=r, **kw)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / (reps * nb) * 1e3
    return us, 2.0 * M * N * K / us / 1e6


def sweep(h):
    """cold operands + the epilogue features the UNet uses, per tile config: input of gemm.hip::choose"""
    cases = [
        ("FF1 GEGLU L0 ln", 131072, 2560, 320, dict(epi=1, ln=True)), ("FF1 GEGLU L1 ln", 32768, 5120, 640, dict(epi=1, ln=True)),
        ("FF1 GEGLU L2 ln", 8192, 10240, 1280, dict(epi=1, ln=True)),
        ("tQKV L0 ln rb", 131072, 960, 320, dict(ln=True, rowbias=True)), ("tQKV L1 ln rb", 32768, 1920, 640, dict(ln=True, rowbias=True)),
        ("tQKV L2 ln rb", 8192, 3840, 1280, dict(ln=True, rowbias=True)),
        ("sQKV L0 heads ln", 131072, 960, 320, dict(epi=2, ln=True, segs=3)), ("sQKV L1 heads ln", 32768, 1920, 640, dict(epi=2, ln=True, segs=3)),
        ("sQKV L2 heads ln", 8192, 3840, 1280, dict(epi=2, ln=True, segs=3)),
        ("q2 L0 heads ln", 131072, 320, 320, dict(epi=2, ln=True, segs=1)), ("q2 L1 heads ln", 32768, 640, 640, dict(epi=2, ln=True, segs=1)),
        ("q2 L2 heads ln", 8192, 1280, 1280, dict(epi=2, ln=True, segs=1)),
        ("to_out L0 res", 131072, 320, 320, dict(res=True)), ("to_out L1 res", 32768, 640, 640, dict(res=True)),
        ("to_out L2 res", 8192, 1280, 1280, dict(res=True)), ("to_out L3 res", 2048, 1280, 1280, dict(res=True)),
        ("proj_in L0", 131072, 320, 320, dict()), ("proj_in L1", 32768, 640, 640, dict()),
        ("FF2|proj L0 res st", 131072, 320, 1600, dict(res=True, stats=True, k2=320)), ("FF2|proj L1 res st", 32768, 640, 3200, dict(res=True, stats=True, k2=640)),
        ("FF2|proj L2 res st", 8192, 1280, 6400, dict(res=True, stats=True, k2=1280)),
        ("conv L0 res st", 131072, 320, 2880, dict(res=True, stats=True, conv=(64, 64, 320))),
        ("conv L0 rb st", 131072, 320, 2880, dict(rowbias=True, stats=True, conv=(64, 64, 320))),
        ("conv L1 res st", 32768, 640, 5760, dict(res=True, stats=True, conv=(32, 32, 640))),
        ("conv L2 res st", 8192, 1280, 11520, dict(res=True, stats=True, conv=(16, 16, 1280))),
        ("conv up L0 st", 131072, 320, 5760, dict(rowbias=True, stats=True, conv=(64, 64, 640))),
        ("conv up L1 st", 32768, 640, 11520, dict(rowbias=True, stats=True, conv=(32, 32, 1280))),
        ("conv up L2 st", 8192, 1280, 23040, dict(rowbias=True, stats=True, conv=(16, 16, 2560))),
    ]
    if os.environ.get("PROBE_SMALL"):      # the 8x8 / 16x16 latent levels: latency-bound K chains, tile ids with the ring depth in bits 8+
        cases = [
            ("to_out L3 res", 2048, 1280, 1280, dict(res=True)), ("proj_in L3", 2048, 1280, 1280, dict()),
            ("tQKV L3 ln rb", 2048, 3840, 1280, dict(ln=True, rowbias=True)), ("FF1 GEGLU L3 ln", 2048, 10240, 1280, dict(epi=1, ln=True)),
            ("FF2|proj L3 res st", 2048, 1280, 6400, dict(res=True, stats=True, k2=1280)), ("shortcut L3", 2048, 1280, 2560, dict()),
            ("to_out L2 res", 8192, 1280, 1280, dict(res=True)), ("proj_in L2", 8192, 1280, 1280, dict()),
            ("tQKV L2 ln rb", 8192, 3840, 1280, dict(ln=True, rowbias=True)),
        ]
    cfgs = [int(c) for c in os.environ.get("PROBE_CFGS", "0,1,3,5,6,7,8").split(",")]
    print("cold operands (8 rotating buffer sets), us per launch; tile 0 = the library's own choice")
    print("case".ljust(22), "shape".ljust(24), " ".join(f"c{c}".rjust(7) for c in cfgs))
    for name, M, N, K, kw in cases:
        row = []
        for c in cfgs:
            try:
                us, _ = run(h, M, N, K, nb=8, tile=c, reps=4, **kw)
                row.append(f"{us:7.1f}")
            except Exception as e:
                row.append("      -")
            torch.cuda.empty_cache()
        print(name.ljust(22), f"{M}x{N}x{K}".ljust(24), " ".join(row), flush=True)


def main():
    h = ops.get()
    h.ensure_init(DEV)
    for kv in filter(None, os.environ.get("PROBE_TUNING", "").split(",")):      # e.g. PROBE_TUNING=5=3,0=1
        k, v = kv.split("=")
        h.set_tuning(int(k), int(v))
    if os.environ.get("PROBE_SWEEP"):
        return sweep(h)
    cases = [
        ("temporal QKV L0", 131072, 960, 320, dict(), [dict(), dict(ln=True), dict(ln=True, rowbias=True)]),
        ("FF1 GEGLU L0", 131072, 2560, 320, dict(epi=1), [dict(), dict(ln=True)]),
        ("to_out L0", 131072, 320, 320, dict(res=True), [dict(), dict(stats=True)]),
        ("attn2 Q L0", 131072, 320, 320, dict(), [dict(), dict(ln=True)]),
        ("temporal QKV L1", 32768, 1920, 640, dict(), [dict(), dict(ln=True, rowbias=True)]),
        ("FF1 GEGLU L1", 32768, 5120, 640, dict(epi=1), [dict(), dict(ln=True)]),
        ("to_out L1", 32768, 640, 640, dict(res=True), [dict()]),
        ("FF1 GEGLU L2", 8192, 10240, 1280, dict(epi=1), [dict(), dict(ln=True)]),
    ]
    tiles = [int(t) for t in os.environ.get("PROBE_TILES", "0").split(",")]
    for name, M, N, K, base, variants in cases:
        for v in variants:
            for tile in tiles:
                row = []
                for nb in (1, 8):
                    us, tf = run(h, M, N, K, nb=nb, tile=tile, **base, **v)
                    row.append(f"{'hot' if nb == 1 else 'cold'} {us:7.1f} us {tf:6.0f} TF")
                tag = ",".join(k for k in v) or "bare"
                print(f"{name:18s} M={M} N={N} K={K} tile={tile} [{tag:12s}]  " + "   ".join(row), flush=True)
                torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
