"""GEMM / conv micro-benchmark over the UNet3D cfg2 shapes: sweeps tile config x ring depth.
usage (GPU box): python tools/gemm_bench.py > gpurun_out/gemm_sweep.txt"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from followyourclick_amd import ops

# (kind, M, N, K, epi, res, conv(H,W,Cin,stride,up2) or None, weight in ms/step of the first profile)
SHAPES = [
    ("gemm", 131072, 2560, 320, 1, 0, None), ("gemm", 32768, 5120, 640, 1, 0, None), ("gemm", 8192, 10240, 1280, 1, 0, None),
    ("gemm", 131072, 320, 320, 0, 1, None), ("gemm", 32768, 640, 640, 0, 1, None), ("gemm", 8192, 1280, 1280, 0, 1, None),
    ("gemm", 131072, 320, 1280, 0, 1, None), ("gemm", 32768, 640, 2560, 0, 1, None), ("gemm", 8192, 1280, 5120, 0, 1, None),
    ("gemm", 131072, 960, 320, 0, 0, None), ("gemm", 32768, 1920, 640, 0, 0, None), ("gemm", 8192, 3840, 1280, 0, 0, None),
    ("gemm", 2048, 1280, 1280, 0, 1, None),
    ("conv", 131072, 320, 2880, 0, 1, (64, 64, 320, 1, 0)), ("conv", 32768, 640, 5760, 0, 1, (32, 32, 640, 1, 0)),
    ("conv", 8192, 1280, 11520, 0, 1, (16, 16, 1280, 1, 0)), ("conv", 2048, 1280, 11520, 0, 1, (8, 8, 1280, 1, 0)),
    ("conv", 131072, 320, 5760, 0, 0, (64, 64, 640, 1, 0)), ("conv", 8192, 1280, 23040, 0, 0, (16, 16, 2560, 1, 0)),
]
import os as _os
STRIP = int(_os.environ.get("FYC_STRIP", "0"))   # tuning key 4: column-strip width of the tile order (-1 = row-major)
CFGS = [(1, 2), (2, 2), (3, 2), (5, 2), (6, 2), (7, 2), (8, 2), (10, 2)]


def main():
    h = ops.get()
    h.ensure_init(torch.device("cuda:0"))
    h.set_tuning(4, STRIP)
    dev = torch.device("cuda:0")
    h.ensure_init(dev)
    T = torch.bfloat16
    print("shape".ljust(52), " ".join(f"c{c}n{n}".rjust(7) for c, n in CFGS), "  (TFLOP/s)")
    for kind, M, N, K, epi, res, conv in SHAPES:
        w = (torch.randn(N, K, device=dev) / K ** 0.5).to(T)
        bias = torch.randn(N, device=dev)
        ocols = N // 2 if epi == 1 else N
        out = torch.empty(M, ocols, dtype=T, device=dev)
        r = torch.randn(M, N, device=dev).to(T) if res else None
        if conv:
            Hh, Ww, Cin, stride, up2 = conv
            frames = M // (Hh * Ww)
            a = torch.randn(frames * Hh * Ww, Cin, device=dev).to(T)
            kw = dict(M=M, N=N, K=K, lda=Cin, ldw=K, ldo=ocols, ldr=N, mode=1, conv=dict(Hout=Hh, Wout=Ww, Hin=Hh, Win=Ww, Cin=Cin, stride=1))
        else:
            a = torch.randn(M, K, device=dev).to(T)
            kw = dict(M=M, N=N, K=K, lda=K, ldw=K, ldo=ocols, ldr=N, epilogue=epi)
        row = []
        for cfg, ns in CFGS:
            h.set_tuning(1, cfg)
            h.set_tuning(2, ns)
            try:
                for _ in range(2):
                    h.gemm(a, w, out, bias=bias, residual=r, **kw)
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(8):
                    h.gemm(a, w, out, bias=bias, residual=r, **kw)
                e.record()
                torch.cuda.synchronize()
                ms = s.elapsed_time(e) / 8
                row.append(2.0 * M * N * K / (ms * 1e-3) / 1e12)
            except Exception as ex:
                row.append(float("nan"))
        h.set_tuning(1, 0)
        h.set_tuning(2, 0)
        print(f"{kind} M={M} N={N} K={K} epi={epi} res={res}".ljust(52), " ".join(f"{v:7.0f}" for v in row))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
