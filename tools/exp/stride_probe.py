"""Do the row strides of the GEMM operands matter?  (round 6)  The K loop of fyc_gemm is paced by the CU's vector-memory path (17-19 B per tick);
rows of A and W are K * 2 bytes apart - 2 560 B at K = 1280, 10 240 B at K = 5120 - and a DMA instruction touches 8 rows.  If the L2's channel
interleave sees such strides as a few channels only, padding the leading dimensions would be the fix.  Times the same problems with lda = ldw = K
and with both padded by 64 elements (128 B), cold operands (8 rotating buffer sets), us per launch.
    python tools/exp/stride_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from followyourclick_amd import ops

T, DEV = torch.bfloat16, torch.device("cuda:0")


def run(h, M, N, K, pad, nb=8, reps=6):
    ld = K + pad
    sets = [(torch.randn(M, ld, device=DEV).to(T), torch.empty(M, N, dtype=T, device=DEV)) for _ in range(nb)]
    w = (torch.randn(N, ld, device=DEV) / K ** 0.5).to(T)
    bias = torch.randn(N, device=DEV)
    kw = dict(M=M, N=N, K=K, lda=ld, ldw=ld, ldo=N, bias=bias)
    for i in range(2 * nb):
        h.gemm(sets[i % nb][0], w, sets[i % nb][1], **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(reps * nb):
        h.gemm(sets[i % nb][0], w, sets[i % nb][1], **kw)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1000 / (reps * nb)
    return us, 2.0 * M * N * K / us / 1e6


def main():
    h = ops.get()
    h.ensure_init(DEV)
    print("shape                      ld = K: us  TF/s | ld = K + 64: us  TF/s | ld = K + 8: us TF/s")
    for M, N, K in [(32768, 640, 640), (32768, 1920, 640), (8192, 1280, 1280), (8192, 3840, 1280), (32768, 640, 2560), (8192, 1280, 5120), (32768, 5120, 640), (8192, 10240, 1280),
                    (32768, 640, 5120), (16384, 2560, 2560), (16384, 2560, 2048), (16384, 2560, 4096)]:
        a, b, c = run(h, M, N, K, 0), run(h, M, N, K, 64), run(h, M, N, K, 8)
        print(f"{M:6d} x {N:5d} x {K:5d}   {a[0]:8.1f} {a[1]:6.0f} | {b[0]:8.1f} {b[1]:6.0f} | {c[0]:8.1f} {c[1]:6.0f}")


if __name__ == "__main__":
    main()
