"""Calibration only (not product): what do the vendor libraries (hipBLASLt via torch.matmul, MIOpen via F.conv2d)
reach on the engine's GEMM / conv shapes?  Gives an attainable-rate yardstick next to tools/gemm_bench.py."""
import torch
import torch.nn.functional as F

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3

GEMMS = [(131072, 2560, 320), (32768, 5120, 640), (8192, 10240, 1280), (131072, 320, 320), (32768, 640, 640),
         (8192, 1280, 1280), (131072, 320, 1280), (32768, 640, 2560), (8192, 1280, 5120), (131072, 960, 320),
         (32768, 1920, 640), (8192, 3840, 1280), (32768, 5120, 5120), (8192, 8192, 8192)]
CONVS = [(32, 64, 64, 320, 320), (32, 32, 32, 640, 640), (32, 16, 16, 1280, 1280), (32, 64, 64, 640, 320), (32, 16, 16, 2560, 1280)]
dev = "cuda"
for M, N, K in GEMMS:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: a @ w.t())
    print(f"matmul M={M} N={N} K={K}: {t*1e6:8.1f} us {2*M*N*K/t/1e12:7.1f} TF/s", flush=True)
for B, H, W, Ci, Co in CONVS:
    x = torch.randn(B, Ci, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Co, Ci, 3, 3, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    t = timeit(lambda: F.conv2d(x, w, padding=1))
    print(f"conv3x3 B={B} {H}x{W} Cin={Ci} Cout={Co}: {t*1e6:8.1f} us {2*B*H*W*Co*Ci*9/t/1e12:7.1f} TF/s", flush=True)
q = torch.randn(32, 8, 4096, 40, device=dev, dtype=torch.bfloat16)
t = timeit(lambda: F.scaled_dot_product_attention(q, q, q), 5)
print(f"sdpa B=32 H=8 N=4096 d=40: {t*1e6:8.1f} us {4*32*8*4096*4096*40/t/1e12:7.1f} TF/s")
