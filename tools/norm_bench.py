"""GB/s of the HBM-bound kernels (GroupNorm statistics / apply, LayerNorm) at the engine's cfg2 shapes."""
import torch
from followyourclick_amd.ops import HipOps

ops = HipOps(); dev = torch.device("cuda:0"); ops.ensure_init(dev)

def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3

# (samples, rows_per_sample, C): resnet GN is per clip over F*H*W rows, transformer GN per frame
GN = [(2, 16 * 4096, 320), (2, 16 * 4096, 640), (2, 16 * 4096, 960), (2, 16 * 1024, 640), (2, 16 * 1024, 1280), (2, 16 * 1024, 1920),
      (2, 16 * 256, 1280), (2, 16 * 256, 2560), (2, 16 * 64, 1280), (2, 16 * 64, 2560),
      (32, 4096, 320), (32, 1024, 640), (32, 256, 1280), (32, 64, 1280)]
for S, R, C in GN:
    rows = S * R
    x = torch.randn(rows, C, device=dev, dtype=torch.bfloat16); y = torch.empty_like(x)
    st = torch.zeros(S * 32 * 2, device=dev, dtype=torch.float64)
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    t1 = timeit(lambda: ops.gn_stats(x, st, rows=rows, C_=C, groups=32, rows_per_sample=R))
    t2 = timeit(lambda: ops.gn_apply(x, st, g, b, y, rows=rows, C_=C, groups=32, rows_per_sample=R, eps=1e-5, silu=True))
    nb = rows * C * 2
    print(f"gn S={S:3d} R={R:6d} C={C:5d}: stats {t1*1e6:7.1f} us {nb/t1/1e9:7.0f} GB/s | apply {t2*1e6:7.1f} us {2*nb/t2/1e9:7.0f} GB/s", flush=True)
for rows, C in [(131072, 320), (32768, 640), (8192, 1280), (2048, 1280)]:
    x = torch.randn(rows, C, device=dev, dtype=torch.bfloat16); y = torch.empty_like(x)
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    t = timeit(lambda: ops.layernorm(x, g, b, y, rows=rows, C_=C))
    print(f"ln rows={rows:6d} C={C:5d}: {t*1e6:7.1f} us {2*rows*C*2/t/1e9:7.0f} GB/s", flush=True)
