"""fyc_temporal_block at the 64x64 level of the 512^2 workload (2 clips x 16 frames x 4096 pixels, C = 320): the LDS-tile kernel
(csrc/temporal_block.hip) against the register-resident one on the packed stream (csrc/temporal_block_rr.hip).  HIP events over
20 launches, operands swept out of the caches in between (a 1-GiB fill) so that x comes from HBM as in the pipeline."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
from followyourclick_amd import ops as ops_mod
from test_kernels_gpu import _temporal_operands, rnd
hip = ops_mod.get()
T, H, d, F, clips, P = torch.bfloat16, 8, 40, 16, 2, 4096
C = H * d
ops_h = {k: (v.cuda() if v is not None else None) for k, v in _temporal_operands(True).items()}
x = (rnd((clips * F * P, C), torch.float32, 6) * 1.5 + 0.3).to(T).cuda()
kw = dict(clips=clips, frames=F, pixels=P, heads=H, d=d, scale=d ** -0.5)
out = torch.empty_like(x)
junk = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
rows = clips * F * P
flops = 2.0 * rows * C * 4 * C + 4.0 * rows * F * C
for name, o in (("LDS-tile kernel", {k: v for k, v in ops_h.items() if k != "wstream"}), ("register-resident kernel", ops_h)):
    ts = []
    for it in range(20):
        junk.fill_(it)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); hip.temporal_block(x, out, **o, **kw); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts = sorted(ts[2:])
    print(f"{name:28s} median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f} us   {flops / ts[len(ts) // 2] / 1e6:6.0f} TFLOP/s (useful flops)")
if hasattr(hip.lib, "fyc_tb_timing"):                        # TB_TIMING build: phase timestamps (shader clock) of wave 0 of every workgroup
    import ctypes, numpy as np
    hip.temporal_block(x, out, **ops_h, **kw); torch.cuda.synchronize()
    n = 1024 * 24
    buf = (ctypes.c_ulonglong * n)()
    assert hip.lib.fyc_tb_timing(buf, n) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 24).astype(np.int64)
    names = ["start", "x+LN done", "A(even h) begin", "A end", "B begin (after barrier)", "A(odd h) begin", "A end", "B begin", "B(even) end", "B(odd) end",
             "last A begin", "last B begin", "last B end", "epilogue: residual landed", "end"]
    for sel, lab in ((slice(0, 256), "first wave of workgroups (0..255)"), (slice(512, 768), "third round (512..767)")):
        d = t[sel]
        print(lab)
        for k in (1, 2, 3, 4, 8, 5, 6, 7, 9, 10, 11, 12, 13, 14):
            print(f"   {names[k]:28s} +{np.median(d[:, k] - d[:, 0]):9.0f} cycles (median over workgroups)")
    print("   (even/odd head marks hold the LAST such head: h = 6 / h = 5)")
