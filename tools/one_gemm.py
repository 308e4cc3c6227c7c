"""Run one GEMM/conv shape a few times (for rocprofv3 --pmc runs).  usage: one_gemm.py kind M N K cfg [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from followyourclick_amd import ops
kind, M, N, K, cfg = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
h = ops.get(); dev = torch.device("cuda:0"); h.ensure_init(dev)
T = torch.bfloat16
w = (torch.randn(N, K, device=dev) / K ** 0.5).to(T); out = torch.empty(M, N, dtype=T, device=dev)
h.set_tuning(1, cfg)
if kind == "conv":
    Cin = K // 9
    side = {131072: 64, 32768: 32, 8192: 16, 2048: 8}[M]
    frames = M // (side * side)
    a = torch.randn(M, Cin, device=dev).to(T)
    kw = dict(M=M, N=N, K=K, lda=Cin, ldw=K, ldo=N, mode=1, conv=dict(Hout=side, Wout=side, Hin=side, Win=side, Cin=Cin, stride=1))
else:
    a = torch.randn(M, K, device=dev).to(T)
    kw = dict(M=M, N=N, K=K, lda=K, ldw=K, ldo=N)
for _ in range(reps):
    h.gemm(a, w, out, **kw)
torch.cuda.synchronize()
