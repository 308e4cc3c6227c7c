"""Run one attention shape a few times (for rocprofv3 --pmc runs).  usage: one_attn.py B H nq nk d [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from followyourclick_amd import ops
B, H, n, nk, d = (int(v) for v in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
h = ops.get(); dev = torch.device("cuda:0"); h.ensure_init(dev)
T = torch.bfloat16
q = torch.randn(B * H, n, d, device=dev).to(T); k = torch.randn(B * H, nk, d, device=dev).to(T)
ld = (nk + 7) // 8 * 8
vt = torch.zeros(B * H, d, ld, device=dev, dtype=T); vt[..., :nk] = torch.randn(B * H, d, nk, device=dev).to(T)
o = torch.empty(B * n, H * d, dtype=T, device=dev)
for _ in range(reps):
    h.attention(q, k, vt, o, batch=B, heads=H, n_q=n, n_k=nk, d=d, ldo=H * d, ldvt=ld, scale=d ** -0.5)
torch.cuda.synchronize()
