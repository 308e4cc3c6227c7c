"""a few register-resident fyc_temporal_block launches at the 64x64-level shape, for rocprofv3 --pmc runs"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
from followyourclick_amd import ops as ops_mod
from test_kernels_gpu import _temporal_operands, rnd
hip = ops_mod.get()
T, H, d, F, clips, P = torch.bfloat16, 8, 40, 16, 2, 4096
C = H * d
ops_h = {k: (v.cuda() if v is not None else None) for k, v in _temporal_operands(True).items()}
x = (rnd((clips * F * P, C), torch.float32, 6) * 1.5 + 0.3).to(T).cuda()
out = torch.empty_like(x)
for _ in range(4):
    hip.temporal_block(x, out, clips=clips, frames=F, pixels=P, heads=H, d=d, scale=d ** -0.5, **ops_h)
torch.cuda.synchronize()
