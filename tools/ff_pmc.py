"""a few fyc_ff_block launches at the 64x64-level shape, for rocprofv3 --pmc runs: python tools/ff_pmc.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from followyourclick_amd import ops
from followyourclick_amd.engine.weights import Packed, pack_ff_block

T, DEV, C, HID, rows = torch.bfloat16, torch.device("cuda:0"), 320, 1280, 131072
h = ops.get()
h.ensure_init(DEV)
w1 = (torch.randn(2 * HID, C, device=DEV) * C ** -0.5).to(T)
ff = Packed(w1=w1, b1=torch.randn(2 * HID, device=DEV) * 0.1, cs1=w1.float().sum(dim=1).contiguous(),
            po_w=(torch.randn(C, C + HID, device=DEV) * (C + HID) ** -0.5).to(T), po_b=torch.randn(C, device=DEV) * 0.1)
ws = pack_ff_block(ff)
x, r, o = torch.randn(rows, C, device=DEV).to(T), torch.randn(rows, C, device=DEV).to(T), torch.empty(rows, C, dtype=T, device=DEV)
for _ in range(4):
    h.ff_block(x, r, o, wstream=ws, b_out=ff.po_b, rows=rows, C_=C, hidden=HID)
torch.cuda.synchronize()
