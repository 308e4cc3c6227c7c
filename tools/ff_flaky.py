"""debug (round 3): the body of tests/test_kernels_gpu.py::test_ff_block_matches_unfused_schedule in a loop, naming the side that is off"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
from followyourclick_amd import ops as ops_mod, _lib as L
from followyourclick_amd.engine.weights import pack_ff_block
from test_kernels_gpu import _ff_operands, rnd
from emu_ops import EmuOps
hip = ops_mod.get(); emu = EmuOps(acc=torch.float64)
T, C, hid = torch.bfloat16, 320, 1280
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
for it in range(16):
    rows = 16384 if it % 2 == 0 else 8192
    ff = _ff_operands(10)
    ws = pack_ff_block(ff)
    xc = (rnd((rows, C), torch.float32, 6) * 1.2 - 0.2).to(T)
    rc = rnd((rows, C), T, 7)
    x, res = xc.cuda(), rc.cuda()
    w1, b1, cs1, po_w, po_b = (t.cuda() for t in (ff.w1, ff.b1, ff.cs1, ff.po_w, ff.po_b))
    o_f = torch.full((rows, C), float("nan"), dtype=T, device="cuda")
    hip.ff_block(x, res, o_f, wstream=ws.cuda(), b_out=po_b, rows=rows, C_=C, hidden=hid)
    st = torch.empty(rows, 2, dtype=torch.float32, device="cuda")
    hip.row_stats(x, st, rows=rows, C_=C)
    hmid = torch.full((rows, hid), float("nan"), dtype=T, device="cuda")
    hip.gemm(x, w1, hmid, M=rows, N=2 * hid, K=C, lda=C, ldw=C, ldo=hid, bias=b1, epilogue=L.EPI_GEGLU, ln_colsum=cs1, ln_stats=st)
    o_u = torch.full((rows, C), float("nan"), dtype=T, device="cuda")
    hip.gemm(x, po_w, o_u, M=rows, N=C, K=C + hid, lda=C, ldw=C + hid, ldo=C, bias=po_b, residual=res, ldr=C, a2=hmid, k_split=C, lda2=hid)
    torch.cuda.synchronize()
    n = 512
    o_e = torch.zeros(n, C, dtype=T)
    emu.ff_block(xc[:n], rc[:n], o_e, wstream=ws, b_out=ff.po_b, rows=n, C_=C, hidden=hid)
    o_e2 = torch.zeros(n, C, dtype=T)
    emu.ff_block(xc[:n], rc[:n], o_e2, wstream=ws, b_out=ff.po_b, rows=n, C_=C, hidden=hid)
    wsd = ws.cuda()
    o_f2 = torch.full((rows, C), float("nan"), dtype=T, device="cuda")
    hip.ff_block(x, res, o_f2, wstream=wsd, b_out=po_b, rows=rows, C_=C, hidden=hid)
    torch.cuda.synchronize()
    print(f"it {it} rows {rows}: ff/spec {rel(o_f[:n].cpu(), o_e):.3e}  unfused/spec {rel(o_u[:n].cpu(), o_e):.3e}  ff/unfused {rel(o_f.cpu(), o_u.cpu()):.3e}  "
          f"spec/spec2 {rel(o_e2, o_e):.3e}  ff2/unfused {rel(o_f2.cpu(), o_u.cpu()):.3e}  ff2==ff {torch.equal(o_f2.view(torch.int16), o_f.view(torch.int16))}", flush=True)
