import ctypes, os, sys
sys.path.insert(0, "/root/repo")
import torch
from followyourclick_amd import _lib, ops
from tools.gemm_probe import run
DEV = torch.device("cuda:0")
h = ops.get(); h.ensure_init(DEV)
lib = _lib.load(); lib.fyc_set_trace.argtypes = [ctypes.c_void_p]
trace = torch.zeros(2 * 65536, dtype=torch.int64, device=DEV); lib.fyc_set_trace(trace.data_ptr())
for c in (5, 6):
    trace.zero_()
    us, tf = run(h, 131072, 960, 320, nb=8, tile=c, reps=1, ln=True, rowbias=True)
    t = trace.cpu(); a, e = t[:65536].reshape(256, 2, 128), t[65536:].reshape(256, 2, 128)
    for b in (0, 100):
        s, x = a[b, 0], e[b, 0]
        t0 = int(s[0])
        print("cfg", c, "block", b, "S:", [int(v) - t0 for v in s[:21] if v != 0])
        print("cfg", c, "block", b, "E:", [int(v) - t0 for v in x[:28] if v != 0])
