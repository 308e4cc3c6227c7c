"""Where does the packed LINEAR epilogue of fyc_gemm spend its time?  Needs the timing build (-DFYC_TRACE):

    FYC_BUILD_EXTRA="-DFYC_TRACE" FYC_BUILD_LIB=tools/exp/libfyc_trace.so python -m followyourclick_amd._build
    FYC_LIB_PATH=tools/exp/libfyc_trace.so python tools/gemm_epilogue_trace.py          (GPU box)

Per case and tile config, mean over workgroups and tiles of waves 0 / 4 (s_memtime ticks of 10 ns): K loop, the wait for the epilogue's first
barrier, its inputs (column constants, row statistics), pass 1 (final values packed), pass 2 (staging + stores), statistics flush, and the gap
to the next tile's K loop (residual load)."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from followyourclick_amd import _lib, ops
from tools.gemm_probe import run

DEV = torch.device("cuda:0")
CASES = [
    ("tQKV L0 ln rb", 131072, 960, 320, dict(ln=True, rowbias=True)),
    ("tQKV L1 ln rb", 32768, 1920, 640, dict(ln=True, rowbias=True)),
    ("to_out L0 res", 131072, 320, 320, dict(res=True)),
    ("to_out L1 res", 32768, 640, 640, dict(res=True)),
    ("proj_in L0", 131072, 320, 320, dict()),
    ("FF2|proj L1 res st", 32768, 640, 3200, dict(res=True, stats=True, k2=640)),
    ("conv L0 res st", 131072, 320, 2880, dict(res=True, stats=True, conv=(64, 64, 320))),
    ("conv L0 rb st", 131072, 320, 2880, dict(rowbias=True, stats=True, conv=(64, 64, 320))),
]


def main():
    h = ops.get()
    h.ensure_init(DEV)
    lib = _lib.load()
    lib.fyc_set_trace.argtypes = [ctypes.c_void_p]
    trace = torch.zeros(2 * 65536, dtype=torch.int64, device=DEV)
    lib.fyc_set_trace(trace.data_ptr())
    for kv in filter(None, os.environ.get("PROBE_TUNING", "").split(",")):
        k, v = kv.split("=")
        h.set_tuning(int(k), int(v))
    cfgs = [int(c) for c in os.environ.get("PROBE_CFGS", "5,6").split(",")]
    print("ticks of s_memtime (100 MHz: 1 tick = 10 ns); mean over workgroups x tiles, [wave 0 | wave 4]")
    print("case".ljust(20), "cfg      us | K loop  barrier  inputs   pass1   pass2   flush     gap")
    for name, M, N, K, kw in CASES:
        for c in cfgs:
            trace.zero_()
            us, tf = run(h, M, N, K, nb=8, tile=c, reps=1, **kw)
            t = trace.cpu()
            a, e = t[:65536].reshape(256, 2, 128), t[65536:].reshape(256, 2, 128)
            out = []
            for w in (0, 1):
                acc = torch.zeros(7, dtype=torch.float64)
                n = 0
                for b in range(256):
                    s, x = a[b, w], e[b, w]
                    ns, nx = int((s != 0).sum()), int((x != 0).sum())
                    nt = min(ns // 3, nx // 4)
                    if nt < 1:
                        continue
                    s, x = s[: nt * 3].reshape(nt, 3).double(), x[: nt * 4].reshape(nt, 4).double()
                    acc[0] += (s[:, 1] - s[:, 0]).sum()          # K loop
                    acc[1] += (x[:, 0] - s[:, 1]).sum()          # -> behind the first barrier of the epilogue
                    acc[2] += (x[:, 1] - x[:, 0]).sum()          # inputs
                    acc[3] += (x[:, 2] - x[:, 1]).sum()          # pass 1
                    acc[4] += (x[:, 3] - x[:, 2]).sum()          # pass 2
                    acc[5] += (s[:, 2] - x[:, 3]).sum()          # statistics flush
                    if nt > 1:
                        acc[6] += (s[1:, 0] - s[:-1, 2]).sum() * nt / (nt - 1)
                    n += nt
                out.append(" ".join(f"{v / max(n, 1):7.0f}" for v in acc))
            print(f"{name:20s} c{c:<3d} {us:7.1f} | " + " | ".join(out), flush=True)
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
