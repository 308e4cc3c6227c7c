"""Where does a GEMM workgroup spend its time?  Needs the timing build of the library (-DFYC_TRACE: waves 0 and 4 of every
workgroup stamp s_memtime in front of the K loop, behind it and behind the epilogue of every output tile):

    FYC_BUILD_EXTRA="-DFYC_TRACE" FYC_BUILD_LIB=tools/exp/libfyc_trace.so python -m followyourclick_amd._build
    FYC_LIB_PATH=tools/exp/libfyc_trace.so python tools/gemm_phase_probe.py > gpurun_out/phase_probe.txt       (GPU box)

Per case and tile config: us per launch, and per workgroup (mean over workgroups, wave 0 / wave 4) the cycles of the K loop per
K tile, of the epilogue per tile, and between the end of an epilogue and the start of the next K loop."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from followyourclick_amd import _lib, ops
from tools.gemm_probe import run

DEV = torch.device("cuda:0")
CASES = [
    ("FF1 GEGLU L1 ln", 32768, 5120, 640, dict(epi=1, ln=True)),
    ("FF1 GEGLU L2 ln", 8192, 10240, 1280, dict(epi=1, ln=True)),
    ("tQKV L1 ln rb", 32768, 1920, 640, dict(ln=True, rowbias=True)),
    ("to_out L2 res", 8192, 1280, 1280, dict(res=True)),
    ("FF2|proj L1 res st", 32768, 640, 3200, dict(res=True, stats=True, k2=640)),
    ("conv L0 res st", 131072, 320, 2880, dict(res=True, stats=True, conv=(64, 64, 320))),
    ("conv L1 res st", 32768, 640, 5760, dict(res=True, stats=True, conv=(32, 32, 640))),
    ("conv L2 res st", 8192, 1280, 11520, dict(res=True, stats=True, conv=(16, 16, 1280))),
]


def main():
    h = ops.get()
    h.ensure_init(DEV)
    lib = _lib.load()
    lib.fyc_set_trace.argtypes = [ctypes.c_void_p]
    trace = torch.zeros(256 * 2 * 128, dtype=torch.int64, device=DEV)
    lib.fyc_set_trace(trace.data_ptr())
    cfgs = [int(c) for c in os.environ.get("PROBE_CFGS", "5,21,6,22").split(",")]
    print("cycles = s_memtime ticks; loop/kt = K loop per K tile of 64, epi = epilogue per tile, gap = epilogue end -> next K loop start; [wave 0 | wave 4]")
    for name, M, N, K, kw in CASES:
        for c in cfgs:
            try:
                trace.zero_()
                us, tf = run(h, M, N, K, nb=8, tile=c, reps=2, **kw)
            except Exception as e:
                print(f"{name:20s} c{c}: {repr(e)[:80]}")
                continue
            t = trace.cpu().reshape(256, 2, 128)
            kt = K // 64
            out = []
            for w in (0, 1):
                loops, epis, gaps, spans = [], [], [], []
                for b in range(256):
                    s = t[b, w]
                    n = int((s != 0).sum())
                    if n < 3:
                        continue
                    s = s[: n - n % 3].reshape(-1, 3).double()
                    loops.append(((s[:, 1] - s[:, 0]).mean() / kt).item())
                    epis.append((s[:, 2] - s[:, 1]).mean().item())
                    if s.shape[0] > 1:
                        gaps.append((s[1:, 0] - s[:-1, 2]).mean().item())
                    spans.append((s[-1, 2] - s[0, 0]).item())
                if loops:
                    m = lambda v: sum(v) / max(len(v), 1)      # noqa: E731
                    out.append(f"loop/kt {m(loops):7.0f} epi {m(epis):7.0f} gap {m(gaps):6.0f} span {m(spans):8.0f} ({len(loops)} wg)")
                else:
                    out.append("-")
            print(f"{name:20s} {M}x{N}x{K} c{c:<3d} {us:7.1f} us {tf:6.0f} TF | " + " | ".join(out), flush=True)
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
