"""Per-kernel register / scratch / LDS usage of libfyc_hip.so's translation units (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/resource_usage.py [file.hip ...]      # default: every source of the build"""
import concurrent.futures as cf
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from followyourclick_amd import _build  # noqa: E402


def one(src):
    cmd = [_build._hipcc(), *_build.FLAGS, "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c",
           os.path.join(_build.CSRC, src), "-o", os.devnull]
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows = []
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = b.split("\n")[0]
        g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(.*$", "", dem).replace("void ", "")
        rows.append((src, dem[:110], g("VGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
    return rows


if __name__ == "__main__":
    srcs = sys.argv[1:] or _build.SOURCES
    with cf.ThreadPoolExecutor(max_workers=8) as ex:
        for rows in ex.map(one, srcs):
            for r in rows:
                print(f"{r[0]:22s} vgpr {r[2]:3d} scratch {r[3]:4d} occ {r[4]} lds {r[5]:6d}  {r[1]}")
