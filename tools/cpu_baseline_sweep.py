"""How the CPU oracle's DDIM step scales with the thread count on THIS host (threads bound to NUMA node 0, physical cores only):
the numbers behind bench.py's CPU_THREADS_MAX.  One CFG-pair UNet3D forward at 512x512 on `frames` frames per thread count.
    python tools/cpu_baseline_sweep.py [frames] [counts...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from followyourclick_amd.engine import UNet3DConfig  # noqa: E402
from followyourclick_amd.engine.schema import random_state_dict, unet_schema  # noqa: E402
from oracle import functional as Fn  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    counts = [int(a) for a in sys.argv[2:]] or [8, 16, 32, 64]
    sd = random_state_dict(unet_schema(UNet3DConfig()), seed=0)
    cfg = Fn.UNetConfig()
    g = torch.Generator().manual_seed(1)
    text = torch.randn(2, 77, cfg.cross_attention_dim, generator=g)
    cores, logical = bench.numa_node_cores(0)
    print(f"host: {logical} logical cpus, NUMA node 0 has {len(cores)} physical cores", flush=True)
    for n in counts:
        use = cores[:n]
        if len(use) < n:
            print(f"{n} threads: node 0 has only {len(cores)} cores, skipped")
            continue
        os.sched_setaffinity(0, set(use))
        torch.set_num_threads(n)
        x9 = torch.randn(2, cfg.conv_in_channels, frames, 64, 64, generator=g)
        t0 = time.time()
        with torch.no_grad():
            Fn.unet3d_forward(sd, cfg, x9, torch.tensor(961), text, torch.tensor([2, 2]), torch.tensor([4, 4]))
        dt = time.time() - t0
        print(f"{n:3d} threads: {dt:6.1f} s for {frames} frame(s) = {dt / frames:5.2f} s/frame (first call at this count, no warm-up)", flush=True)


if __name__ == "__main__":
    main()
