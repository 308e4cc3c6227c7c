"""VAE decode of one 16-frame 512x512 clip (random-init AutoencoderKL decoder, bf16), nothing else on the GPU: the process to put
under `rocprofv3 --kernel-trace --stats` for profiles/rNN_vae_kernel_stats.txt.   python tools/vae_only.py [frames] [size] [repeats]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from followyourclick_amd.engine import VAEDecoderConfig  # noqa: E402
from followyourclick_amd.engine.schema import random_state_dict, vae_decoder_schema  # noqa: E402
from followyourclick_amd.engine.vae import VAEDecoderEngine  # noqa: E402
from followyourclick_amd.engine.weights import pack_vae_decoder  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    dev = torch.device("cuda", 0)
    vcfg = VAEDecoderConfig()
    vae = VAEDecoderEngine(pack_vae_decoder(random_state_dict(vae_decoder_schema(vcfg), 1), vcfg, torch.bfloat16, dev))
    lat = torch.randn(1, 4, frames, size // 8, size // 8, generator=torch.Generator().manual_seed(5)).to(dev)
    vae.decode_video(lat)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        vid = vae.decode_video(lat)
    torch.cuda.synchronize()
    print(f"vae decode {frames}f {size}x{size}: {1000 * (time.time() - t0) / reps:.1f} ms per clip ({reps} timed, 1 warm-up), finite={bool(torch.isfinite(vid).all())}")


if __name__ == "__main__":
    main()
