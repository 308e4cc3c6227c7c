"""`animatediff.utils.util.save_videos_grid` (reference animatediff/utils/util.py:18-30): host-side GIF
writer, outside the hot path.  Needs imageio + torchvision like the reference."""
import os

import numpy as np
import torch


def save_videos_grid(videos: torch.Tensor, path: str, rescale=False, n_rows=6, fps=8):
    try:
        import imageio
        import torchvision
    except ImportError as e:  # the reference has the same hard dependency
        raise ImportError("save_videos_grid needs imageio and torchvision (as in the reference)") from e
    frames = []
    for x in videos.permute(2, 0, 1, 3, 4):          # b c t h w -> t b c h w
        x = torchvision.utils.make_grid(x, nrow=n_rows).transpose(0, 1).transpose(1, 2).squeeze(-1)
        if rescale:
            x = (x + 1.0) / 2.0
        frames.append((x * 255).numpy().astype(np.uint8))
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    imageio.mimsave(path, frames, fps=fps)
