"""`animatediff.utils.util.save_videos_grid` (reference animatediff/utils/util.py:18-30): host-side GIF writer, the last
step of scripts/inference.py (:398-403).

The reference needs torchvision (make_grid) and imageio (mimsave).  Neither is part of the hot path nor of this image, so
the grid layout is restated here (torchvision.utils.make_grid semantics: `nrow` images per row, 2-pixel zero padding, a
single image is returned unpadded) and the GIF is written through imageio when it is importable, else through Pillow.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch


def make_grid(x: torch.Tensor, nrow: int = 8, padding: int = 2, pad_value: float = 0.0) -> torch.Tensor:
    """(B,C,H,W) -> (C', H_grid, W_grid) with C' = 3 for single-channel input, like torchvision.utils.make_grid"""
    if x.dim() != 4:
        raise ValueError("make_grid expects a (B,C,H,W) tensor")
    if x.shape[1] == 1:
        x = x.expand(-1, 3, -1, -1)
    if x.shape[0] == 1:
        return x[0]
    n = x.shape[0]
    xmaps = min(nrow, n)
    ymaps = int(math.ceil(n / xmaps))
    h, w = x.shape[2] + padding, x.shape[3] + padding
    grid = x.new_full((x.shape[1], h * ymaps + padding, w * xmaps + padding), pad_value)
    k = 0
    for yy in range(ymaps):
        for xx in range(xmaps):
            if k >= n:
                break
            grid[:, yy * h + padding:yy * h + h, xx * w + padding:xx * w + w] = x[k]
            k += 1
    return grid


def save_videos_grid(videos: torch.Tensor, path: str, rescale=False, n_rows=6, fps=8):
    """videos (b,c,t,h,w) in [0,1] (or [-1,1] with rescale=True) -> animated GIF of per-frame image grids"""
    frames = []
    for x in videos.permute(2, 0, 1, 3, 4):          # b c t h w -> t b c h w
        x = make_grid(x, nrow=n_rows).permute(1, 2, 0)
        if rescale:
            x = (x + 1.0) / 2.0
        frames.append((x * 255).numpy().astype(np.uint8))
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    try:
        import imageio
        imageio.mimsave(path, frames, fps=fps)
    except ImportError:
        from PIL import Image
        imgs = [Image.fromarray(f) for f in frames]
        imgs[0].save(path, save_all=True, append_images=imgs[1:], duration=int(round(1000.0 / fps)), loop=0)
