"""`animatediff.utils.util`: `save_videos_grid` (reference animatediff/utils/util.py:18-30), the host-side GIF writer that is the
last step of scripts/inference.py (:398-403), and `load_weights` (:91-154), the checkpoint loader scripts/inference_w_camera_lora.py
imports.

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


def _read_checkpoint(path: str) -> dict:
    if path.endswith(".safetensors"):
        from safetensors import safe_open
        with safe_open(path, framework="pt", device="cpu") as f:
            return {k: f.get_tensor(k) for k in f.keys()}
    return torch.load(path, map_location="cpu")


def load_weights(animation_pipeline, motion_module_path="", motion_module_lora_configs=[], dreambooth_model_path="", lora_model_path="",
                 lora_alpha=0.8):
    """reference animatediff/utils/util.py:91-154, same arguments and order of effects: (1) the `motion_modules.` tensors of a
    motion-module checkpoint (with or without the `state_dict` wrapper) go into the UNet, anything the UNet does not know is an
    error; (2) a DreamBooth LDM checkpoint (.safetensors / .ckpt) replaces VAE, UNet spatial weights and text encoder through the
    `convert_ldm_*` key maps; (3) a kohya LoRA is merged with `lora_alpha`; (4) motion LoRAs are merged one by one with their own
    alpha.  Returns the pipeline."""
    from .convert_from_ckpt import convert_ldm_clip_checkpoint, convert_ldm_unet_checkpoint, convert_ldm_vae_checkpoint
    from .convert_lora_safetensor_to_diffusers import convert_lora, convert_motion_lora_ckpt_to_diffusers
    temporal = {}
    if motion_module_path != "":
        print(f"load motion module from {motion_module_path}")
        sd = torch.load(motion_module_path, map_location="cpu")
        sd = sd.get("state_dict", sd)
        temporal = {k: v for k, v in sd.items() if "motion_modules." in k}
    _, unexpected = animation_pipeline.unet.load_state_dict(temporal, strict=False)
    assert len(unexpected) == 0
    if dreambooth_model_path != "":
        print(f"load dreambooth model from {dreambooth_model_path}")
        if not dreambooth_model_path.endswith((".safetensors", ".ckpt")):
            raise ValueError(f"dreambooth checkpoint must be .safetensors or .ckpt: {dreambooth_model_path}")
        ldm = _read_checkpoint(dreambooth_model_path)
        animation_pipeline.vae.load_state_dict(convert_ldm_vae_checkpoint(ldm, animation_pipeline.vae.config))
        animation_pipeline.unet.load_state_dict(convert_ldm_unet_checkpoint(ldm, animation_pipeline.unet.config), strict=False)
        # (the reference builds a fresh CLIPTextModel from the hub config here; offline, the pipeline's own text encoder takes the weights)
        animation_pipeline.text_encoder = convert_ldm_clip_checkpoint(ldm, animation_pipeline.text_encoder)
    if lora_model_path != "":
        print(f"load lora model from {lora_model_path}")
        assert lora_model_path.endswith(".safetensors")
        animation_pipeline = convert_lora(animation_pipeline, _read_checkpoint(lora_model_path), alpha=lora_alpha)
    for entry in motion_module_lora_configs:
        path, alpha = entry["path"], entry["alpha"]
        print(f"load motion LoRA from {path}")
        sd = torch.load(path, map_location="cpu")
        animation_pipeline = convert_motion_lora_ckpt_to_diffusers(animation_pipeline, sd.get("state_dict", sd), alpha)
    return animation_pipeline
