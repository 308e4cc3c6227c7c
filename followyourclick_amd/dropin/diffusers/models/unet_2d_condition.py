"""`diffusers.models.UNet2DConditionModel` on the MI355X engine (first-image synthesis of scripts/inference.py:194-204).

The 2-D SD-1.5 UNet is the spatial half of the 3-D model: same resnets, spatial transformers, up/down-samplers and
state-dict names, no motion modules, no fps/flow embeddings, 4 input channels.  It therefore runs on the same engine as a
one-frame clip (per-image GroupNorm == cross-frame GroupNorm with F = 1).  Call surface: reference
diffusers/models/unet_2d_condition.py:92-123 (constructor), :337-345 (forward).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import Tuple, Union

import torch

from animatediff.models.unet import UNet3DConditionModel

_TO_3D = {"CrossAttnDownBlock2D": "CrossAttnDownBlock3D", "DownBlock2D": "DownBlock3D", "UpBlock2D": "UpBlock3D",
          "CrossAttnUpBlock2D": "CrossAttnUpBlock3D", "UNetMidBlock2DCrossAttn": "UNetMidBlock3DCrossAttn"}


@dataclass
class UNet2DConditionOutput:
    sample: torch.Tensor

    def __getitem__(self, i):
        return (self.sample,)[i]


class UNet2DConditionModel(UNet3DConditionModel):
    def __init__(self, sample_size=None, in_channels: int = 4, out_channels: int = 4, center_input_sample: bool = False,
                 flip_sin_to_cos: bool = True, freq_shift: int = 0,
                 down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
                 mid_block_type: str = "UNetMidBlock2DCrossAttn",
                 up_block_types: Tuple[str, ...] = ("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
                 only_cross_attention=False, block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280), layers_per_block: int = 2,
                 downsample_padding: int = 1, mid_block_scale_factor: float = 1, act_fn: str = "silu", norm_num_groups: int = 32,
                 norm_eps: float = 1e-5, cross_attention_dim: int = 1280, attention_head_dim=8, dual_cross_attention: bool = False,
                 use_linear_projection: bool = False, class_embed_type=None, num_class_embeds=None, upcast_attention: bool = False,
                 resnet_time_scale_shift: str = "default", compute_dtype: torch.dtype = None, **unused):
        cfg2d = {k: v for k, v in locals().items() if k not in ("self", "unused", "__class__", "compute_dtype")}
        try:
            down3, up3, mid3 = [_TO_3D[b] for b in down_block_types], [_TO_3D[b] for b in up_block_types], _TO_3D[mid_block_type]
        except KeyError as e:
            raise NotImplementedError(f"UNet2DConditionModel on the MI355X engine: block type {e} is not implemented") from None
        if only_cross_attention or upcast_attention or downsample_padding != 1 or mid_block_scale_factor != 1:
            raise NotImplementedError("only_cross_attention / upcast_attention / non-default downsample_padding, mid_block_scale_factor")
        kw3 = dict(cfg2d, down_block_types=tuple(down3), up_block_types=tuple(up3), mid_block_type=mid3)
        super().__init__(**kw3, use_motion_module=False, compute_dtype=compute_dtype)
        for k, v in cfg2d.items():          # `.config` and attributes show the 2-D constructor arguments
            setattr(self, k, v)
        self.config = type(self.config)(**cfg2d)

    @torch.no_grad()
    def forward(self, sample: torch.Tensor, timestep: Union[torch.Tensor, float, int], encoder_hidden_states: torch.Tensor,
                class_labels=None, attention_mask=None, return_dict: bool = True):
        if sample.dim() != 4:
            raise ValueError(f"sample must be (batch, channel, height, width), got {tuple(sample.shape)}")
        out = super().forward(sample[:, :, None], timestep, encoder_hidden_states, class_labels=class_labels,
                              attention_mask=attention_mask).sample[:, :, 0]
        if not return_dict:
            return (out,)
        return UNet2DConditionOutput(sample=out)

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, **kwargs):
        """config.json + diffusion_pytorch_model.{bin,safetensors} of a diffusers checkpoint directory (reference
        diffusers/modeling_utils.py:287-470, local-directory branch; there is no hub access here)."""
        if subfolder is not None:
            pretrained_model_path = os.path.join(pretrained_model_path, subfolder)
        config_file = os.path.join(pretrained_model_path, "config.json")
        if not os.path.isfile(config_file):
            raise EnvironmentError(f"Error no file named config.json found in directory {pretrained_model_path}.")
        with open(config_file) as f:
            model = cls.from_config(json.load(f), **kwargs)
        st = os.path.join(pretrained_model_path, "diffusion_pytorch_model.safetensors")
        if os.path.isfile(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            weights = os.path.join(pretrained_model_path, "diffusion_pytorch_model.bin")
            if not os.path.isfile(weights):
                raise EnvironmentError(f"Error no file named diffusion_pytorch_model.bin found in directory {pretrained_model_path}.")
            sd = torch.load(weights, map_location="cpu")
        missing, unexpected = model.load_state_dict(sd, strict=False)
        if missing:
            raise ValueError(f"checkpoint is missing {len(missing)} tensors, e.g. {missing[:3]}")
        return model
