"""Hyper-parameters of the hot path, named after the reference constructor arguments they mirror."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple


@dataclass
class UNet3DConfig:
    """UNet3DConditionModel.__init__ (reference animatediff/models/unet.py:43-104) restricted to the
    options the shipped inference YAMLs exercise (configs/inference/*.yaml)."""
    sample_size: int = 64
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    cross_attention_dim: int = 768
    attention_head_dim: int = 8            # = number of heads (reference unet_blocks.py:437-440)
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D")
    up_block_types: Tuple[str, ...] = ("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D")
    use_motion_module: bool = True
    motion_module_resolutions: Tuple[int, ...] = (1, 2, 4, 8)
    motion_module_mid_block: bool = False
    motion_num_attention_heads: int = 8
    motion_num_transformer_block: int = 1
    motion_attention_blocks: int = 2
    temporal_position_encoding: bool = True
    temporal_position_encoding_max_len: int = 24
    use_fps_condition: bool = True
    use_camera_motion_condition: bool = False      # camera_motion_embedding added to the time embedding (reference unet.py:134-137, 538-544)
    use_first_frame_mask_condition_concat: bool = True
    use_first_frame_condition_concat: bool = False
    use_ip_cross_attention: bool = False
    ip_scale: float = 1.0
    ip_num_tokens: int = 4

    @property
    def conv_in_channels(self) -> int:
        if self.use_first_frame_condition_concat:
            return self.in_channels * 2
        if self.use_first_frame_mask_condition_concat:
            return self.in_channels * 2 + 1
        return self.in_channels

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4

    def validate(self) -> None:
        for c in self.block_out_channels:
            if c % 64:
                raise ValueError(f"block_out_channels must be multiples of 64 (got {c}): K tiles of the MFMA GEMM are 64 bf16")
            if c % self.attention_head_dim or (c // self.attention_head_dim) % 8:
                raise ValueError(f"head dim {c}/{self.attention_head_dim} must be a multiple of 8")
        if self.cross_attention_dim % 8:
            raise ValueError("cross_attention_dim must be a multiple of 8")


@dataclass
class VAEDecoderConfig:
    """AutoencoderKL decoder half (reference diffusers/models/vae.py:147-206, 545-563)."""
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215


@dataclass
class DDIMConfig:
    """DDIMScheduler.__init__ kwargs (reference diffusers/schedulers/scheduling_ddim.py:157-172);
    defaults = noise_scheduler_kwargs of the shipped YAML."""
    num_train_timesteps: int = 1000
    beta_start: float = 0.00085
    beta_end: float = 0.012
    beta_schedule: str = "linear"
    trained_betas: object = None
    clip_sample: bool = False
    set_alpha_to_one: bool = True
    steps_offset: int = 1
    prediction_type: str = "v_prediction"
    rescale_betas_zero_snr: bool = True
