"""`diffusers.AutoencoderKL` surface on the MI355X engine: `decode` (sampling path, reference
diffusers/models/vae.py:147-224, 575-610) and `encode` (first-frame conditioning front-end, :67-144,
565-573).  All parameters are registered under the reference's names so SD-1.5
`vae/diffusion_pytorch_model.bin` loads with a plain load_state_dict."""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional, Tuple

import torch
import torch.nn as nn

from followyourclick_amd import default_compute_dtype, ops as ops_mod
from followyourclick_amd.engine import VAEDecoderConfig
from followyourclick_amd.engine.schema import vae_decoder_schema, vae_encoder_schema
from followyourclick_amd.engine.vae import VAEDecoderEngine, VAEEncoderEngine
from followyourclick_amd.engine.weights import pack_vae_decoder, pack_vae_encoder


@dataclass
class DecoderOutput:
    sample: torch.Tensor


class DiagonalGaussianDistribution:
    """reference diffusers/models/vae.py:341-361 (mean | logvar moments, reparameterised sampling)"""

    def __init__(self, parameters: torch.Tensor, deterministic: bool = False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        noise = torch.randn(self.mean.shape, generator=generator, device=self.parameters.device)
        return self.mean + self.std * noise.to(self.parameters.dtype)

    def mode(self) -> torch.Tensor:
        return self.mean


@dataclass
class AutoencoderKLOutput:
    latent_dist: DiagonalGaussianDistribution


class _Node(nn.Module):
    pass


class AutoencoderKL(nn.Module):
    def __init__(self, in_channels: int = 3, out_channels: int = 3, down_block_types: Tuple[str, ...] = ("DownEncoderBlock2D",),
                 up_block_types: Tuple[str, ...] = ("UpDecoderBlock2D",), block_out_channels: Tuple[int, ...] = (64,),
                 layers_per_block: int = 1, act_fn: str = "silu", latent_channels: int = 4, norm_num_groups: int = 32,
                 sample_size: int = 32, compute_dtype: torch.dtype = None, **unused):
        super().__init__()
        self.engine_config = VAEDecoderConfig(latent_channels=latent_channels, out_channels=out_channels,
                                              block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                                              norm_num_groups=norm_num_groups)
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels, down_block_types=down_block_types,
                                      up_block_types=up_block_types, block_out_channels=tuple(block_out_channels),
                                      layers_per_block=layers_per_block, act_fn=act_fn, latent_channels=latent_channels,
                                      norm_num_groups=norm_num_groups, sample_size=sample_size)
        self.compute_dtype = compute_dtype if compute_dtype is not None else default_compute_dtype()
        g = torch.Generator().manual_seed(0)
        schema = dict(vae_encoder_schema(self.engine_config, in_channels))
        schema.update(vae_decoder_schema(self.engine_config))
        for name, shape in schema.items():
            *path, leaf = name.split(".")
            node = self
            for part in path:
                if part not in node._modules:
                    node.add_module(part, _Node())
                node = node._modules[part]
            if len(shape) == 1:
                init = torch.ones(shape) if leaf == "weight" else torch.zeros(shape)
            else:
                init = torch.randn(shape, generator=g) / (torch.Size(shape[1:]).numel() ** 0.5)
            node.register_parameter(leaf, nn.Parameter(init, requires_grad=False))
        self.use_slicing = False
        self._engine: Optional[VAEDecoderEngine] = None
        self._encoder: Optional[VAEEncoderEngine] = None
        self._engine_key = None
        self._encoder_key = None

    @property
    def dtype(self):
        return self.post_quant_conv.weight.dtype

    @property
    def device(self):
        return self.post_quant_conv.weight.device

    def enable_slicing(self):
        self.use_slicing = True      # accepted for API compatibility; the engine batches frames anyway

    def disable_slicing(self):
        self.use_slicing = False

    def _get_engine(self) -> VAEDecoderEngine:
        key = (self.device, self.compute_dtype) + tuple(p._version for p in self.parameters())
        if self._engine is None or key != self._engine_key:
            if self.device.type != "cuda" and ops_mod.get().name == "hip":
                raise RuntimeError("AutoencoderKL.decode runs on an MI355X HIP device only (call .to('cuda')); no CPU fallback")
            sd = {k: v for k, v in self.state_dict().items() if k.startswith(("decoder.", "post_quant_conv."))}
            packed = pack_vae_decoder(sd, self.engine_config, self.compute_dtype, self.device)
            self._engine = VAEDecoderEngine(packed)
            self._engine_key = key
        return self._engine

    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """x: (N, 3, H, W) in [-1, 1] -> AutoencoderKLOutput(latent_dist); `.latent_dist.sample() * 0.18215` gives the
        first-frame latents exactly as scripts/inference.py:356-358 computes them."""
        key = (self.device, self.compute_dtype) + tuple(p._version for p in self.parameters())
        if self._encoder is None or key != self._encoder_key:
            if self.device.type != "cuda" and ops_mod.get().name == "hip":
                raise RuntimeError("AutoencoderKL.encode runs on an MI355X HIP device only (call .to('cuda')); no CPU fallback")
            sd = {k: v for k, v in self.state_dict().items() if k.startswith(("encoder.", "quant_conv."))}
            self._encoder = VAEEncoderEngine(pack_vae_encoder(sd, self.engine_config, self.compute_dtype, self.device))
            self._encoder_key = key
        moments = self._encoder.encode_moments(x.float()).to(x.dtype)
        posterior = DiagonalGaussianDistribution(moments)
        if not return_dict:
            return (posterior,)
        return AutoencoderKLOutput(latent_dist=posterior)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True):
        """z: (N, 4, h, w) latents already divided by 0.18215 -> DecoderOutput(sample (N,3,8h,8w) in [-1, 1]-ish)"""
        eng = self._get_engine()
        sample = eng.decode(z.float() * self.engine_config.scaling_factor, raw=True).to(z.dtype)   # unclamped, like the reference
        if not return_dict:
            return (sample,)
        return DecoderOutput(sample=sample)

    @torch.no_grad()
    def decode_video01(self, latents: torch.Tensor) -> torch.Tensor:
        """(b,4,f,h,w) model-space latents -> (b,3,f,H,W) f32 in [0,1]: the whole decode_latents of the
        reference pipeline (pipeline_animation.py:400-413) in one engine call, no per-frame loop."""
        return self._get_engine().decode_video(latents)

    @classmethod
    def from_config(cls, config: dict, **kwargs):
        cfg = {k: v for k, v in dict(config).items() if not k.startswith("_")}
        cfg.update(kwargs)
        return cls(**cfg)

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, **kwargs):
        if subfolder is not None:
            pretrained_model_path = os.path.join(pretrained_model_path, subfolder)
        with open(os.path.join(pretrained_model_path, "config.json")) as f:
            model = cls.from_config(json.load(f), **kwargs)
        sd = torch.load(os.path.join(pretrained_model_path, "diffusion_pytorch_model.bin"), map_location="cpu")
        model.load_state_dict(sd, strict=False)
        return model
