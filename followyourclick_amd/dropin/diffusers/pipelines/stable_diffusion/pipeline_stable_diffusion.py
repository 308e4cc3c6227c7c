"""`diffusers.StableDiffusionPipeline` on the MI355X engine: the 2-D first-image synthesis the reference's driver
scripts run before animating (scripts/inference.py:194-204, 300-306).

Call surface, defaults, errors and output type follow reference
diffusers/pipelines/stable_diffusion/pipeline_stable_diffusion.py:72-166 (constructor), :409-565 (`__call__`).  The loop
itself is the engine's DDIM sampler on a one-frame clip: per step ONE layout kernel, the UNet op schedule on the CFG pair and
ONE fused guidance + DDIM-update kernel; `decode_latents` is the VAE decoder engine.  No safety checker is run (the
scripts pass `safety_checker=None`).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Union

import numpy as np
import torch

from followyourclick_amd.engine.sampler import DDIMSampler


@dataclass
class StableDiffusionPipelineOutput:
    images: Union[list, np.ndarray]
    nsfw_content_detected: Optional[List[bool]]


class StableDiffusionPipeline:
    _optional_components = ["safety_checker", "feature_extractor"]

    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler, safety_checker=None, feature_extractor=None,
                 requires_safety_checker: bool = True):
        if safety_checker is not None:
            raise NotImplementedError("StableDiffusionPipeline on the MI355X engine runs without a safety checker "
                                      "(pass safety_checker=None as scripts/inference.py does)")
        if getattr(scheduler.config, "steps_offset", 1) != 1:     # reference :99-110 patches outdated configs
            scheduler.config.steps_offset = 1
            scheduler.engine_config.steps_offset = 1
        if getattr(scheduler.config, "clip_sample", False) is True:  # :112-123
            scheduler.config.clip_sample = False
            scheduler.engine_config.clip_sample = False
        self.vae, self.text_encoder, self.tokenizer, self.unet, self.scheduler = vae, text_encoder, tokenizer, unet, scheduler
        self.safety_checker, self.feature_extractor = None, feature_extractor
        self.vae_scale_factor = 2 ** (len(self.vae.config.block_out_channels) - 1)
        self._device = torch.device("cpu")

    @classmethod
    def from_pretrained(cls, pretrained_model_path, **components):
        """The scripts hand every component over as a keyword (scripts/inference.py:199-202); anything missing would
        have to come from the hub / a model_index.json walk, which this offline drop-in does not do."""
        need = ("vae", "text_encoder", "tokenizer", "unet", "scheduler")
        missing = [k for k in need if components.get(k) is None]
        if missing:
            raise ValueError(f"StableDiffusionPipeline.from_pretrained on the MI355X engine needs {missing} passed as keyword "
                             "arguments (components are not auto-loaded)")
        allowed = need + ("safety_checker", "feature_extractor", "requires_safety_checker")
        return cls(**{k: v for k, v in components.items() if k in allowed})

    def to(self, device):
        self._device = torch.device(device)
        for m in (self.vae, self.text_encoder, self.unet):
            if m is not None and hasattr(m, "to"):
                m.to(self._device)
        return self

    @property
    def device(self) -> torch.device:
        return self._device

    @property
    def _execution_device(self) -> torch.device:
        return self._device

    def enable_vae_slicing(self):
        self.vae.enable_slicing()

    def disable_vae_slicing(self):
        self.vae.disable_slicing()

    def progress_bar(self, iterable=None, total=None):
        try:
            from tqdm import tqdm
            return tqdm(iterable, total=total)
        except ImportError:  # pragma: no cover
            return iterable

    @staticmethod
    def numpy_to_pil(images: np.ndarray):
        """(N,H,W,3) in [0,1] -> list of PIL images (reference diffusers/pipeline_utils.py numpy_to_pil)"""
        from PIL import Image
        if images.ndim == 3:
            images = images[None, ...]
        images = (images * 255).round().astype("uint8")
        return [Image.fromarray(im) for im in images]

    def _encode_prompt(self, prompt, device, num_images_per_prompt, do_classifier_free_guidance, negative_prompt):
        """cat[uncond, cond] text states (reference :224-327)"""
        batch_size = len(prompt) if isinstance(prompt, list) else 1

        def enc(texts, max_length):
            inp = self.tokenizer(texts, padding="max_length", max_length=max_length, truncation=True, return_tensors="pt")
            use_mask = getattr(getattr(self.text_encoder, "config", None), "use_attention_mask", False)
            mask = inp.attention_mask.to(device) if use_mask else None
            emb = self.text_encoder(inp.input_ids.to(device), attention_mask=mask)[0]
            bs, seq, _ = emb.shape
            return emb.repeat(1, num_images_per_prompt, 1).view(bs * num_images_per_prompt, seq, -1), inp.input_ids.shape[-1]

        cond, max_len = enc(prompt, self.tokenizer.model_max_length)
        if not do_classifier_free_guidance:
            return cond
        if negative_prompt is None:
            uncond_tokens = [""] * batch_size
        elif type(prompt) is not type(negative_prompt):
            raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)} != {type(prompt)}.")
        elif isinstance(negative_prompt, str):
            uncond_tokens = [negative_prompt]
        elif batch_size != len(negative_prompt):
            raise ValueError(f"`negative_prompt`: {negative_prompt} has batch size {len(negative_prompt)}, but `prompt`: {prompt} has "
                             f"batch size {batch_size}. Please make sure that passed `negative_prompt` matches the batch size of `prompt`.")
        else:
            uncond_tokens = negative_prompt
        uncond, _ = enc(uncond_tokens, max_len)
        return torch.cat([uncond, cond])

    def check_inputs(self, prompt, height, width, callback_steps):
        if not isinstance(prompt, str) and not isinstance(prompt, list):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if (callback_steps is None) or (not isinstance(callback_steps, int) or callback_steps <= 0):
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps} of type {type(callback_steps)}.")

    def prepare_latents(self, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        shape = (batch_size, num_channels_latents, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                             f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is None:
            if isinstance(generator, list):
                latents = torch.cat([torch.randn((1,) + shape[1:], generator=g, device=g.device, dtype=dtype) for g in generator]).to(device)
            else:
                gdev = generator.device if generator is not None else device
                latents = torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)
        else:
            if latents.shape != shape:
                raise ValueError(f"Unexpected latents shape, got {latents.shape}, expected {shape}")
            latents = latents.to(device)
        return latents * self.scheduler.init_noise_sigma

    def decode_latents(self, latents: torch.Tensor) -> np.ndarray:
        """(N,4,h,w) -> numpy (N,H,W,3) f32 in [0,1] (reference :339-345)"""
        if hasattr(self.vae, "decode_video01"):
            image = self.vae.decode_video01(latents[:, :, None])[:, :, 0]
        else:
            image = (self.vae.decode(latents / 0.18215).sample / 2 + 0.5).clamp(0, 1)
        return image.cpu().permute(0, 2, 3, 1).float().numpy()

    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str]], height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 50, guidance_scale: float = 7.5, negative_prompt: Optional[Union[str, List[str]]] = None,
                 num_images_per_prompt: Optional[int] = 1, eta: float = 0.0, generator=None, latents: Optional[torch.Tensor] = None,
                 output_type: Optional[str] = "pil", return_dict: bool = True,
                 callback: Optional[Callable[[int, int, torch.Tensor], None]] = None, callback_steps: Optional[int] = 1):
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        self.check_inputs(prompt, height, width, callback_steps)
        if eta != 0.0:
            raise NotImplementedError("StableDiffusionPipeline on the MI355X engine: eta > 0 (stochastic DDIM) is not implemented")
        if not hasattr(self.scheduler, "engine_config"):
            raise TypeError("scheduler must be the drop-in diffusers.DDIMScheduler (the update runs as a fused HIP kernel)")
        batch_size = 1 if isinstance(prompt, str) else len(prompt)
        device = self._execution_device
        cfg_on = guidance_scale > 1.0
        text_embeddings = self._encode_prompt(prompt, device, num_images_per_prompt, cfg_on, negative_prompt)
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        latents = self.prepare_latents(batch_size * num_images_per_prompt, self.unet.in_channels, height, width, text_embeddings.dtype,
                                       device, generator, latents)
        sampler = DDIMSampler(self.unet._get_engine(), self.scheduler.engine_config)
        bar = self.progress_bar(total=num_inference_steps)

        def cb(i, t, lat):
            if hasattr(bar, "update"):
                bar.update()
            if callback is not None and i % callback_steps == 0:
                callback(i, t, lat[:, :, 0])

        out = sampler.sample(latents[:, :, None], text_embeddings, num_inference_steps, guidance_scale, callback=cb)[:, :, 0]
        if hasattr(bar, "close"):
            bar.close()
        image = self.decode_latents(out)
        if output_type == "pil":
            image = self.numpy_to_pil(image)
        if not return_dict:
            return (image, None)
        return StableDiffusionPipelineOutput(images=image, nsfw_content_detected=None)
