"""`animatediff.pipelines.pipeline_animation.AnimationPipeline` on the MI355X engine.

Same constructor, attributes, `__call__` signature, error behaviour and output type as the reference
(reference animatediff/pipelines/pipeline_animation.py:41-130, 400-445, 546-788), so
`scripts/inference*.py` can drive it unchanged.  What differs is where the work happens:

  * two conditioning modes: the FollowYourClick one (`use_first_frame_mask_condition_concat=True`: 9-channel input of latents,
    click mask and first-frame latents, scripts/inference.py:374-395) and plain text-to-video on a 4-channel UNet (how
    scripts/inference_org.py:265-289 / animate.py call it);
  * prompt encoding stays with the caller's tokenizer / text encoder (or followyourclick_amd.encoders.ClipTextHip);
  * the DDIM loop does not go module-by-module through torch: per step it is ONE input-assembly kernel,
    the engine's UNet3D op schedule on the CFG pair, and ONE fused guidance + DDIM-update kernel; all
    text/IP K,V projections, time embeddings and scheduler coefficients are prepared once per clip;
  * `decode_latents` batches all frames through the VAE decoder engine (the reference loops over frames).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Union

import numpy as np
import torch

from followyourclick_amd.engine.sampler import DDIMSampler


@dataclass
class AnimationPipelineOutput:
    videos: Union[torch.Tensor, np.ndarray]


class AnimationPipeline:
    _optional_components: list = []

    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler, image_encoder=None, text_encoder_2=None, tokenizer_2=None,
                 ip_adapter=None):
        if getattr(scheduler.config, "steps_offset", 1) != 1:
            scheduler.config.steps_offset = 1            # the reference patches outdated configs the same way (:65-77)
            if hasattr(scheduler, "engine_config"):
                scheduler.engine_config.steps_offset = 1
        if getattr(scheduler.config, "clip_sample", False) is True:
            scheduler.config.clip_sample = False         # (:79-90)
            if hasattr(scheduler, "engine_config"):
                scheduler.engine_config.clip_sample = False
        self.vae, self.text_encoder, self.tokenizer, self.unet, self.scheduler = vae, text_encoder, tokenizer, unet, scheduler
        self.image_encoder, self.text_encoder_2, self.tokenizer_2, self.ip_adapter = image_encoder, text_encoder_2, tokenizer_2, ip_adapter
        self.vae_scale_factor = 2 ** (len(self.vae.config.block_out_channels) - 1)
        self._device = torch.device("cpu")

    # ---- DiffusionPipeline-like plumbing ---------------------------------------------------------
    def to(self, device):
        self._device = torch.device(device)
        for m in (self.vae, self.text_encoder, self.unet, self.image_encoder, self.text_encoder_2):
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

    # ---- conditioning front-end (caller's modules) ----------------------------------------------------
    def _encode_prompt(self, prompt, device, num_videos_per_prompt, do_classifier_free_guidance, negative_prompt):
        """cat[uncond, cond] text states, (2B, 77, D) - reference :158-245, :397"""
        batch_size = len(prompt) if isinstance(prompt, list) else 1

        def enc(texts, max_length):
            inp = self.tokenizer(texts, padding="max_length", max_length=max_length, truncation=True, return_tensors="pt")
            use_mask = getattr(getattr(self.text_encoder, "config", None), "use_attention_mask", False)
            mask = inp.attention_mask.to(device) if use_mask else None
            emb = self.text_encoder(inp.input_ids.to(device), attention_mask=mask)[0]
            bs, seq, _ = emb.shape
            return emb.repeat(1, num_videos_per_prompt, 1).view(bs * num_videos_per_prompt, seq, -1), inp.input_ids.shape[-1]

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

    def prepare_latents(self, batch_size, num_channels_latents, video_length, height, width, dtype, device, generator, latents=None,
                        init_latents=None, init_image=None, use_residual_noise=False, base_lambda=0.9, k=64, use_interpolate_noise=True,
                        use_add_noise=False, first_images_mask=None):
        """reference :448-537, pinned by tests/golden/prepare_latents.npz.  `init_image` (dead there: it needs PIL/transforms names
        the reference never imports, :464) is rejected; `use_add_noise` is unused by the reference as well."""
        shape = (batch_size, num_channels_latents, video_length, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                             f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if init_image is not None and init_latents is None:
            raise NotImplementedError("prepare_latents(init_image=...): pass init_latents (the reference's image branch cannot run, :464)")

        def blend(lat):       # first image blended into every frame with a decaying weight (:501-508, :526-532)
            lat = lat.clone()
            for i in range(video_length):
                a = (video_length - float(i)) / video_length / k
                lat[:, :, i] = init_latents.to(lat) * a + lat[:, :, i] * (1 - a)
            return lat

        if latents is None:
            if isinstance(generator, list):
                latents = torch.cat([torch.randn(shape, generator=g, device=g.device if hasattr(g, "device") else device, dtype=dtype)
                                     for g in generator], dim=0).to(device)
            else:
                gdev = generator.device if generator is not None else device
                latents = torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)
                if use_interpolate_noise:
                    latents = latents[:, :, :1].repeat(1, 1, video_length, 1, 1)
                if init_latents is not None:
                    if first_images_mask is None:     # the reference indexes the mask in this branch (:505) although it never uses it
                        raise TypeError("'NoneType' object is not subscriptable")
                    latents = blend(latents)
                if use_residual_noise:                # :509-513
                    base = latents[:, :, 0].unsqueeze(2).repeat(1, 1, video_length, 1, 1)
                    latents = base_lambda ** 0.5 * base + (1 - base_lambda) ** 0.5 * latents
                    latents[:, :, 0] = base[:, :, 0]
        else:
            if latents.shape != shape:
                raise ValueError(f"Unexpected latents shape, got {latents.shape}, expected {shape}")
            latents = latents.to(device)
            if init_latents is not None:
                latents = blend(latents)
        return latents * self.scheduler.init_noise_sigma

    def decode_latents(self, latents):
        """(b,4,f,h,w) -> numpy (b,3,f,H,W) f32 in [0,1] (reference :400-413)"""
        if hasattr(self.vae, "decode_video01"):
            video = self.vae.decode_video01(latents)
        else:  # a foreign VAE object: follow the reference's per-frame protocol
            b, c, f, h, w = latents.shape
            z = (latents / 0.18215).permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
            frames = torch.cat([self.vae.decode(z[i:i + 1]).sample for i in range(b * f)])
            video = (frames.reshape(b, f, *frames.shape[1:]).permute(0, 2, 1, 3, 4) / 2 + 0.5).clamp(0, 1)
        return video.cpu().float().numpy()

    # ---- sampling -----------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str]], video_length: Optional[int], height: Optional[int] = None,
                 width: Optional[int] = None, num_inference_steps: int = 50, guidance_scale: float = 7.5,
                 negative_prompt: Optional[Union[str, List[str]]] = None, num_videos_per_prompt: Optional[int] = 1, eta: float = 0.0,
                 generator=None, latents: Optional[torch.Tensor] = None, output_type: Optional[str] = "tensor", return_dict: bool = True,
                 callback: Optional[Callable[[int, int, torch.Tensor], None]] = None, callback_steps: Optional[int] = 1,
                 use_first_frame_condition: bool = False, use_first_frame_condition_concat: bool = False,
                 use_first_frame_mask_condition_concat: bool = False, use_first_frame_mask_condition_concat_image_partial_mask=None,
                 first_image_latents=None, use_first_image_as_init_latents=False, video_scale=0, use_ip_cross_attention=False,
                 condition_images=None, use_uncond_images=False, use_camera_motion_condition=False, camera_movement_type=None,
                 use_text_encoder_2=False, use_uncond_text_2=False, use_fps_condition=False, fps_tensor=None,
                 use_interpolate_noise=False, first_images_mask=None, flow_control=None, **kwargs):
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        self.check_inputs(prompt, height, width, callback_steps)
        unsupported = dict(use_text_encoder_2=use_text_encoder_2)
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError(f"AnimationPipeline on the MI355X engine: {bad} not implemented (SURVEY.md 8 scope)")
        concat_model = bool(getattr(self.unet.engine_config, "use_first_frame_mask_condition_concat", False))
        if use_first_frame_mask_condition_concat != concat_model:
            raise ValueError(f"use_first_frame_mask_condition_concat={use_first_frame_mask_condition_concat} but the UNet was built "
                             f"with use_first_frame_mask_condition_concat={concat_model} ({self.unet.engine_config.conv_in_channels} input channels)")
        concat2_model = bool(getattr(self.unet.engine_config, "use_first_frame_condition_concat", False))
        if bool(use_first_frame_condition_concat) != concat2_model:
            raise ValueError(f"use_first_frame_condition_concat={use_first_frame_condition_concat} but the UNet was built with "
                             f"use_first_frame_condition_concat={concat2_model} ({self.unet.engine_config.conv_in_channels} input channels)")
        if use_camera_motion_condition and not getattr(self.unet.engine_config, "use_camera_motion_condition", False):
            raise ValueError("use_camera_motion_condition=True needs a UNet built with use_camera_motion_condition")
        if use_camera_motion_condition and camera_movement_type is None:
            raise ValueError("camera_movement_type is required with use_camera_motion_condition")
        if (use_first_frame_mask_condition_concat or use_first_frame_condition or use_first_frame_condition_concat) and first_image_latents is None:
            raise ValueError("first_image_latents is required with use_first_frame_mask_condition_concat / use_first_frame_condition(_concat)")
        if use_first_frame_condition and use_first_frame_mask_condition_concat:
            raise ValueError("use_first_frame_condition and use_first_frame_mask_condition_concat are alternatives (reference :691-693)")
        if use_first_frame_mask_condition_concat_image_partial_mask is not None and first_image_latents is not None:
            # the first-frame block is multiplied by the partial mask before the concat (reference :698-699); the block is
            # constant over the loop, so the product is taken once here
            first_image_latents = first_image_latents * torch.as_tensor(use_first_frame_mask_condition_concat_image_partial_mask).to(first_image_latents)

        batch_size = 1
        if latents is not None:
            batch_size = latents.shape[0]
        if isinstance(prompt, list):
            batch_size = len(prompt)
        device = self._execution_device
        cfg_on = guidance_scale > 1.0
        prompt = prompt if isinstance(prompt, list) else [prompt] * batch_size
        if negative_prompt is not None:
            negative_prompt = negative_prompt if isinstance(negative_prompt, list) else [negative_prompt] * batch_size
        text_embeddings = self._encode_prompt(prompt, device, num_videos_per_prompt, cfg_on, negative_prompt)

        self.scheduler.set_timesteps(num_inference_steps, device=device)
        mask_final = None
        if first_images_mask is not None:
            mask_final = torch.clamp(first_images_mask[:, :, 0:1].repeat(1, 1, video_length, 1, 1), 0, 1)   # :632-635

        latents = self.prepare_latents(batch_size * num_videos_per_prompt, self.unet.in_channels, video_length, height, width,
                                       text_embeddings.dtype, device, generator, latents,
                                       init_latents=first_image_latents if use_first_image_as_init_latents else None,
                                       use_interpolate_noise=use_interpolate_noise,
                                       first_images_mask=mask_final if use_first_image_as_init_latents else None)   # :636-668

        ip_tokens = None
        if use_ip_cross_attention:
            cond, uncond = self.ip_adapter.get_image_clip_feat(input_image=condition_images)       # :678
            if use_uncond_images:
                cond = uncond.clone()
            feats = torch.cat([uncond, cond]) if cfg_on else cond
            if self.unet.image_proj_model is None:
                raise ValueError("unet.image_proj_model is not set")
            ip_tokens = self.unet.image_proj_model(feats)

        def as_list(v):
            return None if v is None else torch.as_tensor(v).reshape(-1).float().cpu().tolist()

        engine = self.unet._get_engine()
        sampler = DDIMSampler(engine, self.scheduler.engine_config)
        bar = self.progress_bar(total=num_inference_steps)

        def cb(i, t, lat):
            if hasattr(bar, "update"):
                bar.update()
            if callback is not None and i % callback_steps == 0:
                callback(i, t, lat)

        latents_out = sampler.sample(latents, text_embeddings, num_inference_steps, guidance_scale,
                                     first_image_latents=first_image_latents, first_images_mask=mask_final,
                                     fps=as_list(fps_tensor) if use_fps_condition else None,
                                     flow=as_list(flow_control) if use_fps_condition else None, ip_tokens=ip_tokens, callback=cb,
                                     video_scale=float(video_scale or 0.0), first_frame_condition=bool(use_first_frame_condition),
                                     eta=float(eta), generator=generator,            # stochastic DDIM (reference :672, scheduling_ddim.py:336-365)
                                     camera=as_list(camera_movement_type) if use_camera_motion_condition else None)
        if hasattr(bar, "close"):
            bar.close()

        video = self.decode_latents(latents_out)
        if output_type == "tensor":
            video = torch.from_numpy(video)
        if not return_dict:
            return video
        return AnimationPipelineOutput(videos=video)
