"""`diffusers.DDIMScheduler` surface (reference diffusers/schedulers/scheduling_ddim.py:134-376) on the
engine's host-side tables.  AnimationPipeline does NOT call `step()` in its loop - guidance + the DDIM
update are one fused kernel there - but the method is kept for scripts that drive a scheduler by hand."""
from __future__ import annotations

import inspect
import json
import os
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional, Union

import torch

from followyourclick_amd.engine import DDIMConfig
from followyourclick_amd.engine.scheduler import DDIMTables


@dataclass
class DDIMSchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: Optional[torch.Tensor] = None


class DDIMScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02,
                 beta_schedule: str = "linear", trained_betas=None, clip_sample: bool = True, set_alpha_to_one: bool = True,
                 steps_offset: int = 0, prediction_type: str = "epsilon", rescale_betas_zero_snr: bool = False, **kwargs):
        self.engine_config = DDIMConfig(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                        beta_schedule=beta_schedule, trained_betas=trained_betas, clip_sample=clip_sample,
                                        set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset, prediction_type=prediction_type,
                                        rescale_betas_zero_snr=rescale_betas_zero_snr)
        self.config = SimpleNamespace(**vars(self.engine_config))
        self.tables = DDIMTables(self.engine_config)
        self.betas, self.alphas, self.alphas_cumprod = self.tables.betas, self.tables.alphas, self.tables.alphas_cumprod
        self.final_alpha_cumprod = self.tables.final_alpha_cumprod
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1, dtype=torch.int64)

    @classmethod
    def from_config(cls, config: dict, **kwargs):
        """Keeps the keys this scheduler's constructor knows and drops the rest (a checkpoint's scheduler_config.json
        usually belongs to another scheduler class, e.g. PNDM's `skip_prk_steps`) - reference configuration_utils.py:189-226."""
        known = set(inspect.signature(cls.__init__).parameters) - {"self", "kwargs"}
        cfg = {k: v for k, v in dict(config).items() if k in known}
        cfg.update({k: v for k, v in kwargs.items() if k in known})
        return cls(**cfg)

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, **kwargs):
        """`DDIMScheduler.from_pretrained(path, subfolder="scheduler")` (scripts/inference.py:198): local directory only."""
        if subfolder is not None:
            pretrained_model_path = os.path.join(pretrained_model_path, subfolder)
        config_file = os.path.join(pretrained_model_path, "scheduler_config.json")
        if not os.path.isfile(config_file):
            raise EnvironmentError(f"Error no file named scheduler_config.json found in directory {pretrained_model_path}.")
        with open(config_file) as f:
            return cls.from_config(json.load(f), **kwargs)

    def scale_model_input(self, sample: torch.Tensor, timestep=None) -> torch.Tensor:
        return sample

    def set_timesteps(self, num_inference_steps: int, device: Union[str, torch.device, None] = None) -> None:
        self.num_inference_steps = num_inference_steps
        self.timesteps = self.tables.timesteps(num_inference_steps).to(device)

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, eta: float = 0.0, use_clipped_model_output: bool = False,
             generator=None, variance_noise=None, return_dict: bool = True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        sa, sb, sap, sbp, sigma = self.tables.step_coefficients(int(timestep), self.num_inference_steps, float(eta))
        if self.tables.pred_type == 1:
            x0, eps = sa * sample - sb * model_output, sa * model_output + sb * sample
        elif self.tables.pred_type == 0:
            x0, eps = (sample - sb * model_output) / sa, model_output
        else:
            x0, eps = model_output, model_output
        if self.config.clip_sample:
            x0 = x0.clamp(-1, 1)
        if use_clipped_model_output:                     # reference scheduling_ddim.py:342-344
            eps = (sample - sa * x0) / sb
        prev = sap * x0 + sbp * eps
        if eta > 0:                                      # :346-363
            if variance_noise is not None and generator is not None:
                raise ValueError("Cannot pass both generator and variance_noise. Please make sure that either `generator` or"
                                 " `variance_noise` stays `None`.")
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype)
            prev = prev + sigma * variance_noise
        if not return_dict:
            return (prev,)
        return DDIMSchedulerOutput(prev_sample=prev, pred_original_sample=x0)
