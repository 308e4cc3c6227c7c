"""DDIM tables on the host (reference diffusers/schedulers/scheduling_ddim.py, patched with
zero-terminal-SNR rescaling and v-prediction).

The reference indexes a CPU `alphas_cumprod` table with a device tensor inside `step()` (a GPU->CPU sync
every step, scheduling_ddim.py:308-312).  Here everything that depends only on (t, n) is precomputed in
f32 with the same op sequence as the reference (bit-identical table), and each step's four coefficients
{sqrt(abar_t), sqrt(1-abar_t), sqrt(abar_prev), sqrt(1-abar_prev)} are uploaded once per clip; the
update itself is the fused guidance+DDIM kernel (fyc_cfg_ddim_step).
"""
from __future__ import annotations

from typing import List

import torch

from .config import DDIMConfig

PRED_TYPES = {"epsilon": 0, "v_prediction": 1, "sample": 2}


def rescale_zero_terminal_snr(betas: torch.Tensor) -> torch.Tensor:
    """Algorithm 1 of arXiv:2305.08891 as the reference applies it (scheduling_ddim.py:78-111)."""
    abar_sqrt = torch.cumprod(1.0 - betas, dim=0).sqrt()
    a0, aT = abar_sqrt[0].clone(), abar_sqrt[-1].clone()
    abar_sqrt = abar_sqrt - aT
    abar_sqrt = abar_sqrt * (a0 / (a0 - aT))
    abar = abar_sqrt ** 2
    alphas = torch.cat([abar[0:1], abar[1:] / abar[:-1]])
    return 1 - alphas


class DDIMTables:
    def __init__(self, cfg: DDIMConfig):
        self.cfg = cfg
        n = cfg.num_train_timesteps
        if cfg.trained_betas is not None:
            betas = torch.tensor(cfg.trained_betas, dtype=torch.float32)
        elif cfg.beta_schedule == "linear":
            betas = torch.linspace(cfg.beta_start, cfg.beta_end, n, dtype=torch.float32)
        elif cfg.beta_schedule == "scaled_linear":
            betas = torch.linspace(cfg.beta_start ** 0.5, cfg.beta_end ** 0.5, n, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"{cfg.beta_schedule} is not implemented for DDIMTables")
        if cfg.rescale_betas_zero_snr:
            betas = rescale_zero_terminal_snr(betas)
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if cfg.set_alpha_to_one else self.alphas_cumprod[0]
        if cfg.prediction_type not in PRED_TYPES:
            raise ValueError(f"prediction_type given as {cfg.prediction_type} must be one of `epsilon`, `sample`, or `v_prediction`")
        self.pred_type = PRED_TYPES[cfg.prediction_type]

    def timesteps(self, num_inference_steps: int) -> torch.Tensor:
        """set_timesteps (scheduling_ddim.py:238-252): arange(n) * (T // n), reversed, + steps_offset."""
        ratio = self.cfg.num_train_timesteps // num_inference_steps
        return (torch.arange(0, num_inference_steps, dtype=torch.int64) * ratio).flip(0) + self.cfg.steps_offset

    def step_coefficients(self, t: int, num_inference_steps: int, eta: float = 0.0) -> List[float]:
        """{sqrt(abar_t), sqrt(1 - abar_t), sqrt(abar_prev), sqrt(1 - abar_prev - sigma^2), sigma} with sigma = eta * sqrt(variance_t)
        (scheduling_ddim.py:229-236 `_get_variance`, :336-349): the same f32 tensor arithmetic as the reference, eta = 0 gives sigma = 0"""
        prev_t = t - self.cfg.num_train_timesteps // num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        variance = ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)
        std = eta * variance ** 0.5
        return [float(a_t ** 0.5), float((1 - a_t) ** 0.5), float(a_prev ** 0.5), float((1 - a_prev - std ** 2) ** 0.5), float(std)]

    def coefficient_table(self, num_inference_steps: int, eta: float = 0.0) -> torch.Tensor:
        ts = self.timesteps(num_inference_steps).tolist()
        return torch.tensor([self.step_coefficients(t, num_inference_steps, eta) for t in ts], dtype=torch.float32)
