"""The conditioning-encoder engines on a real MI355X (HIP kernels through the C ABI) against the real implementations' recorded
outputs (transformers CLIP, the reference's ImageProjModel / Resampler)."""
import os

import numpy as np
import pytest
import torch

from followyourclick_amd.engine import encoders as EN
from oracle import encoders as E

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    return {k: torch.from_numpy(v) if v.shape and v.dtype.kind in "fi" else v for k, v in np.load(os.path.join(golden_dir, name)).items()}


def _cfg(cls, ocfg):
    return cls(**vars(ocfg))


def rel(a, b):
    a = a.cpu()
    return ((a - b).norm() / b.norm()).item()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2)])
def test_clip_text_engine(golden_dir, dtype, tol):
    g = _load(golden_dir, "enc_clip_text.npz")
    sd = E.make_encoder_weights(E.clip_text_shapes(E.TINY_TEXT), int(g["weight_seed"]))
    eng = EN.ClipTextEngine(EN.pack_clip_text({"text_model." + k: v for k, v in sd.items()}, _cfg(EN.ClipTextConfig, E.TINY_TEXT), dtype, "cuda"))
    out = eng.encode(g["input_ids"])
    assert out.shape == g["last_hidden_state"].shape and out.dtype == torch.float32
    assert rel(out, g["last_hidden_state"]) < tol
    short = eng.encode(g["input_ids"][:, :20])                 # causal: a shorter prompt is a prefix computation
    assert rel(short, g["last_hidden_state"][:, :20]) < tol
    with pytest.raises(IndexError):
        eng.encode(torch.full((1, 77), 10 ** 6))
    with pytest.raises(ValueError):
        eng.encode(torch.zeros(1, 78, dtype=torch.int64))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2)])
def test_clip_vision_engine(golden_dir, dtype, tol):
    g = _load(golden_dir, "enc_clip_vision.npz")
    sd = E.make_encoder_weights(E.clip_vision_shapes(E.TINY_VISION), int(g["weight_seed"]))
    eng = EN.ClipVisionEngine(EN.pack_clip_vision(sd, _cfg(EN.ClipVisionConfig, E.TINY_VISION), dtype, "cuda"))
    out = eng.encode(g["pixel_values"], want=("image_embeds", "penultimate", "last"))
    for name in ("penultimate", "last", "image_embeds"):
        assert out[name].shape == g[name].shape
        assert rel(out[name], g[name]) < tol, name
    only = eng.encode(g["pixel_values"], want=("penultimate",))
    assert list(only) == ["penultimate"] and torch.equal(only["penultimate"].cpu(), out["penultimate"].cpu())
    with pytest.raises(ValueError):
        eng.encode(torch.zeros(1, 3, 32, 32))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2)])
def test_image_proj_and_resampler_engines(golden_dir, dtype, tol):
    g = _load(golden_dir, "enc_ip_adapter.npz")
    sd_p = E.make_encoder_weights(E.image_proj_shapes(E.TINY_VISION.projection_dim, 64, 4), int(g["proj_seed"]))
    proj = EN.ImageProjEngine(EN.pack_image_proj(sd_p, dtype, "cuda"))
    assert rel(proj.project(g["image_embeds"]), g["proj_tokens"]) < tol
    assert rel(proj.project(torch.zeros_like(g["image_embeds"])), g["proj_tokens_uncond"]) < tol
    sd_r = E.make_encoder_weights(E.resampler_shapes(E.TINY_RESAMPLER), int(g["resampler_seed"]))
    res = EN.ResamplerEngine(EN.pack_resampler(sd_r, _cfg(EN.ResamplerConfig, E.TINY_RESAMPLER), dtype, "cuda"))
    out = res.resample(g["clip_hidden"])
    assert out.shape == g["resampler_tokens"].shape
    assert rel(out, g["resampler_tokens"]) < tol
