"""oracle/encoders.py (CPU restatement of the CLIP text / vision encoders, ImageProjModel and Resampler) against outputs of the
real implementations recorded in tests/golden/enc_*.npz (transformers 5.15.0 CLIP; the reference's ip_adapter classes)."""
import os

import numpy as np
import torch

from oracle import encoders as E


def _load(golden_dir, name):
    return {k: torch.from_numpy(v) if v.shape and v.dtype.kind in "fi" else v for k, v in np.load(os.path.join(golden_dir, name)).items()}


def test_clip_text(golden_dir):
    g = _load(golden_dir, "enc_clip_text.npz")
    sd = E.make_encoder_weights(E.clip_text_shapes(E.TINY_TEXT), int(g["weight_seed"]))
    out = E.clip_text_forward(sd, E.TINY_TEXT, g["input_ids"])
    assert torch.allclose(out, g["last_hidden_state"], atol=2e-5, rtol=1e-4), (out - g["last_hidden_state"]).abs().max()
    # transformers-4 style checkpoints carry a `text_model.` prefix
    out2 = E.clip_text_forward({"text_model." + k: v for k, v in sd.items()}, E.TINY_TEXT, g["input_ids"])
    assert torch.equal(out, out2)
    # causal: a token's state must not depend on later tokens
    ids = g["input_ids"].clone()
    ids[:, 40:] = 5
    assert torch.allclose(E.clip_text_forward(sd, E.TINY_TEXT, ids)[:, :40], out[:, :40], atol=1e-6)


def test_clip_vision(golden_dir):
    g = _load(golden_dir, "enc_clip_vision.npz")
    sd = E.make_encoder_weights(E.clip_vision_shapes(E.TINY_VISION), int(g["weight_seed"]))
    hs, emb = E.clip_vision_forward(sd, E.TINY_VISION, g["pixel_values"])
    assert len(hs) == int(g["n_hidden_states"])
    for mine, ref in ((hs[-2], g["penultimate"]), (hs[-1], g["last"]), (emb, g["image_embeds"])):
        assert torch.allclose(mine, ref, atol=5e-5, rtol=1e-4), (mine - ref).abs().max()


def test_image_proj_and_resampler(golden_dir):
    g = _load(golden_dir, "enc_ip_adapter.npz")
    sd_p = E.make_encoder_weights(E.image_proj_shapes(E.TINY_VISION.projection_dim, 64, 4), int(g["proj_seed"]))
    assert torch.allclose(E.image_proj_forward(sd_p, g["image_embeds"], 4, 64), g["proj_tokens"], atol=2e-5)
    assert torch.allclose(E.image_proj_forward(sd_p, torch.zeros_like(g["image_embeds"]), 4, 64), g["proj_tokens_uncond"], atol=2e-5)
    sd_r = E.make_encoder_weights(E.resampler_shapes(E.TINY_RESAMPLER), int(g["resampler_seed"]))
    out = E.resampler_forward(sd_r, E.TINY_RESAMPLER, g["clip_hidden"])
    assert out.shape == (2, 4, 64)
    assert torch.allclose(out, g["resampler_tokens"], atol=2e-5, rtol=1e-4), (out - g["resampler_tokens"]).abs().max()
