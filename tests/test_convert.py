"""Checkpoint / LoRA ingestion against fixtures produced by the REAL reference's converters
(oracle/make_golden_convert.py): every LDM key must land on the diffusers name the reference gives it, and a LoRA merge
must change the same weights by the same amounts."""
import json
import os
import sys

import numpy as np
import pytest
import torch

import followyourclick_amd
from oracle import functional as Fn
from oracle import weights as W


@pytest.fixture(scope="module")
def dropin():
    followyourclick_amd.install_dropin(force=True)
    yield
    for name in [k for k in sys.modules if k.split(".")[0] in ("animatediff", "diffusers", "ip_adapter")]:
        del sys.modules[name]
    if followyourclick_amd.DROPIN_DIR in sys.path:
        sys.path.remove(followyourclick_amd.DROPIN_DIR)


@pytest.fixture(scope="module")
def keymap(golden_dir):
    with open(os.path.join(golden_dir, "convert_keymap.json")) as f:
        return json.load(f)


def _tagged(ldm_keys, shapes=None):
    return {k: torch.full(tuple((shapes or {}).get(k, (2,))), float(i)) for i, k in enumerate(ldm_keys)}


def _map(src, out):
    by_tag = {float(v.flatten()[0]): k for k, v in src.items()}
    return sorted([by_tag[float(v.flatten()[0])], k] for k, v in out.items())


@pytest.mark.parametrize("concat", [False, True])
def test_unet_keys_match_reference(dropin, keymap, golden_dir, concat):
    from animatediff.utils.convert_from_ckpt import convert_ldm_unet_checkpoint
    ref = keymap["unet_img_embed_concat" if concat else "unet"]
    # the synthetic checkpoint also carries a foreign (text-encoder) key, as real single-file checkpoints do
    src = _tagged([r[0] for r in keymap["unet"]] + ["cond_stage_model.transformer.text_model.final_layer_norm.weight"])
    before = dict(src)
    from types import SimpleNamespace
    for config in ({"layers_per_block": 2, "class_embed_type": None}, SimpleNamespace(layers_per_block=2, class_embed_type=None)):
        out = convert_ldm_unet_checkpoint(src, config, need_img_embed_concat=concat)
        assert _map(src, out) == [r[:2] for r in ref]
    assert src.keys() == before.keys()                                     # input not consumed
    assert ("conv_in.weight" in out) == (not concat)
    # the converted key set is exactly the spatial (2-D) UNet's state dict
    with open(os.path.join(golden_dir, "schema_unet2d_tiny.json")) as f:
        want = set(json.load(f))
    assert set(out) | ({"conv_in.weight", "conv_in.bias"} if concat else set()) == want
    with pytest.raises(KeyError):
        convert_ldm_unet_checkpoint({"first_stage_model.x": torch.zeros(1)}, {"layers_per_block": 2, "class_embed_type": None})


def test_unet_ema_extraction(dropin, keymap):
    from animatediff.utils.convert_from_ckpt import convert_ldm_unet_checkpoint
    keys = [r[0] for r in keymap["unet"]]
    src = _tagged(keys)
    ema = {"model_ema." + "".join(k.split(".")[1:]): v + 10000 for k, v in src.items()}
    cfg = {"layers_per_block": 2, "class_embed_type": None}
    plain = convert_ldm_unet_checkpoint({**src, **ema}, cfg)
    picked = convert_ldm_unet_checkpoint({**src, **ema}, cfg, extract_ema=True)
    assert all(float(picked[k][0]) == float(plain[k][0]) + 10000 for k in plain)


def test_vae_keys_match_reference(dropin, keymap):
    from animatediff.utils.convert_from_ckpt import convert_ldm_vae_checkpoint
    ref = keymap["vae"]
    src = _tagged(list(keymap["vae_src_shapes"]), keymap["vae_src_shapes"])
    out = convert_ldm_vae_checkpoint(src, {})
    assert _map(src, out) == [r[:2] for r in ref]
    for ldm, name, shape in ref:
        mine = list(out[name].shape)
        if name.endswith("proj_attn.weight"):
            assert mine == shape[:2]          # reference leaves (C,C,1) here (its `[:, :, 0]` on a 4-D conv weight); Linear wants (C,C)
        else:
            assert mine == shape, name
    # what comes out loads into the drop-in VAE's schema names
    from followyourclick_amd.engine.schema import vae_decoder_schema, vae_encoder_schema
    from followyourclick_amd.engine import VAEDecoderConfig
    names = set(vae_decoder_schema(VAEDecoderConfig())) | set(vae_encoder_schema(VAEDecoderConfig()))
    assert names == set(out)


def test_clip_text_keys(dropin):
    from animatediff.utils.convert_from_ckpt import convert_ldm_clip_checkpoint
    sd = convert_ldm_clip_checkpoint({"cond_stage_model.transformer.text_model.final_layer_norm.weight": torch.ones(3),
                                      "model.diffusion_model.out.0.weight": torch.zeros(1)})
    assert list(sd) == ["text_model.final_layer_norm.weight"]


def test_lora_merges_match_reference(dropin, golden_dir):
    from animatediff.models.unet import UNet3DConditionModel
    from animatediff.utils.convert_lora_safetensor_to_diffusers import convert_lora, convert_motion_lora_ckpt_to_diffusers
    from test_dropin_api import TINY
    g = np.load(os.path.join(golden_dir, "convert_lora.npz"))
    unet = UNet3DConditionModel(**TINY)
    sd0 = W.make_weights(W.unet_state_shapes(Fn.tiny_unet_config()), seed=0)
    unet.load_state_dict(sd0)
    unet._engine_key = "packed"
    lora = {k[len("lora/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("lora/")}
    motion = {k[len("motion/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("motion/")}

    class Pipe:
        pass

    pipe = Pipe()
    pipe.unet, pipe.text_encoder = unet, None
    assert convert_lora(pipe, lora, alpha=0.8) is pipe
    assert unet._engine_key is None                      # packed engine weights are invalidated
    convert_motion_lora_ckpt_to_diffusers(pipe, motion, alpha=0.5)
    sd1 = unet.state_dict()
    changed = sorted(k for k in sd0 if not torch.equal(sd0[k], sd1[k]))
    assert changed == sorted(g["changed"].tolist())
    for k in changed:
        assert torch.allclose(sd1[k], torch.from_numpy(g["after/" + k]), atol=1e-6), k
    with pytest.raises(KeyError):
        convert_lora(pipe, {"lora_unet_no_such_layer.lora_down.weight": torch.zeros(1, 1), "lora_unet_no_such_layer.lora_up.weight": torch.zeros(1, 1)})


def test_save_videos_grid(dropin, tmp_path):
    from PIL import Image
    from animatediff.utils.util import make_grid, save_videos_grid
    x = torch.arange(5 * 3 * 4 * 6, dtype=torch.float32).reshape(5, 3, 4, 6)
    g = make_grid(x, nrow=3)                                     # torchvision layout: 2 rows x 3 columns, 2-px zero border
    assert g.shape == (3, 2 * (4 + 2) + 2, 3 * (6 + 2) + 2)
    assert torch.equal(g[:, 2:6, 2:8], x[0]) and torch.equal(g[:, 2:6, 10:16], x[1]) and torch.equal(g[:, 8:12, 10:16], x[4])
    assert float(g[:, :2].abs().sum()) == 0 and float(g[:, 8:12, 18:24].abs().sum()) == 0      # padding and the empty sixth cell
    assert torch.equal(make_grid(x[:1]), x[0])                  # a single image comes back unpadded
    assert make_grid(x[:2, :1]).shape[0] == 3                   # grey -> 3 channels
    vid = torch.rand(2, 3, 4, 8, 8)
    path = str(tmp_path / "out" / "sample.gif")
    save_videos_grid(vid, path, n_rows=6, fps=8)
    im = Image.open(path)
    assert im.n_frames == 4 and im.size == (2 * 10 + 2, 8 + 4)


def test_load_weights(dropin, golden_dir, tmp_path):
    """`animatediff.utils.util.load_weights` (reference animatediff/utils/util.py:91-154): only the `motion_modules.` tensors of the
    motion-module checkpoint are taken (with or without the `state_dict` wrapper), unknown keys among them are an error, motion LoRAs are
    merged one by one with their own alpha - the same merge `test_lora_merges_match_reference` pins against the reference's output."""
    from animatediff.models.unet import UNet3DConditionModel
    from animatediff.utils.util import load_weights
    from test_dropin_api import TINY
    g = np.load(os.path.join(golden_dir, "convert_lora.npz"))
    unet = UNet3DConditionModel(**TINY)
    sd0 = W.make_weights(W.unet_state_shapes(Fn.tiny_unet_config()), seed=0)
    unet.load_state_dict(sd0)

    class Pipe:
        pass

    pipe = Pipe()
    pipe.unet, pipe.text_encoder, pipe.vae = unet, None, None
    temporal = {k: v + 1.0 for k, v in sd0.items() if "motion_modules." in k and v.is_floating_point()}
    spatial_key = next(k for k in sd0 if "motion_modules." not in k and k.endswith("weight"))
    ckpt = str(tmp_path / "mm.ckpt")
    torch.save({"state_dict": dict(temporal, **{spatial_key: sd0[spatial_key] + 5.0})}, ckpt)     # the spatial tensor must be ignored
    motion = {k[len("motion/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("motion/")}
    lora_path = str(tmp_path / "motion_lora.ckpt")
    torch.save(motion, lora_path)
    assert load_weights(pipe, motion_module_path=ckpt, motion_module_lora_configs=[dict(path=lora_path, alpha=0.5)]) is pipe
    sd1 = unet.state_dict()
    assert torch.equal(sd1[spatial_key], sd0[spatial_key])
    merged = {k for k in g["changed"].tolist() if "motion_modules." in k}
    assert merged
    for k, v in temporal.items():
        if k in merged:
            delta = torch.from_numpy(g["after/" + k]) - _before_motion_merge(g, sd0, k)
            assert torch.allclose(sd1[k], v + delta, atol=1e-5), k
        else:
            assert torch.equal(sd1[k], v), k
    bad = str(tmp_path / "bad.ckpt")
    torch.save({"down_blocks.0.motion_modules.0.no_such_tensor": torch.zeros(1)}, bad)
    with pytest.raises(AssertionError):
        load_weights(pipe, motion_module_path=bad)
    with pytest.raises(ValueError):
        load_weights(pipe, dreambooth_model_path=str(tmp_path / "model.bin"))


def _before_motion_merge(g, sd0, k):
    """the golden's `after/<k>` holds W0 + motion-LoRA delta for temporal keys (the kohya LoRA of that fixture touches spatial layers only)"""
    return sd0[k]
