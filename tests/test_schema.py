"""The engine's state-dict schema must equal what the reference modules register (golden JSON dumped
from the real reference by oracle/make_golden.py)."""
import json
import os

from followyourclick_amd.engine import UNet3DConfig, VAEDecoderConfig
from followyourclick_amd.engine.schema import random_state_dict, unet_schema, vae_decoder_schema


def _j(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return [(k, tuple(v)) for k, v in json.load(f).items()]


def test_unet_schema_full(golden_dir):
    mine = dict(unet_schema(UNet3DConfig()))
    ref = dict(_j(golden_dir, "schema_unet_full.json"))
    assert len(mine) == 1254
    assert mine == ref          # same names and shapes (load_state_dict is order-independent)


def test_unet_schema_ip(golden_dir):
    assert dict(unet_schema(UNet3DConfig(use_ip_cross_attention=True, ip_num_tokens=16))) == dict(_j(golden_dir, "schema_unet_full_ip.json"))


def test_vae_schema(golden_dir):
    assert dict(vae_decoder_schema(VAEDecoderConfig())) == dict(_j(golden_dir, "schema_vae.json"))


def test_random_state_dict_is_seed_deterministic():
    cfg = UNet3DConfig(block_out_channels=(64, 128, 256, 256), cross_attention_dim=64)
    a, b = random_state_dict(unet_schema(cfg), 5), random_state_dict(unet_schema(cfg), 5)
    assert all((a[k] == b[k]).all() for k in a)
    assert abs(a["mid_block.attentions.0.proj_in.weight"].std().item() - 1 / 16) < 5e-3
