"""`animatediff.utils.convert_from_ckpt`: CompVis/LDM single-file checkpoints -> the diffusers-style state dicts the
drop-in modules load (DreamBooth / base-model ingestion of scripts/inference.py:221-243).

Host-side name surgery only (no tensor math apart from the VAE attention conv1x1 -> linear reshape).  Behaviour follows
reference animatediff/utils/convert_from_ckpt.py:328-560 (UNet), :560-664 (VAE), :717-728 (CLIP text) - same arguments,
same output keys - but is written as per-key translation rules instead of the reference's list-rewriting passes.
Differences, on purpose:
  * the input `checkpoint` dict is not mutated (the reference pops the UNet keys out of it);
  * VAE `proj_attn.weight` always comes out 2-D (out, in): the reference's `[:, :, 0]` leaves conv2d-style (C,C,1,1)
    weights 3-D, which `load_state_dict` then rejects;
  * `convert_ldm_clip_checkpoint` takes the model to fill as an argument (the reference hard-codes an absolute path).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

_RESNET = (("in_layers.0", "norm1"), ("in_layers.2", "conv1"), ("emb_layers.1", "time_emb_proj"), ("out_layers.0", "norm2"),
           ("out_layers.3", "conv2"), ("skip_connection", "conv_shortcut"))


def _cfg(config, name):
    return config[name] if isinstance(config, dict) or hasattr(config, "__getitem__") else getattr(config, name)


def _resnet_tail(rest: str) -> str:
    for old, new in _RESNET:
        if rest.startswith(old + "."):
            return new + rest[len(old):]
    raise KeyError(f"unknown resnet parameter '{rest}'")


def _unet_key(key: str, layers_per_block: int) -> Optional[str]:
    """one LDM UNet key (without the `model.diffusion_model.` prefix) -> diffusers name; None for keys the reference drops"""
    per = layers_per_block + 1
    p = key.split(".")
    head = p[0]
    if head == "time_embed":
        return f"time_embedding.linear_{1 if p[1] == '0' else 2}.{p[2]}"
    if head == "label_emb":                                     # label_emb.0.{0,2}.* (class_embed_type timestep/projection)
        return f"class_embedding.linear_{1 if p[2] == '0' else 2}.{p[3]}"
    if head == "out":
        return {"0": "conv_norm_out", "2": "conv_out"}[p[1]] + "." + p[2]
    if head == "input_blocks":
        i, sub, rest = int(p[1]), p[2], ".".join(p[3:])
        if i == 0:
            return "conv_in." + rest
        blk, lay = (i - 1) // per, (i - 1) % per
        if sub == "0" and rest.startswith("op."):
            return f"down_blocks.{blk}.downsamplers.0.conv.{rest[3:]}"
        if sub == "0":
            return f"down_blocks.{blk}.resnets.{lay}.{_resnet_tail(rest)}"
        if sub == "1":
            return f"down_blocks.{blk}.attentions.{lay}.{rest}"
        return None
    if head == "middle_block":
        sub, rest = p[1], ".".join(p[2:])
        if sub == "0":
            return "mid_block.resnets.0." + _resnet_tail(rest)
        if sub == "2":
            return "mid_block.resnets.1." + _resnet_tail(rest)
        if sub == "1":
            return "mid_block.attentions.0." + rest
        return None
    if head == "output_blocks":
        i, sub, rest = int(p[1]), p[2], ".".join(p[3:])
        blk, lay = i // per, i % per
        if sub == "0":
            return f"up_blocks.{blk}.resnets.{lay}.{_resnet_tail(rest)}"
        if rest.startswith("conv."):                             # the upsampler sits at .1 (no attention) or .2
            return f"up_blocks.{blk}.upsamplers.0.{rest}"
        if sub == "1":
            return f"up_blocks.{blk}.attentions.{lay}.{rest}"
        return None
    return None


def convert_ldm_unet_checkpoint(checkpoint: Dict[str, torch.Tensor], config, path=None, extract_ema: bool = False,
                                controlnet: bool = False, need_img_embed_concat: bool = False) -> Dict[str, torch.Tensor]:
    """LDM UNet -> diffusers UNet2DCondition/UNet3DCondition spatial keys.  `need_img_embed_concat=True` leaves conv_in out:
    the 9-channel conv_in of the mask/first-frame-concat model keeps its own weights (reference :382-384)."""
    if controlnet:
        raise NotImplementedError("ControlNet checkpoints are outside the FollowYourClick path")
    prefix = "model.diffusion_model."
    keys = list(checkpoint.keys())
    has_ema = sum(k.startswith("model_ema") for k in keys) > 100
    src: Dict[str, torch.Tensor] = {}
    if has_ema and extract_ema:
        print(f"Checkpoint {path} has both EMA and non-EMA weights.")
        for k in keys:
            if k.startswith("model.diffusion_model"):
                src[k[len(prefix):]] = checkpoint["model_ema." + "".join(k.split(".")[1:])]
    else:
        if has_ema:
            print("In this conversion only the non-EMA weights are extracted. If you want to instead extract the EMA"
                  " weights (usually better for inference), please make sure to add the `--extract_ema` flag.")
        for k in keys:
            if k.startswith(prefix):
                src[k[len(prefix):]] = checkpoint[k]
    cet = _cfg(config, "class_embed_type")
    if cet not in (None, "timestep", "projection"):
        raise NotImplementedError(f"Not implemented `class_embed_type`: {cet}")
    lpb = _cfg(config, "layers_per_block")
    out: Dict[str, torch.Tensor] = {}
    for k, v in src.items():
        nk = _unet_key(k, lpb)
        if nk is None or (cet is None and nk.startswith("class_embedding.")):
            continue
        if need_img_embed_concat and nk.startswith("conv_in."):
            continue
        out[nk] = v
    for must in ("time_embedding.linear_1.weight", "conv_norm_out.weight", "conv_out.weight"):
        if must not in out:
            raise KeyError(f"checkpoint has no UNet weights for {must} (prefix '{prefix}')")
    return out


_VAE_ATTN = (("norm.", "group_norm."), ("q.", "query."), ("k.", "key."), ("v.", "value."), ("proj_out.", "proj_attn."))


def _vae_key(key: str, num_up_blocks: int) -> Optional[str]:
    p = key.split(".")
    if p[0] in ("quant_conv", "post_quant_conv"):
        return key
    if p[0] not in ("encoder", "decoder"):
        return None
    side, rest = p[0], p[1:]
    if rest[0] in ("conv_in", "conv_out"):
        return key
    if rest[0] == "norm_out":
        return f"{side}.conv_norm_out.{rest[1]}"
    if rest[0] == "mid":
        if rest[1] in ("block_1", "block_2"):
            tail = ".".join(rest[2:]).replace("nin_shortcut", "conv_shortcut")
            return f"{side}.mid_block.resnets.{int(rest[1][-1]) - 1}.{tail}"
        if rest[1] == "attn_1":
            tail = ".".join(rest[2:])
            for old, new in _VAE_ATTN:
                if tail.startswith(old):
                    return f"{side}.mid_block.attentions.0.{new}{tail[len(old):]}"
        return None
    if rest[0] in ("down", "up"):
        i = int(rest[1])
        blk = i if rest[0] == "down" else num_up_blocks - 1 - i          # the LDM decoder counts levels from the output side
        name = "down_blocks" if rest[0] == "down" else "up_blocks"
        if rest[2] == "block":
            tail = ".".join(rest[4:]).replace("nin_shortcut", "conv_shortcut")
            return f"{side}.{name}.{blk}.resnets.{rest[3]}.{tail}"
        if rest[2] == "downsample":
            return f"{side}.{name}.{blk}.downsamplers.0.{'.'.join(rest[3:])}"
        if rest[2] == "upsample":
            return f"{side}.{name}.{blk}.upsamplers.0.{'.'.join(rest[3:])}"
    return None


def convert_ldm_vae_checkpoint(checkpoint: Dict[str, torch.Tensor], config) -> Dict[str, torch.Tensor]:
    prefix = "first_stage_model."
    src = {k[len(prefix):]: v for k, v in checkpoint.items() if k.startswith(prefix)}
    if "encoder.conv_in.weight" not in src or "decoder.conv_in.weight" not in src:
        raise KeyError(f"checkpoint has no VAE weights (prefix '{prefix}')")
    n_up = len({k.split(".")[2] for k in src if k.startswith("decoder.up.")})
    out: Dict[str, torch.Tensor] = {}
    for k, v in src.items():
        nk = _vae_key(k, n_up)
        if nk is None:
            continue
        if ".attentions." in nk and nk.endswith(".weight") and v.ndim > 2 and not nk.endswith("group_norm.weight"):
            v = v.reshape(v.shape[0], v.shape[1])                 # 1x1 conv -> linear
        out[nk] = v
    return out


def convert_ldm_clip_checkpoint(checkpoint: Dict[str, torch.Tensor], text_model=None):
    """`cond_stage_model.transformer.*` -> CLIPTextModel state dict.  With `text_model` given the weights are loaded into it
    (and it is returned, like the reference); otherwise the renamed state dict is returned."""
    prefix = "cond_stage_model.transformer."
    sd = {k[len(prefix):]: v for k, v in checkpoint.items() if k.startswith(prefix)}
    if text_model is None:
        return sd
    own = text_model.state_dict()
    text_model.load_state_dict({k: v for k, v in sd.items() if k in own or not k.endswith("position_ids")}, strict=False)
    return text_model
