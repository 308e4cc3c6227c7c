"""Generate tests/golden/convert_*.json|npz by running the REAL reference's checkpoint / LoRA converters (container only):
animatediff/utils/convert_from_ckpt.py::{convert_ldm_unet_checkpoint, convert_ldm_vae_checkpoint} and
animatediff/utils/convert_lora_safetensor_to_diffusers.py::{convert_lora, convert_motion_lora_ckpt_to_diffusers}.

Run:  python -m oracle.make_golden_convert

  convert_keymap.json   for a synthetic LDM-layout checkpoint (SD-1.5 topology, every tensor tagged with its index):
                        [ldm_key, diffusers_key, shape_after] triples produced by the reference, for UNet (plain and
                        need_img_embed_concat) and VAE
  convert_lora.npz      a kohya-style LoRA and a motion LoRA merged into the reference tiny UNet3D by the reference
                        functions: the LoRA tensors and the touched weights before/after
"""
import json
import os

import numpy as np
import torch

from . import functional as Fn
from . import refshim
from . import weights as W
from .make_golden import OUT, ref_unet

RES = ("in_layers.0", "in_layers.2", "emb_layers.1", "out_layers.0", "out_layers.3")
ATT = ["norm.weight", "norm.bias", "proj_in.weight", "proj_in.bias", "proj_out.weight", "proj_out.bias"] + \
      [f"transformer_blocks.0.{n}" for n in (
          "attn1.to_q.weight", "attn1.to_k.weight", "attn1.to_v.weight", "attn1.to_out.0.weight", "attn1.to_out.0.bias",
          "attn2.to_q.weight", "attn2.to_k.weight", "attn2.to_v.weight", "attn2.to_out.0.weight", "attn2.to_out.0.bias",
          "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias", "norm3.weight", "norm3.bias",
          "ff.net.0.proj.weight", "ff.net.0.proj.bias", "ff.net.2.weight", "ff.net.2.bias")]


def ldm_unet_keys():
    """SD-1.5 UNet in CompVis naming: 4 levels, 2 resnets per level, attention on levels 0-2"""
    keys = [f"time_embed.{i}.{p}" for i in (0, 2) for p in ("weight", "bias")]
    keys += ["input_blocks.0.0.weight", "input_blocks.0.0.bias", "out.0.weight", "out.0.bias", "out.2.weight", "out.2.bias"]

    def res(prefix, skip):
        return [f"{prefix}.{r}.{p}" for r in RES + (("skip_connection",) if skip else ()) for p in ("weight", "bias")]

    i = 1
    for level in range(4):
        for j in range(2):
            keys += res(f"input_blocks.{i}.0", skip=(j == 0 and level in (1, 2)))
            if level < 3:
                keys += [f"input_blocks.{i}.1.{a}" for a in ATT]
            i += 1
        if level < 3:
            keys += [f"input_blocks.{i}.0.op.weight", f"input_blocks.{i}.0.op.bias"]
            i += 1
    keys += res("middle_block.0", False) + [f"middle_block.1.{a}" for a in ATT] + res("middle_block.2", False)
    for i in range(12):
        blk, j = i // 3, i % 3
        keys += res(f"output_blocks.{i}.0", skip=True)
        if blk >= 1:
            keys += [f"output_blocks.{i}.1.{a}" for a in ATT]
        if j == 2 and blk < 3:
            sub = 2 if blk >= 1 else 1
            keys += [f"output_blocks.{i}.{sub}.conv.weight", f"output_blocks.{i}.{sub}.conv.bias"]
    return ["model.diffusion_model." + k for k in keys]


def ldm_vae_tensors():
    """AutoencoderKL in CompVis naming; attention q/k/v/proj_out are 1x1 convs there"""
    t = {}
    n = [0]

    def add(k, shape=(2,)):
        t["first_stage_model." + k] = torch.full(shape, float(n[0]))
        n[0] += 1

    def res(prefix, skip):
        for r in ("norm1", "conv1", "norm2", "conv2") + (("nin_shortcut",) if skip else ()):
            add(f"{prefix}.{r}.weight"), add(f"{prefix}.{r}.bias")

    for side, per in (("encoder", 2), ("decoder", 3)):
        for name in ("conv_in", "norm_out", "conv_out"):
            add(f"{side}.{name}.weight"), add(f"{side}.{name}.bias")
        for lvl in range(4):
            grp = "down" if side == "encoder" else "up"
            for j in range(per):
                # channel changes: encoder 128->256->512 entering levels 1, 2; decoder (levels counted from the output) 512->256->128 entering 1, 0
                res(f"{side}.{grp}.{lvl}.block.{j}", skip=(j == 0 and lvl in ((1, 2) if side == "encoder" else (0, 1))))
            if (side == "encoder" and lvl < 3) or (side == "decoder" and lvl > 0):
                s = "downsample" if side == "encoder" else "upsample"
                add(f"{side}.{grp}.{lvl}.{s}.conv.weight"), add(f"{side}.{grp}.{lvl}.{s}.conv.bias")
        res(f"{side}.mid.block_1", False), res(f"{side}.mid.block_2", False)
        add(f"{side}.mid.attn_1.norm.weight", (6,)), add(f"{side}.mid.attn_1.norm.bias", (6,))
        for a in ("q", "k", "v", "proj_out"):
            add(f"{side}.mid.attn_1.{a}.weight", (6, 6, 1, 1)), add(f"{side}.mid.attn_1.{a}.bias", (6,))
    for name in ("quant_conv", "post_quant_conv"):
        add(f"{name}.weight"), add(f"{name}.bias")
    return t


def keymap(src, out):
    by_tag = {float(v.flatten()[0]): k for k, v in src.items()}
    return sorted([by_tag[float(v.flatten()[0])], k, list(v.shape)] for k, v in out.items())


def main():
    refshim.install()
    from animatediff.utils import convert_from_ckpt as C
    from animatediff.utils import convert_lora_safetensor_to_diffusers as L
    cfg = {"layers_per_block": 2, "class_embed_type": None}
    unet_src = {k: torch.full((2,), float(i)) for i, k in enumerate(ldm_unet_keys())}
    unet_src["cond_stage_model.transformer.text_model.final_layer_norm.weight"] = torch.full((2,), -1.0)   # foreign keys are ignored
    vae_src = ldm_vae_tensors()
    golden = {
        "unet": keymap(unet_src, C.convert_ldm_unet_checkpoint(dict(unet_src), cfg)),
        "unet_img_embed_concat": keymap(unet_src, C.convert_ldm_unet_checkpoint(dict(unet_src), cfg, need_img_embed_concat=True)),
        "vae": keymap(vae_src, C.convert_ldm_vae_checkpoint(dict(vae_src), {})),
        "vae_src_shapes": {k: list(v.shape) for k, v in vae_src.items()},
    }
    with open(os.path.join(OUT, "convert_keymap.json"), "w") as f:
        json.dump(golden, f, indent=0)

    # ---- LoRA merges on the reference tiny UNet3D ----------------------------------------------------------------
    ucfg = Fn.tiny_unet_config()
    unet = ref_unet(ucfg).eval()
    unet.load_state_dict(W.make_weights(W.unet_state_shapes(ucfg), seed=0))
    g = torch.Generator().manual_seed(77)
    r = 4
    targets = {  # kohya name -> state-dict key (small layers: the fixture stores the merged weights)
        "lora_unet_down_blocks_0_attentions_0_transformer_blocks_0_attn1_to_q": "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight",
        "lora_unet_mid_block_attentions_0_transformer_blocks_0_attn2_to_k": "mid_block.attentions.0.transformer_blocks.0.attn2.to_k.weight",
        "lora_unet_up_blocks_3_attentions_2_transformer_blocks_0_attn2_to_out_0": "up_blocks.3.attentions.2.transformer_blocks.0.attn2.to_out.0.weight",
        "lora_unet_down_blocks_0_attentions_1_transformer_blocks_0_ff_net_0_proj": "down_blocks.0.attentions.1.transformer_blocks.0.ff.net.0.proj.weight",
        "lora_unet_down_blocks_0_attentions_1_proj_in": "down_blocks.0.attentions.1.proj_in.weight",          # 1x1 conv: 4-D LoRA
    }
    sd0 = {k: v.clone() for k, v in unet.state_dict().items()}
    lora = {}
    for name, key in targets.items():
        w = sd0[key]
        if w.dim() == 4:
            lora[name + ".lora_down.weight"] = torch.randn(r, w.shape[1], 1, 1, generator=g) * 0.1
            lora[name + ".lora_up.weight"] = torch.randn(w.shape[0], r, 1, 1, generator=g) * 0.1
        else:
            lora[name + ".lora_down.weight"] = torch.randn(r, w.shape[1], generator=g) * 0.1
            lora[name + ".lora_up.weight"] = torch.randn(w.shape[0], r, generator=g) * 0.1
        lora[name + ".alpha"] = torch.tensor(float(r))

    class Pipe:
        pass

    pipe = Pipe()
    pipe.unet, pipe.text_encoder = unet, None
    L.convert_lora(pipe, lora, alpha=0.8)
    mm = "down_blocks.0.motion_modules.0.temporal_transformer.transformer_blocks.0.attention_blocks.0"
    motion = {}
    for proj in ("to_q", "to_out"):
        key = f"{mm}.{proj}.weight" if proj != "to_out" else f"{mm}.to_out.0.weight"
        w = sd0[key]
        motion[f"module.{mm}.processor.{proj}_lora.down.weight"] = torch.randn(r, w.shape[1], generator=g) * 0.1
        motion[f"module.{mm}.processor.{proj}_lora.up.weight"] = torch.randn(w.shape[0], r, generator=g) * 0.1
    L.convert_motion_lora_ckpt_to_diffusers(pipe, motion, alpha=0.5)
    sd1 = unet.state_dict()
    changed = [k for k in sd0 if not torch.equal(sd0[k], sd1[k])]
    out = {"changed": np.array(changed)}
    for k, v in lora.items():
        out["lora/" + k] = v.numpy()
    for k, v in motion.items():
        out["motion/" + k] = v.numpy()
    for k in changed:
        out["after/" + k] = sd1[k].numpy()
    np.savez_compressed(os.path.join(OUT, "convert_lora.npz"), **out)
    print("changed:", changed)
    for f in ("convert_keymap.json", "convert_lora.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
