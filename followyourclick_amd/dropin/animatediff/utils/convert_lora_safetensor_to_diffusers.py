"""`animatediff.utils.convert_lora_safetensor_to_diffusers`: merge LoRA deltas into the model weights
(scripts/inference.py:245-246, scripts/inference_w_camera_lora.py).

Same functions, arguments and arithmetic as reference animatediff/utils/convert_lora_safetensor_to_diffusers.py:26-51
(motion LoRA) and :94-154 (kohya-style `lora_unet_* / lora_te_*`): W += alpha * up @ down, in f32, on the module found by
walking the `_`-joined layer name through the module tree.  Works on any nn.Module tree with the reference's names - the
drop-in UNet registers its parameters under exactly those.
"""
from __future__ import annotations

import torch


def _child(module, name: str):
    mods = getattr(module, "_modules", {})
    if name in mods and mods[name] is not None:
        return mods[name]
    return None


def _walk_underscored(root, tokens):
    """`down_blocks_0_attentions_0_..._to_q` -> module: extend the candidate name token by token until a child matches"""
    cur, i = root, 0
    while i < len(tokens):
        for j in range(i + 1, len(tokens) + 1):
            nxt = _child(cur, "_".join(tokens[i:j]))
            if nxt is not None:
                cur, i = nxt, j
                break
        else:
            raise KeyError(f"no module for '{'_'.join(tokens)}' (stuck at '{'_'.join(tokens[i:])}')")
    return cur


def _touch(pipeline) -> None:
    """`.data` updates do not bump tensor versions: tell an engine-backed UNet that its packed weights are stale"""
    inval = getattr(getattr(pipeline, "unet", None), "invalidate_engine", None)
    if inval is not None:
        inval()


def convert_lora(pipeline, state_dict, LORA_PREFIX_UNET="lora_unet", LORA_PREFIX_TEXT_ENCODER="lora_te", alpha=0.6):
    visited = set()
    for key in state_dict:
        if ".alpha" in key or key in visited:
            continue
        if "text" in key:
            tokens = key.split(".")[0].split(LORA_PREFIX_TEXT_ENCODER + "_")[-1].split("_")
            layer = _walk_underscored(pipeline.text_encoder, tokens)
        else:
            tokens = key.split(".")[0].split(LORA_PREFIX_UNET + "_")[-1].split("_")
            layer = _walk_underscored(pipeline.unet, tokens)
        if "lora_down" in key:
            up_key, down_key = key.replace("lora_down", "lora_up"), key
        else:
            up_key, down_key = key, key.replace("lora_up", "lora_down")
        up, down = state_dict[up_key].to(torch.float32), state_dict[down_key].to(torch.float32)
        w = layer.weight.data
        if up.dim() == 4:
            delta = torch.mm(up.squeeze(3).squeeze(2), down.squeeze(3).squeeze(2)).unsqueeze(2).unsqueeze(3)
        else:
            delta = torch.mm(up, down)
        w += alpha * delta.to(w.device)
        visited.update((up_key, down_key))
    _touch(pipeline)
    return pipeline


def convert_motion_lora_ckpt_to_diffusers(pipeline, state_dict, alpha=1.0):
    for key in state_dict:
        if "lora" not in key or "up." in key:
            continue
        up_key = key.replace(".down.", ".up.")
        model_key = key.replace("processor.", "").replace("_lora", "").replace("down.", "").replace("up.", "").replace("module.", "")
        model_key = model_key.replace("to_out.", "to_out.0.")
        layer = pipeline.unet
        for name in model_key.split(".")[:-1]:
            nxt = _child(layer, name)
            if nxt is None:
                raise KeyError(f"no module '{name}' on the way to '{model_key}'")
            layer = nxt
        w = layer.weight.data
        w += alpha * torch.mm(state_dict[up_key], state_dict[key]).to(w.device)
    _touch(pipeline)
    return pipeline
