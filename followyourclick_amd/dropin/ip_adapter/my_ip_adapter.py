"""`ip_adapter.my_ip_adapter` on the MI355X engine: ImageProjModel, MyIPAdapter, MyIPAdapterPlus
(reference ip_adapter/my_ip_adapter.py:28-45, 48-147, 237-330) - same constructors, attributes and methods; the CLIP
vision tower and the projection / resampler run as HIP op schedules (followyourclick_amd.engine.encoders).

Still host-side, as in the reference: `CLIPImageProcessor` (PIL resize / crop / normalise) and reading the checkpoint files.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from followyourclick_amd.encoders import ClipVisionHip, EngineBacked
from followyourclick_amd.engine import encoders as EN

from .resampler import Resampler


class ImageProjModel(EngineBacked):
    """Projection Model: Linear(clip_dim -> tokens*D) -> (B, tokens, D) -> LayerNorm(D)"""

    def __init__(self, cross_attention_dim=1024, clip_embeddings_dim=1024, clip_extra_context_tokens=4,
                 compute_dtype: torch.dtype = None):
        super().__init__()
        self.cross_attention_dim = cross_attention_dim
        self.clip_extra_context_tokens = clip_extra_context_tokens
        self.proj = nn.Linear(clip_embeddings_dim, clip_extra_context_tokens * cross_attention_dim)
        self.norm = nn.LayerNorm(cross_attention_dim)
        for p in self.parameters():
            p.requires_grad_(False)
        self._init_engine_state(compute_dtype)

    def _build_engine(self, sd, device):
        return EN.ImageProjEngine(EN.pack_image_proj(sd, self.compute_dtype, device))

    @torch.no_grad()
    def forward(self, image_embeds: torch.Tensor) -> torch.Tensor:
        return self._get_engine().project(image_embeds).to(image_embeds.dtype)


def _is_pil(x) -> bool:
    try:
        from PIL import Image
        return isinstance(x, Image.Image)
    except ImportError:  # pragma: no cover
        return False


class MyIPAdapter:
    def __init__(self, unet, image_encoder_path, ip_ckpt, device, num_tokens=4):
        self.device = device
        self.image_encoder_path = image_encoder_path
        self.ip_ckpt = ip_ckpt
        self.num_tokens = num_tokens
        self.unet = unet
        if isinstance(image_encoder_path, (str, os.PathLike)):
            from transformers import CLIPVisionModelWithProjection
            torch_model = CLIPVisionModelWithProjection.from_pretrained(image_encoder_path)
        else:                                   # an already constructed vision model (tests, custom loaders)
            torch_model = image_encoder_path
        enc = torch_model if isinstance(torch_model, ClipVisionHip) else ClipVisionHip.from_transformers(torch_model)
        self.image_encoder = enc.to(self.device)
        self.image_encoder.requires_grad_(False)
        try:
            from transformers import CLIPImageProcessor
            self.clip_image_processor = CLIPImageProcessor()
        except Exception:  # pragma: no cover - transformers without an image backend
            self.clip_image_processor = None
        self.image_proj_model = self.init_proj()

    def init_proj(self):
        return ImageProjModel(cross_attention_dim=self.unet.config.cross_attention_dim,
                              clip_embeddings_dim=self.image_encoder.config.projection_dim,
                              clip_extra_context_tokens=self.num_tokens).to(self.device)

    def get_ip_adapter_state_dict(self):
        if os.path.splitext(self.ip_ckpt)[-1] == ".safetensors":
            from safetensors import safe_open
            state_dict = {"image_proj": {}, "ip_adapter": {}}
            with safe_open(self.ip_ckpt, framework="pt", device="cpu") as f:
                for key in f.keys():
                    for part in ("image_proj", "ip_adapter"):
                        if key.startswith(part + "."):
                            state_dict[part][key[len(part) + 1:]] = f.get_tensor(key)
            return state_dict
        return torch.load(self.ip_ckpt, map_location="cpu")

    def _load_image_proj(self, state_dict, unet, use_unet_image_proj_model, unet_state_dict):
        if use_unet_image_proj_model:
            for k in ("proj.weight", "proj.bias", "norm.weight", "norm.bias"):
                unet_state_dict["image_proj_model." + k] = state_dict["image_proj"][k]
        else:
            m, u = self.image_proj_model.load_state_dict(state_dict["image_proj"])
            print("load image_proj_model: missing keys: {}, unexpected keys: {}".format(len(m), len(u)))

    def _load_ip_kv(self, state_dict, unet_state_dict):
        """the checkpoint's `ip_adapter` entries are matched IN ORDER with the UNet's `*_ip*` parameters (reference :104-119)"""
        ip_keys = list(state_dict["ip_adapter"].keys())
        model_keys = [k for k in unet_state_dict if "_ip" in k]
        for k1, k2 in zip(model_keys, ip_keys):
            assert unet_state_dict[k1].shape == state_dict["ip_adapter"][k2].shape, (k1, k2)
            unet_state_dict[k1] = state_dict["ip_adapter"][k2]

    def load_ip_adapter(self, unet=None, use_unet_image_proj_model=False):
        state_dict = self.get_ip_adapter_state_dict()
        target = unet if unet is not None else self.unet
        unet_state_dict = target.state_dict()
        self._load_image_proj(state_dict, target, use_unet_image_proj_model, unet_state_dict)
        self._load_ip_kv(state_dict, unet_state_dict)
        missing, unexpected = target.load_state_dict(unet_state_dict, strict=False)
        print("load ip_adapter to unet: missing keys: {}, unexpected keys: {}".format(len(missing), len(unexpected)))
        return missing, unexpected

    def _pixels(self, input_image):
        if _is_pil(input_image):
            input_image = [input_image]
        if isinstance(input_image, (list, tuple)):
            input_image = self.clip_image_processor(images=list(input_image), return_tensors="pt").pixel_values
        return input_image.to(self.device)

    @torch.no_grad()
    def get_image_clip_feat(self, input_image=None):
        clip_image_embeds = self.image_encoder(self._pixels(input_image)).image_embeds
        return clip_image_embeds, torch.zeros_like(clip_image_embeds)

    @torch.inference_mode()
    def get_image_embeds(self, input_image=None, clip_image_embeds=None, image_proj_model=None):
        if input_image is not None:
            clip_image_embeds = self.image_encoder(self._pixels(input_image)).image_embeds
        else:
            clip_image_embeds = clip_image_embeds.to(self.device)
        proj = self.image_proj_model if image_proj_model is None else image_proj_model
        return proj(clip_image_embeds), proj(torch.zeros_like(clip_image_embeds))


class MyIPAdapterPlus(MyIPAdapter):
    """IP-Adapter with fine-grained features: penultimate CLIP hidden states -> Resampler"""

    def init_proj(self):
        return Resampler(dim=self.unet.config.cross_attention_dim, depth=4, dim_head=64, heads=12, num_queries=self.num_tokens,
                         embedding_dim=self.image_encoder.config.hidden_size, output_dim=self.unet.config.cross_attention_dim,
                         ff_mult=4).to(self.device)

    def _load_image_proj(self, state_dict, unet, use_unet_image_proj_model, unet_state_dict):
        if use_unet_image_proj_model:
            unet.image_proj_model = self.init_proj()
            m, u = unet.image_proj_model.load_state_dict(state_dict["image_proj"])
        else:
            m, u = self.image_proj_model.load_state_dict(state_dict["image_proj"])
        print("load image_proj_model: missing keys: {}, unexpected keys: {}".format(len(m), len(u)))

    @torch.no_grad()
    def get_image_clip_feat(self, input_image=None):
        px = self._pixels(input_image)
        both = self.image_encoder(torch.cat([px, torch.zeros_like(px)]), output_hidden_states=True).hidden_states[-2]
        return both[:px.shape[0]], both[px.shape[0]:]

    @torch.inference_mode()
    def get_image_embeds(self, input_image=None, clip_image_embeds=None, image_proj_model=None):
        cond, uncond = self.get_image_clip_feat(input_image)
        proj = self.image_proj_model if image_proj_model is None else image_proj_model
        return proj(cond), proj(uncond)
