"""nn.Module-shaped fronts for the conditioning-encoder engines (followyourclick_amd.engine.encoders).

`ClipTextHip` stands in for the `transformers.CLIPTextModel` the reference pipelines hold as `text_encoder`
(pipeline_animation.py:183-186: `self.text_encoder(ids, attention_mask=None)[0]`), `ClipVisionHip` for the
`CLIPVisionModelWithProjection` of ip_adapter/my_ip_adapter.py:58 (`.image_embeds`, `.hidden_states[-2]`).  Both keep
the original parameters (same state-dict names) and pack them for the HIP kernels on first use; there is no CPU fallback.

    pipeline.text_encoder = ClipTextHip.from_transformers(pipeline.text_encoder)
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace
from typing import List, Optional

import torch
import torch.nn as nn

from .engine import encoders as EN


class EngineBacked(nn.Module):
    """parameters live on the module under their checkpoint names; the packed device copy is rebuilt when they change"""
    compute_dtype: torch.dtype = None

    def _init_engine_state(self, compute_dtype: torch.dtype = None) -> None:
        from . import default_compute_dtype
        self.compute_dtype = compute_dtype if compute_dtype is not None else default_compute_dtype()
        self._engine = None
        self._engine_key = None

    def invalidate_engine(self) -> None:
        self._engine_key = None

    def _device(self) -> torch.device:
        return next(self.parameters()).device

    def _weights_key(self):
        return (self._device(), self.compute_dtype) + tuple(p._version for p in self.parameters())

    def _build_engine(self, sd, device):  # pragma: no cover - abstract
        raise NotImplementedError

    def _get_engine(self):
        key = self._weights_key()
        if self._engine is None or key != self._engine_key:
            dev = self._device()
            if dev.type != "cuda":
                raise RuntimeError(f"{type(self).__name__} runs on an MI355X HIP device only (call .to('cuda')); no CPU fallback")
            self._engine = self._build_engine({k: v.detach() for k, v in self.state_dict().items()}, dev)
            self._engine_key = key
        return self._engine

    @property
    def device(self) -> torch.device:
        return self._device()

    @property
    def dtype(self) -> torch.dtype:
        return next(self.parameters()).dtype


def _register(module: nn.Module, sd) -> None:
    """attach tensors under dotted names (children are plain containers)"""
    for name, t in sd.items():
        *path, leaf = name.split(".")
        node = module
        for part in path:
            if part not in node._modules:
                node.add_module(part, nn.Module())
            node = node._modules[part]
        node.register_parameter(leaf, nn.Parameter(t.detach().clone().float(), requires_grad=False))


@dataclass
class TextEncoderOutput:
    last_hidden_state: torch.Tensor
    pooler_output: Optional[torch.Tensor] = None

    def __getitem__(self, i):
        return (self.last_hidden_state, self.pooler_output)[i]


class ClipTextHip(EngineBacked):
    def __init__(self, state_dict, config, compute_dtype: torch.dtype = None):
        super().__init__()
        get = (lambda k, d=None: getattr(config, k, d)) if not isinstance(config, dict) else (lambda k, d=None: config.get(k, d))
        self.engine_config = EN.ClipTextConfig(
            vocab_size=get("vocab_size"), hidden_size=get("hidden_size"), intermediate_size=get("intermediate_size"),
            num_hidden_layers=get("num_hidden_layers"), num_attention_heads=get("num_attention_heads"),
            max_position_embeddings=get("max_position_embeddings", 77), hidden_act=get("hidden_act", "quick_gelu"),
            layer_norm_eps=get("layer_norm_eps", 1e-5))
        if self.engine_config.hidden_act not in EN.ACTS:
            raise NotImplementedError(f"CLIP text encoder activation '{self.engine_config.hidden_act}'")
        self.config = SimpleNamespace(**config) if isinstance(config, dict) else config
        self.eos_token_id = get("eos_token_id", 2)
        _register(self, {k: v for k, v in state_dict.items() if not k.endswith("position_ids")})
        self._init_engine_state(compute_dtype)

    @classmethod
    def from_transformers(cls, model, compute_dtype: torch.dtype = None) -> "ClipTextHip":
        m = cls(model.state_dict(), model.config, compute_dtype)
        return m.to(next(model.parameters()).device)

    def _build_engine(self, sd, device):
        return EN.ClipTextEngine(EN.pack_clip_text(sd, self.engine_config, self.compute_dtype, device))

    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor, attention_mask=None, position_ids=None, **unused):
        if attention_mask is not None:
            raise NotImplementedError("CLIP text encoder on the MI355X engine: padding attention_mask is not implemented "
                                      "(the reference passes attention_mask=None, pipeline_animation.py:178-181)")
        ids = input_ids.reshape(-1, input_ids.shape[-1])
        h = self._get_engine().encode(ids)
        e = self.eos_token_id
        idx = ids.to(h.device).argmax(-1) if e == 2 else (ids.to(h.device) == e).int().argmax(-1)   # transformers' pooled-token rule
        return TextEncoderOutput(last_hidden_state=h, pooler_output=h[torch.arange(h.shape[0], device=h.device), idx])


@dataclass
class VisionEncoderOutput:
    image_embeds: Optional[torch.Tensor]
    last_hidden_state: Optional[torch.Tensor]
    hidden_states: Optional[List[Optional[torch.Tensor]]]


class ClipVisionHip(EngineBacked):
    def __init__(self, state_dict, config, compute_dtype: torch.dtype = None):
        super().__init__()
        get = (lambda k, d=None: getattr(config, k, d)) if not isinstance(config, dict) else (lambda k, d=None: config.get(k, d))
        self.engine_config = EN.ClipVisionConfig(
            hidden_size=get("hidden_size"), intermediate_size=get("intermediate_size"), num_hidden_layers=get("num_hidden_layers"),
            num_attention_heads=get("num_attention_heads"), image_size=get("image_size", 224), patch_size=get("patch_size", 14),
            projection_dim=get("projection_dim", 1024), hidden_act=get("hidden_act", "gelu"), layer_norm_eps=get("layer_norm_eps", 1e-5))
        if self.engine_config.hidden_act not in EN.ACTS:
            raise NotImplementedError(f"CLIP vision tower activation '{self.engine_config.hidden_act}'")
        self.config = SimpleNamespace(**config) if isinstance(config, dict) else config
        _register(self, {k: v for k, v in state_dict.items() if not k.endswith("position_ids")})
        self._init_engine_state(compute_dtype)

    @classmethod
    def from_transformers(cls, model, compute_dtype: torch.dtype = None) -> "ClipVisionHip":
        m = cls(model.state_dict(), model.config, compute_dtype)
        return m.to(next(model.parameters()).device)

    def requires_grad_(self, flag: bool = True):
        if flag:
            raise NotImplementedError("inference-only module")
        return self

    def _build_engine(self, sd, device):
        return EN.ClipVisionEngine(EN.pack_clip_vision(sd, self.engine_config, self.compute_dtype, device))

    @torch.no_grad()
    def forward(self, pixel_values: torch.Tensor, output_hidden_states: bool = False, **unused) -> VisionEncoderOutput:
        """`hidden_states` holds only its last two entries (what ip_adapter reads: `[-2]`); earlier ones are None"""
        eng = self._get_engine()
        if output_hidden_states:
            out = eng.encode(pixel_values, want=("image_embeds", "penultimate", "last") if eng.P.proj_w is not None else ("penultimate", "last"))
            hs = [None] * (self.engine_config.num_hidden_layers - 1) + [out["penultimate"], out["last"]]
            return VisionEncoderOutput(image_embeds=out.get("image_embeds"), last_hidden_state=out["last"], hidden_states=hs)
        out = eng.encode(pixel_values, want=("image_embeds",))
        return VisionEncoderOutput(image_embeds=out["image_embeds"], last_hidden_state=None, hidden_states=None)


def config_namespace(**kw) -> SimpleNamespace:
    return SimpleNamespace(**kw)
