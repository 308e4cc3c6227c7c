"""`animatediff.models.unet.UNet3DConditionModel` on the MI355X engine.

Keeps the reference call surface (reference animatediff/models/unet.py: constructor kwargs :43-104,
`forward` :422-444 / :669-672, `from_pretrained_2d` :674-726) and - because the checkpoint format of the
reference IS its state-dict key set - registers every parameter / buffer under the reference's exact
name and shape (followyourclick_amd.engine.schema), so `state_dict()`, `load_state_dict(strict=False)`
and the motion-module / DreamBooth / LoRA loading code of scripts/inference.py work unchanged.

The math runs in libfyc_hip.so: on the first forward (and again whenever a parameter changed) the
reference-layout weights are packed into the engine's kernel-friendly device buffers.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from followyourclick_amd import default_compute_dtype, ops as ops_mod
from followyourclick_amd.engine import UNet3DConfig
from followyourclick_amd.engine.schema import unet_schema
from followyourclick_amd.engine.unet3d import UNet3DEngine
from followyourclick_amd.engine.weights import pack_unet, pad_channels, sinusoidal_pe


@dataclass
class UNet3DConditionOutput:
    sample: torch.Tensor

    def __getitem__(self, i):  # tuple-style access like diffusers' BaseOutput
        return (self.sample,)[i]


class _Node(nn.Module):
    """container whose children may have numeric names (ModuleList-style keys such as 'resnets.0')"""


def _attach(root: nn.Module, name: str, shape, buffer: bool) -> None:
    *path, leaf = name.split(".")
    node = root
    for part in path:
        if part not in node._modules:
            node.add_module(part, _Node())
        node = node._modules[part]
    if buffer:
        node.register_buffer(leaf, sinusoidal_pe(shape[2], shape[1])[None].clone(), persistent=True)
    else:
        node.register_parameter(leaf, nn.Parameter(torch.empty(shape), requires_grad=False))


class UNet3DConditionModel(nn.Module):
    config_name = "config.json"

    def __init__(self, sample_size: Optional[int] = None, in_channels: int = 4, out_channels: int = 4,
                 center_input_sample: bool = False, flip_sin_to_cos: bool = True, freq_shift: int = 0,
                 down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"),
                 mid_block_type: str = "UNetMidBlock3DCrossAttn",
                 up_block_types: Tuple[str, ...] = ("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"),
                 only_cross_attention=False, block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280), layers_per_block: int = 2,
                 downsample_padding: int = 1, mid_block_scale_factor: float = 1, act_fn: str = "silu", norm_num_groups: int = 32,
                 norm_eps: float = 1e-5, cross_attention_dim: int = 1280, attention_head_dim=8, dual_cross_attention: bool = False,
                 use_linear_projection: bool = False, class_embed_type=None, num_class_embeds=None, upcast_attention: bool = False,
                 resnet_time_scale_shift: str = "default",
                 use_motion_module=False, motion_module_resolutions=(1, 2, 4, 8), motion_module_mid_block=False,
                 motion_module_decoder_only=False, motion_module_type=None, motion_module_kwargs=None,
                 unet_use_cross_frame_attention=None, unet_use_temporal_attention=None, use_pseudo_conv3d=False,
                 use_first_frame_condition_concat=False, image_condition_dim=1024, use_ip_cross_attention=False, scale=1.0,
                 num_tokens=4, use_camera_motion_condition=False, use_text_encoder_2=False, text_encoder_2_dim=4096,
                 use_inflated_groupnorm=False, use_fps_condition=False, use_temporal_conv=False,
                 use_first_frame_mask_condition_concat=False, compute_dtype: torch.dtype = None, **unused):
        super().__init__()
        kwargs = {k: v for k, v in locals().items() if k not in ("self", "unused", "__class__", "compute_dtype")}
        mm = dict(motion_module_kwargs or {})
        unsupported = {
            "center_input_sample": center_input_sample, "dual_cross_attention": dual_cross_attention,
            "use_linear_projection": use_linear_projection, "class_embed_type": class_embed_type, "num_class_embeds": num_class_embeds,
            "unet_use_cross_frame_attention": unet_use_cross_frame_attention, "unet_use_temporal_attention": unet_use_temporal_attention,
            "use_pseudo_conv3d": use_pseudo_conv3d,
            "use_text_encoder_2": use_text_encoder_2, "use_inflated_groupnorm": use_inflated_groupnorm,
            "use_temporal_conv": use_temporal_conv, "motion_module_decoder_only": motion_module_decoder_only,
            "use_rope_postion_encoding": mm.get("use_rope_postion_encoding", False), "add_temporal_lora": mm.get("add_temporal_lora", False),
        }
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError(f"UNet3DConditionModel on the MI355X engine: unsupported options {bad} "
                                      "(outside the shipped inference configs, SURVEY.md 8)")
        if not flip_sin_to_cos or freq_shift != 0 or act_fn != "silu" or resnet_time_scale_shift != "default":
            raise NotImplementedError("only flip_sin_to_cos=True, freq_shift=0, act_fn='silu', resnet_time_scale_shift='default'")
        if use_motion_module and motion_module_type != "Vanilla":
            raise ValueError("motion_module_type must be 'Vanilla'")
        blocks = mm.get("attention_block_types", ("Temporal_Self", "Temporal_Self"))
        if any(b != "Temporal_Self" for b in blocks) or mm.get("temporal_attention_dim_div", 1) != 1:
            raise NotImplementedError("motion module: only Temporal_Self attention blocks with dim_div 1")
        head = attention_head_dim if isinstance(attention_head_dim, int) else attention_head_dim[0]
        self.engine_config = UNet3DConfig(
            sample_size=sample_size or 64, in_channels=in_channels, out_channels=out_channels,
            block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block, cross_attention_dim=cross_attention_dim,
            attention_head_dim=head, norm_num_groups=norm_num_groups, norm_eps=norm_eps, down_block_types=tuple(down_block_types),
            up_block_types=tuple(up_block_types), use_motion_module=bool(use_motion_module),
            motion_module_resolutions=tuple(motion_module_resolutions), motion_module_mid_block=bool(motion_module_mid_block),
            motion_num_attention_heads=mm.get("num_attention_heads", 8), motion_num_transformer_block=mm.get("num_transformer_block", 2),
            motion_attention_blocks=len(blocks), temporal_position_encoding=bool(mm.get("temporal_position_encoding", False)),
            temporal_position_encoding_max_len=mm.get("temporal_position_encoding_max_len", 24), use_fps_condition=bool(use_fps_condition),
            use_first_frame_mask_condition_concat=bool(use_first_frame_mask_condition_concat),
            use_first_frame_condition_concat=bool(use_first_frame_condition_concat), use_camera_motion_condition=bool(use_camera_motion_condition),
            use_ip_cross_attention=bool(use_ip_cross_attention),
            ip_scale=float(scale), ip_num_tokens=int(num_tokens))
        # diffusers' @register_to_config contract: every ctor kwarg is an attribute and a `.config` entry
        # (the pipeline reads unet.in_channels and unet.config.sample_size, reference pipeline_animation.py:586-587,639)
        for k, v in kwargs.items():
            if k not in ("scale",):
                setattr(self, k, v)
        self.ip_scale = scale
        self.config = SimpleNamespace(**kwargs)
        self.compute_dtype = compute_dtype if compute_dtype is not None else default_compute_dtype()
        for name, shape in unet_schema(self.engine_config).items():
            _attach(self, name, shape, buffer=name.endswith("pos_encoder.pe"))
        self.image_proj_model = None   # set by scripts/inference.py:167 (`unet.image_proj_model = ip_adapter.init_proj()`)
        self._engine: Optional[UNet3DEngine] = None
        self._engine_key = None
        self._ctx_key = None
        self._ctx_cache_obj = None
        self._reset_parameters()

    # ---- nn.Module plumbing -------------------------------------------------------------------
    def _reset_parameters(self) -> None:
        g = torch.Generator().manual_seed(0)
        for name, p in self.named_parameters():
            if name.startswith("image_proj_model"):
                continue
            if p.dim() == 1:
                p.data.fill_(1.0 if name.endswith("weight") else 0.0)
            else:
                fan_in = p[0].numel()
                p.data.copy_(torch.randn(p.shape, generator=g) / fan_in ** 0.5)
        if self.engine_config.use_motion_module:  # zero_initialize=True (reference motion_module.py:87-88)
            for name, p in self.named_parameters():
                if ".temporal_transformer.proj_out." in name:
                    p.data.zero_()
        for emb in ("fps_embedding", "motion_embedding"):  # reference unet.py:141-146
            if hasattr(self, emb):
                getattr(self, emb).linear_2.weight.data.zero_()
                getattr(self, emb).linear_2.bias.data.zero_()

    @property
    def dtype(self) -> torch.dtype:
        return self.conv_in.weight.dtype

    @property
    def device(self) -> torch.device:
        return self.conv_in.weight.device

    def enable_xformers_memory_efficient_attention(self, *a, **k) -> None:
        """no-op: the fused MFMA attention kernel is always used (scripts/inference.py:157-158 calls this)"""

    def disable_xformers_memory_efficient_attention(self) -> None:
        pass

    def set_attention_slice(self, slice_size) -> None:
        """no-op: attention never materialises the score matrix, there is nothing to slice"""

    def invalidate_engine(self) -> None:
        """force re-packing on the next forward (for in-place `.data` edits, which do not bump tensor versions)"""
        self._engine_key = None

    def _weights_key(self):
        return (self.device, self.compute_dtype) + tuple(p._version for p in self.parameters(recurse=True) if p.dim() > 0)

    def _get_engine(self) -> UNet3DEngine:
        key = self._weights_key()
        if self._engine is None or key != self._engine_key:
            if self.device.type != "cuda" and ops_mod.get().name == "hip":
                raise RuntimeError("UNet3DConditionModel runs on an MI355X HIP device only (call .to('cuda')); no CPU fallback")
            sd = {k: v for k, v in self.state_dict().items() if not k.startswith("image_proj_model")}
            packed = pack_unet(sd, self.engine_config, self.compute_dtype, self.device)
            self._engine = UNet3DEngine(packed)
            self._engine_key, self._ctx_key = key, None
        return self._engine

    # ---- forward --------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, sample: torch.Tensor, timestep: Union[torch.Tensor, float, int], encoder_hidden_states: torch.Tensor,
                class_labels=None, attention_mask=None, return_dict: bool = True, use_first_frame_condition: bool = False,
                use_first_frame_condition_concat: bool = False, use_ip_cross_attention: bool = False, reference_images_latent=None,
                reference_images_clip_feat=None, use_camera_motion_condition=False, camera_movement_type_tensor=None,
                use_image_concat_training=False, use_text_encoder_2=False, encoder_hidden_states_2=None, use_fps_condition=False,
                fps_tensor=None, first_images_mask=None, flow_control=None):
        if use_first_frame_condition or use_text_encoder_2:
            raise NotImplementedError("use_first_frame_condition (through AnimationPipeline only) / use_text_encoder_2 are not available on the module's forward")
        if attention_mask is not None or class_labels is not None:
            raise NotImplementedError("attention_mask / class_labels are not used by the FollowYourClick path")
        eng = self._get_engine()
        cfg, o = self.engine_config, eng.ops
        if use_first_frame_condition_concat:
            # reference unet.py:580-586: the clean first-frame latents are repeated over the frames and concatenated to the sample;
            # the `sample / 2` behind conv_in (:589-590) is part of the packed conv_in weights of a model built with the option
            if not cfg.use_first_frame_condition_concat:
                raise ValueError("use_first_frame_condition_concat=True needs a model built with use_first_frame_condition_concat")
            if reference_images_latent is not None:
                sample = torch.cat((sample, reference_images_latent.unsqueeze(2).repeat(1, 1, sample.shape[2], 1, 1).to(sample)), dim=1)
        elif cfg.use_first_frame_condition_concat:
            raise ValueError("this model halves conv_in's output (built with use_first_frame_condition_concat): call it with use_first_frame_condition_concat=True")
        if use_camera_motion_condition and not cfg.use_camera_motion_condition:
            raise ValueError("use_camera_motion_condition=True needs a model built with use_camera_motion_condition")
        B, C, F, H, W = sample.shape
        if C != cfg.conv_in_channels:
            raise ValueError(f"sample has {C} channels, the model expects {cfg.conv_in_channels}")
        # (b,c,f,h,w) -> channels-last [b*f][h*w][pad64(c)]
        frames = sample.to(torch.float32).permute(0, 2, 1, 3, 4).reshape(B * F, C, H * W).contiguous()
        cp = pad_channels(C)
        x = eng.new(B * F * H * W, cp)
        o.nchw_to_nhwc(frames, x, N=B * F, C_=C, HW=H * W, c_pad=cp, scale=1.0)
        # conditioning: text (+ projected image tokens); cached while the same tensors come back every step
        ip_tokens = None
        if use_ip_cross_attention:
            if not cfg.use_ip_cross_attention:
                raise ValueError("use_ip_cross_attention=True needs a model built with use_ip_cross_attention")
            if self.image_proj_model is None:
                raise ValueError("set unet.image_proj_model (ip_adapter.init_proj()) before using IP cross-attention")
            ip_tokens = self.image_proj_model(reference_images_clip_feat)
        # The cached K / V^T are valid only for the very tensors they were projected from: compare by identity + version and
        # keep the tensors alive (an address can be recycled by a new prompt's embeddings), and make sure the engine's cache is
        # still ours (AnimationPipeline / DDIMSampler.prepare re-project it for their own runs).
        cur = (encoder_hidden_states, encoder_hidden_states._version,
               reference_images_clip_feat if ip_tokens is not None else None,
               reference_images_clip_feat._version if ip_tokens is not None else None)
        hit = (self._ctx_key is not None and self._ctx_key[0] is cur[0] and self._ctx_key[1] == cur[1]
               and self._ctx_key[2] is cur[2] and self._ctx_key[3] == cur[3] and eng.ctx_cache is self._ctx_cache_obj)
        if not hit:
            if encoder_hidden_states.shape[0] != B:
                raise ValueError("encoder_hidden_states batch does not match sample batch")
            eng.prepare_context(encoder_hidden_states.float(), None if ip_tokens is None else ip_tokens.float())
            self._ctx_key, self._ctx_cache_obj = cur, eng.ctx_cache

        def vec(v, n):
            if v is None:
                return None
            v = torch.as_tensor(v).reshape(-1).float().cpu().tolist()
            return v * n if len(v) == 1 else v

        t = vec(timestep, 1)
        if len(t) not in (1, B):
            raise ValueError(f"timestep has {len(t)} entries for a batch of {B}")
        fps = vec(fps_tensor, B) if (use_fps_condition and cfg.use_fps_condition) else None
        flow = vec(flow_control, B) if fps is not None else None
        camera = vec(camera_movement_type_tensor, B) if use_camera_motion_condition else None      # reference unet.py:498-508, 538-544
        if len(set(t)) == 1:
            _, temb = eng.prepare_time_embeddings([t[0]], fps, flow, B, camera=camera)
        else:       # per-sample timesteps (reference unet.py:488-521 broadcasts a (B,) tensor): one embedding row per batch element
            _, temb = eng.prepare_time_embeddings(t, fps, flow, 1, camera=camera)
        pred = eng.forward(x, temb, B, F, H, W)
        out = eng.new(B * F, cfg.out_channels, H * W, dtype=torch.float32)
        o.nhwc_to_nchw(pred, out, N=B * F, C_=cfg.out_channels, HW=H * W, ld=pred.shape[1])
        out = out.reshape(B, F, cfg.out_channels, H, W).permute(0, 2, 1, 3, 4).to(sample.dtype)
        if not return_dict:
            return (out,)
        return UNet3DConditionOutput(sample=out)

    # ---- loading ----------------------------------------------------------------------------------
    @classmethod
    def from_config(cls, config: dict, **kwargs):
        cfg = {k: v for k, v in dict(config).items() if not k.startswith("_")}
        cfg.update(kwargs)
        return cls(**cfg)

    @classmethod
    def from_pretrained_2d(cls, pretrained_model_path, subfolder=None, unet_additional_kwargs=None):
        """2-D SD-1.5 UNet weights -> 3-D model (reference unet.py:674-726): reads config.json +
        diffusion_pytorch_model.bin, zero-extends conv_in to the concat-conditioning channel count."""
        if subfolder is not None:
            pretrained_model_path = os.path.join(pretrained_model_path, subfolder)
        config_file = os.path.join(pretrained_model_path, "config.json")
        if not os.path.isfile(config_file):
            raise RuntimeError(f"{config_file} does not exist")
        with open(config_file, "r") as f:
            config = json.load(f)
        config["down_block_types"] = ["CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"]
        config["up_block_types"] = ["UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"]
        extra = dict(unet_additional_kwargs or {})
        model = cls.from_config(config, **extra)
        model_file = os.path.join(pretrained_model_path, "diffusion_pytorch_model.bin")
        if not os.path.isfile(model_file):
            raise RuntimeError(f"{model_file} does not exist")
        state_dict = torch.load(model_file, map_location="cpu")
        if extra.get("use_first_frame_condition_concat") or extra.get("use_first_frame_mask_condition_concat"):
            w = torch.zeros_like(model.conv_in.weight)
            w[:, :4] = state_dict["conv_in.weight"]
            state_dict["conv_in.weight"] = w
        m, u = model.load_state_dict(state_dict, strict=False)
        print(f"### missing keys: {len(m)}; \n### unexpected keys: {len(u)};")
        n_temporal = sum(p.numel() for n, p in model.named_parameters() if "temporal" in n)
        print(f"### Temporal Module Parameters: {n_temporal / 1e6} M")
        return model
