"""Generate tests/golden/sd2d_*.npz by executing the REAL reference's 2-D Stable Diffusion path (container only):
`diffusers.models.unet_2d_condition.UNet2DConditionModel` and
`diffusers.pipelines.stable_diffusion.StableDiffusionPipeline` - the first-image synthesis front-end of
scripts/inference.py:194-204, 300-306.

Run:  python -m oracle.make_golden_2d     (needs /root/reference)

  schema_unet2d_tiny.json  state_dict() names -> shapes of the reference 2-D UNet (tiny config)
  sd2d_unet_fwd.npz        UNet2DConditionModel.forward at 8x8 and at an odd 10x12 latent
  sd2d_pipeline.npz        StableDiffusionPipeline.__call__ (4 DDIM steps, epsilon prediction, scaled_linear betas,
                           set_alpha_to_one=False, CFG 8): per-step latents via `callback`, final images
Weights are re-derived from seeds by oracle/weights.py::make_weights (not stored).
"""
import dataclasses
import importlib.machinery as M
import os
import sys
import types

import numpy as np
import torch

from . import functional as Fn
from . import refshim, stubs
from . import weights as W
from .make_golden import OUT, dump_schema, ref_vae


def cfg_2d() -> Fn.UNetConfig:
    return Fn.tiny_unet_config(use_motion_module=False, use_fps_condition=False, use_first_frame_mask_condition_concat=False)


SCHED_2D = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                set_alpha_to_one=False, steps_offset=1, prediction_type="epsilon")


def install_sd_pipeline():
    """diffusers/pipelines/__init__.py imports every pipeline (and names that transformers 5 removed); import the one
    module we need through stub packages instead."""
    refshim.install()
    import transformers

    class _RemovedFeatureExtractor:  # transformers.CLIPFeatureExtractor (only used by the safety checker, which is off)
        pass

    if not hasattr(transformers, "CLIPFeatureExtractor"):
        transformers.CLIPFeatureExtractor = _RemovedFeatureExtractor
    root = refshim.REFERENCE_ROOT
    sys.modules["diffusers.pipelines"].__path__ = [os.path.join(root, "diffusers", "pipelines")]
    name = "diffusers.pipelines.stable_diffusion"
    if name not in sys.modules:
        sd = types.ModuleType(name)
        sd.__path__ = [os.path.join(root, "diffusers", "pipelines", "stable_diffusion")]
        sd.__spec__ = M.ModuleSpec(name, None, is_package=True)

        @dataclasses.dataclass
        class StableDiffusionPipelineOutput:  # diffusers/pipelines/stable_diffusion/__init__.py:16-30
            images: object
            nsfw_content_detected: object

        sd.StableDiffusionPipelineOutput = StableDiffusionPipelineOutput
        sys.modules[name] = sd
    from diffusers.pipelines.stable_diffusion.pipeline_stable_diffusion import StableDiffusionPipeline
    return StableDiffusionPipeline


def ref_unet2d(cfg: Fn.UNetConfig):
    from diffusers.models.unet_2d_condition import UNet2DConditionModel
    return UNet2DConditionModel(sample_size=cfg.sample_size, in_channels=cfg.in_channels, out_channels=cfg.out_channels,
                                block_out_channels=cfg.block_out_channels, layers_per_block=cfg.layers_per_block,
                                cross_attention_dim=cfg.cross_attention_dim, attention_head_dim=cfg.attention_head_dim,
                                norm_num_groups=cfg.norm_num_groups, norm_eps=cfg.norm_eps)


def main():
    SDPipe = install_sd_pipeline()
    from diffusers.schedulers.scheduling_ddim import DDIMScheduler
    os.makedirs(OUT, exist_ok=True)
    cfg = cfg_2d()
    unet = ref_unet2d(cfg).eval()
    dump_schema("schema_unet2d_tiny.json", unet.state_dict())
    sd = W.make_weights(W.unet_state_shapes(cfg), seed=12)
    unet.load_state_dict(sd, strict=True)

    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 4, 8, 8, generator=g)
    text = torch.randn(2, 77, cfg.cross_attention_dim, generator=g)
    xo = torch.randn(2, 4, 10, 12, generator=g)
    with torch.no_grad():
        y = unet(x, torch.tensor(481), text).sample
        yo = unet(xo, torch.tensor(21), text).sample
    np.savez_compressed(os.path.join(OUT, "sd2d_unet_fwd.npz"), sample=x.numpy(), text=text.numpy(), timestep=np.int64(481),
                        out=y.numpy(), sample_odd=xo.numpy(), timestep_odd=np.int64(21), out_odd=yo.numpy(), weight_seed=np.int64(12))

    vcfg = Fn.VAEConfig(block_out_channels=(64, 128, 128, 128))
    vae = ref_vae(vcfg).eval()
    vae.load_state_dict(W.make_weights(W.vae_decoder_state_shapes(vcfg), seed=3), strict=False)
    tok, txt = stubs.FakeTokenizer(), stubs.StubTextEncoder(cfg.cross_attention_dim)
    pipe = SDPipe(vae=vae, text_encoder=txt, tokenizer=tok, unet=unet, scheduler=DDIMScheduler(**SCHED_2D), safety_checker=None,
                  feature_extractor=None, requires_safety_checker=False)
    lat = torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(41))
    traj = []
    out = pipe("a corgi on the beach", height=64, width=64, num_inference_steps=4, guidance_scale=8.0, negative_prompt="blurry",
               latents=lat.clone(), output_type="np", callback=lambda i, t, l: traj.append(l.clone()), callback_steps=1)
    with torch.no_grad():
        text_emb = pipe._encode_prompt("a corgi on the beach", "cpu", 1, True, "blurry")
    np.savez_compressed(os.path.join(OUT, "sd2d_pipeline.npz"), latents=lat.numpy(), text_embeddings=text_emb.numpy(),
                        trajectory=torch.stack(traj).numpy(), images=np.asarray(out.images), unet_weight_seed=np.int64(12),
                        vae_weight_seed=np.int64(3))
    for f in ("schema_unet2d_tiny.json", "sd2d_unet_fwd.npz", "sd2d_pipeline.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
