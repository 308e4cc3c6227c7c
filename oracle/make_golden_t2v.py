"""Generate tests/golden/pipeline_tiny_t2v.npz: the REAL reference's AnimationPipeline.__call__ in plain text-to-video mode
(no mask / first-frame concat, no fps conditioning: how scripts/inference_org.py:265-289 and scripts/animate.py drive it),
UNet3D with 4 input channels and motion modules.  Run:  python -m oracle.make_golden_t2v"""
import os

import numpy as np
import torch

from . import functional as Fn
from . import refshim, stubs
from . import weights as W
from .make_golden import MM_KW, OUT, ref_vae


def cfg_t2v() -> Fn.UNetConfig:
    return Fn.tiny_unet_config(use_fps_condition=False, use_first_frame_mask_condition_concat=False)


def main():
    refshim.install()
    from animatediff.models.unet import UNet3DConditionModel
    from animatediff.pipelines.pipeline_animation import AnimationPipeline
    from diffusers.schedulers.scheduling_ddim import DDIMScheduler
    cfg = cfg_t2v()
    unet = UNet3DConditionModel(
        sample_size=cfg.sample_size, in_channels=4, out_channels=4, block_out_channels=cfg.block_out_channels,
        layers_per_block=cfg.layers_per_block, cross_attention_dim=cfg.cross_attention_dim, attention_head_dim=cfg.attention_head_dim,
        use_motion_module=True, motion_module_resolutions=(1, 2, 4, 8), unet_use_cross_frame_attention=False,
        unet_use_temporal_attention=False, motion_module_type="Vanilla", motion_module_kwargs=dict(MM_KW)).eval()
    unet.load_state_dict(W.make_weights(W.unet_state_shapes(cfg), seed=8), strict=True)
    vcfg = Fn.VAEConfig(block_out_channels=(64, 128, 128, 128))
    vae = ref_vae(vcfg).eval()
    vae.load_state_dict(W.make_weights(W.vae_decoder_state_shapes(vcfg), seed=3), strict=False)
    skw = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False)
    pipe = AnimationPipeline(vae=vae, text_encoder=stubs.StubTextEncoder(cfg.cross_attention_dim), tokenizer=stubs.FakeTokenizer(),
                             unet=unet, scheduler=DDIMScheduler(**skw))
    lat = torch.randn(1, 4, 4, 8, 8, generator=torch.Generator().manual_seed(61))
    traj = []
    out = pipe("a corgi waving its tail", video_length=4, height=64, width=64, num_inference_steps=4, guidance_scale=7.5,
               negative_prompt="blurry", latents=lat.clone(), callback=lambda i, t, l: traj.append(l.clone()), callback_steps=1)
    with torch.no_grad():
        text_emb = pipe._encode_prompt(["a corgi waving its tail"], "cpu", 1, True, ["blurry"])
    # video_scale > 0: extra per-frame unconditional UNet pass and three-way guidance (scripts/inference_org.py --video_scale)
    traj_vs = []
    pipe("a corgi waving its tail", video_length=4, height=64, width=64, num_inference_steps=3, guidance_scale=7.5,
         negative_prompt="blurry", latents=lat.clone(), video_scale=0.7, callback=lambda i, t, l: traj_vs.append(l.clone()), callback_steps=1)
    # use_first_frame_condition: frame 0 pinned to the clean first-frame latents, timestep-0 embedding for that frame
    first = 0.18215 * torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(62))
    traj_ff = []
    pipe("a corgi waving its tail", video_length=4, height=64, width=64, num_inference_steps=3, guidance_scale=7.5,
         negative_prompt="blurry", latents=lat.clone(), use_first_frame_condition=True, first_image_latents=first,
         callback=lambda i, t, l: traj_ff.append(l.clone()), callback_steps=1)
    np.savez_compressed(os.path.join(OUT, "pipeline_tiny_t2v.npz"), latents=lat.numpy(), text_embeddings=text_emb.numpy(),
                        first_image_latents=first.numpy(), trajectory_first_frame=torch.stack(traj_ff).numpy(),
                        trajectory=torch.stack(traj).numpy(), videos=out.videos.numpy(), unet_weight_seed=np.int64(8),
                        vae_weight_seed=np.int64(3), trajectory_video_scale=torch.stack(traj_vs).numpy(), video_scale=np.float32(0.7))
    print("pipeline_tiny_t2v.npz", os.path.getsize(os.path.join(OUT, "pipeline_tiny_t2v.npz")))


if __name__ == "__main__":
    main()
