"""Generate tests/golden/* by executing the REAL reference (container only).

Run:  python -m oracle.make_golden      (needs /root/reference; ~1 min on 8 cores)

Outputs (all small, committed):
  schema_unet_full.json / schema_unet_tiny.json / schema_unet_tiny_ip.json / schema_vae.json
      parameter+buffer names -> shapes from the reference modules' state_dict()
  ddim.npz             alphas_cumprod table, timesteps for n=5/25/50, one scheduler.step()
  unet_tiny_fwd.npz    one UNet3DConditionModel.forward on seeded inputs/weights (tiny config)
  unet_tiny_ip_fwd.npz same with use_ip_cross_attention (reference CPU code path; see
                       UNetConfig.ip_reference_cpu_scale_quirk)
  vae_tiny.npz         AutoencoderKL.decode on a tiny decoder
  pipeline_tiny.npz    AnimationPipeline.__call__ end to end (5 DDIM steps, CFG 8, mask+first-frame
                       concat, fps/flow conditioning): per-step latents via `callback` and final video
Weights are NOT stored: they are re-derived from seeds by oracle/weights.py::make_weights.
"""
import json
import os

import numpy as np
import torch

from . import functional as Fn
from . import refshim, stubs
from . import weights as W

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

MM_KW = dict(num_attention_heads=8, num_transformer_block=1, attention_block_types=["Temporal_Self", "Temporal_Self"],
             temporal_position_encoding=True, temporal_position_encoding_max_len=24, temporal_attention_dim_div=1,
             zero_initialize=True)


def ref_unet(cfg: Fn.UNetConfig):
    from animatediff.models.unet import UNet3DConditionModel
    extra = {}
    if cfg.use_first_frame_condition_concat:
        extra["use_first_frame_condition_concat"] = True
    if cfg.use_camera_motion_condition:
        extra["use_camera_motion_condition"] = True
    return UNet3DConditionModel(
        sample_size=cfg.sample_size, in_channels=cfg.in_channels, out_channels=cfg.out_channels,
        block_out_channels=cfg.block_out_channels, layers_per_block=cfg.layers_per_block,
        cross_attention_dim=cfg.cross_attention_dim, attention_head_dim=cfg.attention_head_dim,
        norm_num_groups=cfg.norm_num_groups, norm_eps=cfg.norm_eps, act_fn="silu", use_linear_projection=False,
        use_motion_module=True, motion_module_resolutions=(1, 2, 4, 8), unet_use_cross_frame_attention=False,
        unet_use_temporal_attention=False, use_fps_condition=True,
        use_first_frame_mask_condition_concat=cfg.use_first_frame_mask_condition_concat and not cfg.use_first_frame_condition_concat,
        motion_module_type="Vanilla", motion_module_kwargs=dict(MM_KW, temporal_position_encoding_max_len=cfg.temporal_position_encoding_max_len),
        use_ip_cross_attention=cfg.use_ip_cross_attention, num_tokens=cfg.ip_num_tokens, scale=cfg.ip_scale,
        **extra)


def ref_vae(vcfg: Fn.VAEConfig):
    from diffusers.models.vae import AutoencoderKL
    n = len(vcfg.block_out_channels)
    return AutoencoderKL(in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * n,
                         up_block_types=("UpDecoderBlock2D",) * n, block_out_channels=vcfg.block_out_channels,
                         layers_per_block=vcfg.layers_per_block, latent_channels=vcfg.latent_channels,
                         norm_num_groups=vcfg.norm_num_groups, sample_size=64)


def dump_schema(name, sd):
    with open(os.path.join(OUT, name), "w") as f:
        json.dump({k: list(v.shape) for k, v in sd.items()}, f, indent=0)


def main():
    refshim.install()
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    from diffusers.schedulers.scheduling_ddim import DDIMScheduler
    from animatediff.pipelines.pipeline_animation import AnimationPipeline

    # ---- schemas --------------------------------------------------------------------------
    with torch.device("meta"):
        dump_schema("schema_unet_full.json", ref_unet(Fn.UNetConfig()).state_dict())
        dump_schema("schema_unet_full_ip.json", ref_unet(Fn.UNetConfig(use_ip_cross_attention=True, ip_num_tokens=16)).state_dict())
        vae_full = ref_vae(Fn.VAEConfig()).state_dict()
        dump_schema("schema_vae.json", {k: v for k, v in vae_full.items() if k.startswith(("decoder", "post_quant"))})

    # ---- scheduler ------------------------------------------------------------------------
    skw = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1,
               clip_sample=False, prediction_type="v_prediction", rescale_betas_zero_snr=True)
    sch = DDIMScheduler(**skw)
    d = {"alphas_cumprod": sch.alphas_cumprod.numpy()}
    for n in (5, 25, 50):
        sch.set_timesteps(n)
        d[f"timesteps_{n}"] = sch.timesteps.numpy()
    sch.set_timesteps(25)
    g = torch.Generator().manual_seed(11)
    x, v = torch.randn(1, 4, 2, 4, 4, generator=g), torch.randn(1, 4, 2, 4, 4, generator=g)
    d["step_sample"], d["step_model_output"] = x.numpy(), v.numpy()
    d["step_out_t961"] = sch.step(v, 961, x, eta=0.0).prev_sample.numpy()
    d["step_out_t1"] = sch.step(v, 1, x, eta=0.0).prev_sample.numpy()
    np.savez_compressed(os.path.join(OUT, "ddim.npz"), **d)

    # ---- UNet forward (tiny) --------------------------------------------------------------
    fps, flow = torch.tensor([2, 2]), torch.tensor([4, 4])
    for ip in (False, True):
        cfg = Fn.tiny_unet_config(use_ip_cross_attention=ip, ip_scale=0.7 if ip else 1.0, ip_reference_cpu_scale_quirk=ip)
        unet = ref_unet(cfg).eval()
        dump_schema("schema_unet_tiny_ip.json" if ip else "schema_unet_tiny.json", unet.state_dict())
        sd = W.make_weights(W.unet_state_shapes(cfg), seed=0)
        unet.load_state_dict(sd, strict=True)
        inp = W.seeded_inputs(cfg, 1, 4, 8, 8, seed=7)
        x9 = torch.cat([Fn.build_model_input(inp["latents"], inp["first_image_latents"], inp["first_images_mask"])] * 2)
        if ip:
            class Proj(torch.nn.Module):
                def forward(self, feat):
                    return feat
            unet.image_proj_model = Proj()
        with torch.no_grad():
            y = unet(x9, torch.tensor(961), inp["text"], use_fps_condition=True, fps_tensor=fps, flow_control=flow,
                     use_ip_cross_attention=ip, reference_images_clip_feat=inp["ip_tokens"] if ip else None).sample
        np.savez_compressed(os.path.join(OUT, "unet_tiny_ip_fwd.npz" if ip else "unet_tiny_fwd.npz"),
                            sample=x9.numpy(), timestep=np.int64(961), text=inp["text"].numpy(),
                            ip_tokens=inp["ip_tokens"].numpy(), fps=fps.numpy(), flow=flow.numpy(), out=y.numpy(),
                            weight_seed=np.int64(0), input_seed=np.int64(7))
        if not ip:
            unet_plain, sd_plain, cfg_plain = unet, sd, cfg
            # odd latent size (10x12 -> 5x6 -> 3x3 -> 2x2): exercises the forwarded `upsample_size` path
            go = torch.Generator().manual_seed(9)
            xo = torch.randn(2, 9, 2, 10, 12, generator=go)
            to = torch.randn(2, 77, cfg.cross_attention_dim, generator=go)
            with torch.no_grad():
                yo = unet(xo, torch.tensor(481), to, use_fps_condition=True, fps_tensor=fps, flow_control=flow).sample
            np.savez_compressed(os.path.join(OUT, "unet_tiny_odd_fwd.npz"), sample=xo.numpy(), timestep=np.int64(481), text=to.numpy(),
                                fps=fps.numpy(), flow=flow.numpy(), out=yo.numpy(), weight_seed=np.int64(0))

    # ---- VAE decode (tiny) ----------------------------------------------------------------
    vcfg = Fn.VAEConfig(block_out_channels=(64, 128, 128, 128))
    vae = ref_vae(vcfg).eval()
    dump_schema("schema_vae_tiny.json", {k: v for k, v in vae.state_dict().items() if k.startswith(("decoder", "post_quant"))})
    sdv = W.make_weights(W.vae_decoder_state_shapes(vcfg), seed=3)
    vae.load_state_dict(sdv, strict=False)
    z = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        img = vae.decode(z).sample
    np.savez_compressed(os.path.join(OUT, "vae_tiny.npz"), z=z.numpy(), out=img.numpy(), weight_seed=np.int64(3))
    # encoder half (first-frame conditioning front-end): moments of vae.encode on a 64x48 image
    dump_schema("schema_vae_enc_tiny.json", {k: v for k, v in vae.state_dict().items() if k.startswith(("encoder", "quant_conv"))})
    sde = W.make_weights(W.vae_encoder_state_shapes(vcfg), seed=4)
    vae.load_state_dict(sde, strict=False)
    img_in = torch.rand(2, 3, 64, 48, generator=torch.Generator().manual_seed(6)) * 2 - 1
    with torch.no_grad():
        post = vae.encode(img_in).latent_dist
    np.savez_compressed(os.path.join(OUT, "vae_enc_tiny.npz"), x=img_in.numpy(), moments=post.parameters.numpy(), weight_seed=np.int64(4))
    vae.load_state_dict(sdv, strict=False)

    # ---- full pipeline (tiny): the cfg1-shaped plumbing run -------------------------------
    tok, txt = stubs.FakeTokenizer(), stubs.StubTextEncoder(cfg_plain.cross_attention_dim)
    pipe = AnimationPipeline(vae=vae, text_encoder=txt, tokenizer=tok, unet=unet_plain, scheduler=DDIMScheduler(**skw))
    inp = W.seeded_inputs(cfg_plain, 1, 4, 8, 8, seed=21)
    traj = []
    out = pipe("a corgi waving its tail", video_length=4, height=64, width=64, num_inference_steps=5, guidance_scale=8.0,
               negative_prompt="blurry", latents=inp["latents"].clone(), first_image_latents=inp["first_image_latents"],
               first_images_mask=inp["first_images_mask"], use_first_frame_mask_condition_concat=True,
               use_fps_condition=True, fps_tensor=torch.tensor([2]), flow_control=torch.tensor([4]),
               callback=lambda i, t, l: traj.append(l.clone()), callback_steps=1)
    with torch.no_grad():
        text_emb = pipe._encode_prompt(["a corgi waving its tail"], "cpu", 1, True, ["blurry"])
    np.savez_compressed(os.path.join(OUT, "pipeline_tiny.npz"), latents=inp["latents"].numpy(),
                        first_image_latents=inp["first_image_latents"].numpy(),
                        first_images_mask=inp["first_images_mask"].numpy(), text_embeddings=text_emb.numpy(),
                        trajectory=torch.stack(traj).numpy(), videos=out.videos.numpy(),
                        unet_weight_seed=np.int64(0), vae_weight_seed=np.int64(3), input_seed=np.int64(21))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
