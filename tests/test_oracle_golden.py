"""Pins the CPU oracle (oracle/functional.py) against vectors produced by the REAL reference
(tests/golden/*, written by oracle/make_golden.py in the build container).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import functional as Fn
from oracle import weights as W

TOL = 2e-5  # fp32 re-association noise between two CPU op orders; outputs are O(1)


def _load(golden_dir, name):
    return {k: torch.from_numpy(v) if v.shape else v for k, v in np.load(os.path.join(golden_dir, name)).items()}


def _schema(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


def test_schema_unet_full(golden_dir):
    ref = _schema(golden_dir, "schema_unet_full.json")
    mine = W.unet_state_shapes(Fn.UNetConfig())
    assert list(mine.keys()).__len__() == 1254
    assert {k: tuple(v) for k, v in mine.items()} == ref


def test_schema_unet_full_ip(golden_dir):
    ref = _schema(golden_dir, "schema_unet_full_ip.json")
    mine = W.unet_state_shapes(Fn.UNetConfig(use_ip_cross_attention=True, ip_num_tokens=16))
    assert {k: tuple(v) for k, v in mine.items()} == ref


def test_schema_tiny_and_vae(golden_dir):
    assert {k: tuple(v) for k, v in W.unet_state_shapes(Fn.tiny_unet_config()).items()} == _schema(golden_dir, "schema_unet_tiny.json")
    assert {k: tuple(v) for k, v in W.unet_state_shapes(Fn.tiny_unet_config(use_ip_cross_attention=True)).items()} == \
        _schema(golden_dir, "schema_unet_tiny_ip.json")
    assert {k: tuple(v) for k, v in W.vae_decoder_state_shapes(Fn.VAEConfig()).items()} == _schema(golden_dir, "schema_vae.json")


def test_ddim_tables_and_step(golden_dir):
    g = _load(golden_dir, "ddim.npz")
    c = Fn.DDIMConfig()
    abar = Fn.ddim_alphas_cumprod(c)
    assert torch.equal(abar, g["alphas_cumprod"])          # same fp32 op sequence -> bit exact
    assert abar[999].item() == 0.0
    for n in (5, 25, 50):
        assert torch.equal(Fn.ddim_timesteps(c, n), g[f"timesteps_{n}"])
    assert Fn.ddim_timesteps(c, 25)[0].item() == 961
    for t in (961, 1):
        out = Fn.ddim_step(c, abar, 25, g["step_model_output"], t, g["step_sample"])
        assert torch.allclose(out, g[f"step_out_t{t}"], atol=1e-6, rtol=0)


def test_unet_forward_tiny(golden_dir):
    g = _load(golden_dir, "unet_tiny_fwd.npz")
    cfg = Fn.tiny_unet_config()
    sd = W.make_weights(W.unet_state_shapes(cfg), int(g["weight_seed"]))
    with torch.no_grad():
        y = Fn.unet3d_forward(sd, cfg, g["sample"], torch.tensor(int(g["timestep"])), g["text"], g["fps"], g["flow"])
    assert y.shape == g["out"].shape
    assert (y - g["out"]).abs().max().item() < TOL


def test_unet_forward_tiny_ip(golden_dir):
    g = _load(golden_dir, "unet_tiny_ip_fwd.npz")
    cfg = Fn.tiny_unet_config(use_ip_cross_attention=True, ip_scale=0.7, ip_reference_cpu_scale_quirk=True)
    sd = W.make_weights(W.unet_state_shapes(cfg), int(g["weight_seed"]))
    with torch.no_grad():
        y = Fn.unet3d_forward(sd, cfg, g["sample"], torch.tensor(961), g["text"], g["fps"], g["flow"], g["ip_tokens"])
    assert (y - g["out"]).abs().max().item() < TOL


@pytest.mark.parametrize("ip", [False, True])
def test_unet_forward_full_width(golden_dir, ip):
    """SD-1.5 widths (320/640/1280/1280, ctx 768, head dims 40/80/160), F=4, 16x16 latent, the real reference's f32 output
    (oracle/make_golden_full.py small / ip).  The IP golden is of the reference's CPU code path (attn2 temperature quirk)."""
    g = _load(golden_dir, "unet_full_ip_fwd.npz" if ip else "unet_full_small_fwd.npz")
    cfg = Fn.UNetConfig(use_ip_cross_attention=True, ip_num_tokens=16, ip_scale=0.7, ip_reference_cpu_scale_quirk=True) if ip else Fn.UNetConfig()
    sd = W.make_weights(W.unet_state_shapes(cfg), int(g["weight_seed"]))
    inp = W.seeded_inputs(cfg, 1, int(g["frames"]), int(g["h"]), int(g["w"]), seed=int(g["input_seed"]))
    x9 = torch.cat([Fn.build_model_input(inp["latents"], inp["first_image_latents"], inp["first_images_mask"])] * 2)
    with torch.no_grad():
        y = Fn.unet3d_forward(sd, cfg, x9, torch.tensor(int(g["timestep"])), inp["text"], g["fps"], g["flow"], inp["ip_tokens"] if ip else None)
    ref = g["out_f32"]
    assert (torch.linalg.norm(y - ref) / torch.linalg.norm(ref)).item() < 2e-5
    assert (y - ref).abs().max().item() < 1e-4


def test_cfg5_ip_mask_trajectory_full_width(golden_dir):
    """BASELINE configs[4] as a trajectory (IP-Adapter branch + rectangle region mask + first-frame concat, CFG 8, 5 DDIM steps, full
    widths): the oracle's denoise loop WITH the reference's CPU-path temperature quirk against AnimationPipeline.__call__ of the real
    reference at every step (oracle/make_golden_full.py cfg5); the same file holds the oracle WITHOUT the quirk, which
    tests/test_fullwidth_gpu.py holds the engine to - checked here to be what the oracle produces, so the two tests meet."""
    g = _load(golden_dir, "cfg5_trajectory.npz")
    steps, F, lat = int(g["steps"]), int(g["frames"]), int(g["lat"])
    kw = dict(use_ip_cross_attention=True, ip_num_tokens=int(g["ip_num_tokens"]), ip_scale=float(g["ip_scale"]))
    cfg_q = Fn.UNetConfig(ip_reference_cpu_scale_quirk=True, **kw)
    sd = W.make_weights(W.unet_state_shapes(cfg_q), int(g["weight_seed"]))
    inp = W.seeded_inputs(cfg_q, 1, F, lat, lat, seed=int(g["input_seed"]))
    for cfg, key, tol in ((cfg_q, "ref_f32", 2e-4), (Fn.UNetConfig(**kw), "oracle_noquirk", 1e-5)):      # (the oracle against its own stored run: 0 with the thread count of the run that stored it, ~2e-6 with another - oneDNN splits reductions by thread count)
        traj = {}
        with torch.no_grad():
            Fn.denoise(sd, cfg, Fn.DDIMConfig(), inp["latents"].clone(), g["text_embeddings"], steps, 8.0, inp["first_image_latents"], g["first_images_mask"],
                       torch.tensor([2]), torch.tensor([4]), ip_tokens=inp["ip_tokens"], callback=lambda i, t, l: traj.__setitem__(i, l.clone()))
        for i in range(steps):
            ref = g[f"step{i}_{key}"]
            r = (torch.linalg.norm(traj[i] - ref) / torch.linalg.norm(ref)).item()
            assert r < tol, (key, i, r)
        if key == "ref_f32":      # the two semantics really differ (otherwise this golden would pin nothing about the quirk)
            assert (torch.linalg.norm(g[f"step{steps - 1}_oracle_noquirk"] - ref) / torch.linalg.norm(ref)).item() > 0.1


def test_vae_decode_tiny(golden_dir):
    g = _load(golden_dir, "vae_tiny.npz")
    vcfg = Fn.VAEConfig(block_out_channels=(64, 128, 128, 128))
    sd = W.make_weights(W.vae_decoder_state_shapes(vcfg), int(g["weight_seed"]))
    with torch.no_grad():
        y = Fn.vae_decode(sd, vcfg, g["z"])
    assert (y - g["out"]).abs().max().item() < TOL


def test_pipeline_trajectory_tiny(golden_dir):
    """Oracle denoise loop + decode vs AnimationPipeline.__call__ of the real reference."""
    g = _load(golden_dir, "pipeline_tiny.npz")
    cfg = Fn.tiny_unet_config()
    sd = W.make_weights(W.unet_state_shapes(cfg), int(g["unet_weight_seed"]))
    traj = []
    with torch.no_grad():
        lat = Fn.denoise(sd, cfg, Fn.DDIMConfig(), g["latents"], g["text_embeddings"], 5, 8.0,
                         g["first_image_latents"], g["first_images_mask"], torch.tensor([2]), torch.tensor([4]),
                         callback=lambda i, t, l: traj.append(l.clone()))
    traj = torch.stack(traj)
    assert traj.shape == g["trajectory"].shape
    err = (traj - g["trajectory"]).abs().amax(dim=(1, 2, 3, 4, 5))
    assert err.max().item() < 1e-4, err
    vcfg = Fn.VAEConfig(block_out_channels=(64, 128, 128, 128))
    sdv = W.make_weights(W.vae_decoder_state_shapes(vcfg), int(g["vae_weight_seed"]))
    with torch.no_grad():
        vid = Fn.decode_latents(sdv, vcfg, lat)
    assert (vid - g["videos"]).abs().max().item() < 1e-4


def test_unet_forward_odd_size_upsample_forwarding(golden_dir):
    """latent 10x12 -> 5x6 -> 3x3 -> 2x2: Upsample3D gets the skip's size instead of scale_factor=2"""
    g = _load(golden_dir, "unet_tiny_odd_fwd.npz")
    cfg = Fn.tiny_unet_config()
    sd = W.make_weights(W.unet_state_shapes(cfg), int(g["weight_seed"]))
    with torch.no_grad():
        y = Fn.unet3d_forward(sd, cfg, g["sample"], torch.tensor(int(g["timestep"])), g["text"], g["fps"], g["flow"])
    assert (y - g["out"]).abs().max().item() < TOL


def test_vae_encode_tiny(golden_dir):
    g = _load(golden_dir, "vae_enc_tiny.npz")
    vcfg = Fn.VAEConfig(block_out_channels=(64, 128, 128, 128))
    assert {k: tuple(v) for k, v in W.vae_encoder_state_shapes(vcfg).items()} == _schema(golden_dir, "schema_vae_enc_tiny.json")
    sd = W.make_weights(W.vae_encoder_state_shapes(vcfg), int(g["weight_seed"]))
    with torch.no_grad():
        m = Fn.vae_encode_moments(sd, vcfg, g["x"])
    assert (m - g["moments"]).abs().max().item() < TOL


# ---- 2-D Stable Diffusion first-image path: the reference's UNet2DConditionModel / StableDiffusionPipeline ----
def _cfg2d():
    return Fn.tiny_unet_config(use_motion_module=False, use_fps_condition=False, use_first_frame_mask_condition_concat=False)


SCHED_2D = dict(beta_schedule="scaled_linear", set_alpha_to_one=False, prediction_type="epsilon", rescale_betas_zero_snr=False)


def test_schema_unet2d(golden_dir):
    ref = _schema(golden_dir, "schema_unet2d_tiny.json")
    mine = {k: tuple(v) for k, v in W.unet_state_shapes(_cfg2d()).items()}
    assert mine == ref


def test_unet2d_forward(golden_dir):
    g = _load(golden_dir, "sd2d_unet_fwd.npz")
    cfg = _cfg2d()
    sd = W.make_weights(W.unet_state_shapes(cfg), int(g["weight_seed"]))
    for s, t, o in (("sample", "timestep", "out"), ("sample_odd", "timestep_odd", "out_odd")):
        x = g[s][:, :, None]                          # (B,4,H,W) -> one-frame clip (B,4,1,H,W)
        y = Fn.unet3d_forward(sd, cfg, x, torch.tensor(int(g[t])), g["text"])[:, :, 0]
        assert torch.allclose(y, g[o], atol=2e-5, rtol=1e-4), (y - g[o]).abs().max()


def test_sd2d_pipeline_trajectory(golden_dir):
    g = _load(golden_dir, "sd2d_pipeline.npz")
    cfg = _cfg2d()
    sd = W.make_weights(W.unet_state_shapes(cfg), int(g["unet_weight_seed"]))
    sched = Fn.DDIMConfig(**SCHED_2D)
    traj = []
    lat = Fn.denoise(sd, cfg, sched, g["latents"][:, :, None], g["text_embeddings"], 4, 8.0,
                     callback=lambda i, t, l: traj.append(l[:, :, 0].clone()))
    traj = torch.stack(traj)          # random weights + epsilon prediction: latents grow to |x| ~ 20, so compare relatively
    err = (traj - g["trajectory"]).flatten(1).norm(dim=1) / g["trajectory"].flatten(1).norm(dim=1)
    assert err.max().item() < 2e-5, err
    vcfg = Fn.VAEConfig(block_out_channels=(64, 128, 128, 128))
    sdv = W.make_weights(W.vae_decoder_state_shapes(vcfg), int(g["vae_weight_seed"]))
    img = (Fn.vae_decode(sdv, vcfg, lat[:, :, 0] / 0.18215) / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1)
    assert torch.allclose(img, g["images"], atol=2e-5), (img - g["images"]).abs().max()


def test_pipeline_trajectory_plain_text_to_video(golden_dir):
    """AnimationPipeline without the concat conditioning (scripts/inference_org.py mode): 4-channel UNet3D, epsilon prediction"""
    g = _load(golden_dir, "pipeline_tiny_t2v.npz")
    cfg = Fn.tiny_unet_config(use_fps_condition=False, use_first_frame_mask_condition_concat=False)
    sd = W.make_weights(W.unet_state_shapes(cfg), int(g["unet_weight_seed"]))
    sched = Fn.DDIMConfig(prediction_type="epsilon", rescale_betas_zero_snr=False)
    traj = []
    Fn.denoise(sd, cfg, sched, g["latents"], g["text_embeddings"], 4, 7.5, callback=lambda i, t, l: traj.append(l.clone()))
    traj = torch.stack(traj)          # random weights + epsilon prediction: latents grow to |x| ~ 20, so compare relatively
    err = (traj - g["trajectory"]).flatten(1).norm(dim=1) / g["trajectory"].flatten(1).norm(dim=1)
    assert err.max().item() < 2e-5, err


def test_pipeline_trajectory_video_scale(golden_dir):
    """video_scale > 0 (scripts/inference_org.py --video_scale): extra per-frame pass with the reference's text-batch construction"""
    g = _load(golden_dir, "pipeline_tiny_t2v.npz")
    cfg = Fn.tiny_unet_config(use_fps_condition=False, use_first_frame_mask_condition_concat=False)
    sd = W.make_weights(W.unet_state_shapes(cfg), int(g["unet_weight_seed"]))
    sched = Fn.DDIMConfig(prediction_type="epsilon", rescale_betas_zero_snr=False)
    traj = []
    Fn.denoise(sd, cfg, sched, g["latents"], g["text_embeddings"], 3, 7.5, callback=lambda i, t, l: traj.append(l.clone()),
               video_scale=float(g["video_scale"]))
    traj = torch.stack(traj)
    ref = g["trajectory_video_scale"]
    err = (traj - ref).flatten(1).norm(dim=1) / ref.flatten(1).norm(dim=1)
    assert err.max().item() < 2e-5, err


def test_pipeline_trajectory_first_frame_condition(golden_dir):
    """use_first_frame_condition: frame 0 pinned to the first-frame latents, ResNets give frame 0 the timestep-0 embedding"""
    g = _load(golden_dir, "pipeline_tiny_t2v.npz")
    cfg = Fn.tiny_unet_config(use_fps_condition=False, use_first_frame_mask_condition_concat=False)
    sd = W.make_weights(W.unet_state_shapes(cfg), int(g["unet_weight_seed"]))
    sched = Fn.DDIMConfig(prediction_type="epsilon", rescale_betas_zero_snr=False)
    traj = []
    Fn.denoise(sd, cfg, sched, g["latents"], g["text_embeddings"], 3, 7.5, first_image_latents=g["first_image_latents"],
               callback=lambda i, t, l: traj.append(l.clone()), use_first_frame_condition=True)
    traj, ref = torch.stack(traj), g["trajectory_first_frame"]
    err = (traj - ref).flatten(1).norm(dim=1) / ref.flatten(1).norm(dim=1)
    assert err.max().item() < 2e-5, err
