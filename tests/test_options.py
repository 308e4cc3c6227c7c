"""Reference options outside the shipped YAMLs that the engine implements since round 5 (round-4 review, "missing" item 5): stochastic
DDIM (`eta > 0`, `use_clipped_model_output`: diffusers/schedulers/scheduling_ddim.py:336-365), the camera-motion embedding
(animatediff/models/unet.py:134-137, 538-544) and `use_first_frame_condition_concat` (:580-590; pipeline_animation.py:705-706).
Golden vectors: tests/golden/options_tiny.npz + schema_unet_tiny_{camera,concat}.json, produced by oracle/make_golden_options.py from
the REAL reference.  CPU: the oracle, the engine on the op emulator and the drop-in API against them; `-m gpu`: the HIP kernels."""
import json
import os

import numpy as np
import pytest
import torch

from emu_ops import EmuOps
from followyourclick_amd.engine import DDIMConfig, UNet3DConfig
from followyourclick_amd.engine.sampler import DDIMSampler
from followyourclick_amd.engine.scheduler import DDIMTables
from followyourclick_amd.engine.schema import unet_schema
from followyourclick_amd.engine.unet3d import UNet3DEngine
from followyourclick_amd.engine.weights import pack_unet, pad_channels
from oracle import functional as Fn
from oracle import weights as W

TINY = dict(block_out_channels=(64, 128, 256, 256), cross_attention_dim=64, sample_size=8)
SCHED_CASES = (("v_eta", dict(), dict(eta=0.7)), ("v_eta_clipped", dict(clip_sample=True), dict(eta=0.7, use_clipped_model_output=True)),
               ("eps_eta", dict(prediction_type="epsilon"), dict(eta=0.35)), ("v_clipped_eta0", dict(clip_sample=True), dict(use_clipped_model_output=True)))


@pytest.fixture(scope="module")
def g(golden_dir):
    return {k: torch.from_numpy(v) if v.shape else v for k, v in np.load(os.path.join(golden_dir, "options_tiny.npz")).items()}


def rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


# ---- the oracle (CPU restatement) is pinned to the reference's outputs ---------------------------------------------------------------
def test_oracle_ddim_step_with_eta_and_clipped_model_output(g):
    x, v, noise = g["sched_sample"], g["sched_model_output"], g["sched_noise"]
    for tag, ckw, skw in SCHED_CASES:
        c = Fn.DDIMConfig(**ckw)
        abar = Fn.ddim_alphas_cumprod(c)
        for t in (961, 481, 41):
            out = Fn.ddim_step(c, abar, 25, v, t, x, eta=skw.get("eta", 0.0), variance_noise=noise, use_clipped_model_output=skw.get("use_clipped_model_output", False))
            assert rel(out, g[f"sched_{tag}_{t}"]) < 2e-6, (tag, t)


def test_oracle_option_schemas_and_forwards(golden_dir, g):
    for name, cfg in (("schema_unet_tiny_camera.json", Fn.tiny_unet_config(use_camera_motion_condition=True)),
                      ("schema_unet_tiny_concat.json", Fn.tiny_unet_config(use_first_frame_condition_concat=True, use_first_frame_mask_condition_concat=False))):
        with open(os.path.join(golden_dir, name)) as f:
            ref = {k: tuple(v) for k, v in json.load(f).items()}
        assert {k: tuple(v) for k, v in W.unet_state_shapes(cfg).items()} == ref, name
    fps, flow = torch.tensor([2, 2]), torch.tensor([4, 4])
    ccfg = Fn.tiny_unet_config(use_camera_motion_condition=True)
    inp = W.seeded_inputs(ccfg, 1, 2, 8, 8, seed=int(g["camera_input_seed"]))
    x9 = torch.cat([Fn.build_model_input(inp["latents"], inp["first_image_latents"], inp["first_images_mask"])] * 2)
    with torch.no_grad():
        y = Fn.unet3d_forward(W.make_weights(W.unet_state_shapes(ccfg), 0), ccfg, x9, torch.tensor(int(g["camera_timestep"])), inp["text"], fps, flow, camera=g["camera_type"])
    assert rel(y, g["camera_out"]) < 2e-5
    kcfg = Fn.tiny_unet_config(use_first_frame_condition_concat=True, use_first_frame_mask_condition_concat=False)
    inp = W.seeded_inputs(kcfg, 1, 2, 8, 8, seed=int(g["concat_input_seed"]))
    with torch.no_grad():
        y = Fn.unet3d_forward(W.make_weights(W.unet_state_shapes(kcfg), 0), kcfg, torch.cat([inp["latents"]] * 2), torch.tensor(int(g["concat_timestep"])), inp["text"], fps, flow,
                              reference_images_latent=torch.cat([inp["first_image_latents"]] * 2))
    assert rel(y, g["concat_out"]) < 2e-5


# ---- the engine's host side + op schedule on the op emulator --------------------------------------------------------------------------
def test_engine_schemas_of_the_option_models(golden_dir):
    for name, cfg in (("schema_unet_tiny_camera.json", UNet3DConfig(use_camera_motion_condition=True, **TINY)),
                      ("schema_unet_tiny_concat.json", UNet3DConfig(use_first_frame_condition_concat=True, use_first_frame_mask_condition_concat=False, **TINY))):
        with open(os.path.join(golden_dir, name)) as f:
            ref = {k: tuple(v) for k, v in json.load(f).items()}
        assert dict(unet_schema(cfg)) == ref, name


def test_scheduler_tables_with_eta(g):
    """coefficients + the fused update (emulator) = the reference's DDIMScheduler.step for every option combination"""
    x, v, noise = g["sched_sample"], g["sched_model_output"], g["sched_noise"]
    B, C, F, H, Wd = x.shape
    o = EmuOps()
    for tag, ckw, skw in SCHED_CASES:
        tb = DDIMTables(DDIMConfig(**ckw))
        for t in (961, 481, 41):
            coef = torch.tensor(tb.step_coefficients(t, 25, skw.get("eta", 0.0)))
            lat = x.clone()
            pred = v.permute(0, 2, 3, 4, 1).reshape(B * F * H * Wd, C).contiguous()       # channels-last prediction, no CFG
            o.cfg_ddim_step(pred, lat, coef, B=B, F=F, HW=H * Wd, c_latent=C, ld=C, cfg=False, guidance=1.0, pred_type=tb.pred_type,
                            clip_sample=bool(ckw.get("clip_sample", False)), variance_noise=noise if skw.get("eta", 0) > 0 else None,
                            sigma=float(coef[4]), clipped_model_output=skw.get("use_clipped_model_output", False))
            assert rel(lat, g[f"sched_{tag}_{t}"]) < 2e-6, (tag, t)
    assert DDIMTables(DDIMConfig()).step_coefficients(481, 25)[4] == 0.0          # eta = 0: no noise term


def test_sampler_with_eta_matches_the_reference_pipeline(g):
    """AnimationPipeline.__call__(eta=0.6, generator=seeded) over 3 steps: the sampler draws the step noise exactly as the reference's
    scheduler does (torch.randn(shape, generator=generator, device, dtype)), so the stochastic trajectory reproduces"""
    cfg = Fn.tiny_unet_config()
    sd = W.make_weights(W.unet_state_shapes(cfg), 0)
    eng = UNet3DEngine(pack_unet(sd, UNet3DConfig(**TINY), torch.float32, "cpu"), ops=EmuOps())
    frames, lat, steps = int(g["pipe_eta_frames"]), int(g["pipe_eta_lat"]), int(g["pipe_eta_steps"])
    inp = W.seeded_inputs(cfg, 1, frames, lat, lat, seed=int(g["pipe_eta_input_seed"]))
    traj = []
    DDIMSampler(eng, DDIMConfig()).sample(inp["latents"], g["pipe_eta_text_embeddings"], steps, 8.0, inp["first_image_latents"], inp["first_images_mask"],
                                          fps=[2], flow=[4], eta=float(g["pipe_eta"]), generator=torch.Generator().manual_seed(int(g["pipe_eta_generator_seed"])),
                                          callback=lambda i, t, l: traj.append(l.clone()))
    for i in range(steps):
        assert rel(traj[i], g[f"pipe_eta_step{i}"]) < 5e-4, i
    # and the oracle's loop with the same noises
    gen = torch.Generator().manual_seed(int(g["pipe_eta_generator_seed"]))
    noises = [torch.randn(inp["latents"].shape, generator=gen) for _ in range(steps)]
    got = {}
    with torch.no_grad():
        Fn.denoise(sd, cfg, Fn.DDIMConfig(), inp["latents"].clone(), g["pipe_eta_text_embeddings"], steps, 8.0, inp["first_image_latents"], inp["first_images_mask"],
                   torch.tensor([2]), torch.tensor([4]), eta=float(g["pipe_eta"]), variance_noises=noises, callback=lambda i, t, l: got.__setitem__(i, l.clone()))
    for i in range(steps):
        assert rel(got[i], g[f"pipe_eta_step{i}"]) < 2e-5, i


def _nhwc(x, dtype=torch.float32):
    B, C, F, H, Wd = x.shape
    out = torch.zeros(B * F * H * Wd, pad_channels(C))
    out[:, :C] = x.permute(0, 2, 3, 4, 1).reshape(-1, C)
    return out.to(dtype)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 6e-2)])
def test_engine_camera_motion_forward(g, dtype, tol):
    ccfg = Fn.tiny_unet_config(use_camera_motion_condition=True)
    sd = W.make_weights(W.unet_state_shapes(ccfg), 0)
    eng = UNet3DEngine(pack_unet(sd, UNet3DConfig(use_camera_motion_condition=True, **TINY), dtype, "cpu"), ops=EmuOps())
    inp = W.seeded_inputs(ccfg, 1, 2, 8, 8, seed=int(g["camera_input_seed"]))
    x9 = torch.cat([Fn.build_model_input(inp["latents"], inp["first_image_latents"], inp["first_images_mask"])] * 2)
    eng.prepare_context(inp["text"])
    _, temb = eng.prepare_time_embeddings([int(g["camera_timestep"])], [2, 2], [4, 4], 2, camera=g["camera_type"].tolist())
    out = eng.forward(_nhwc(x9, dtype), temb, 2, 2, 8, 8).float().reshape(2, 2, 8, 8, 4).permute(0, 4, 1, 2, 3)
    assert rel(out, g["camera_out"]) < tol
    # without the camera type the embedding is left out (the reference adds it only under `use_camera_motion_condition`)
    _, temb0 = eng.prepare_time_embeddings([int(g["camera_timestep"])], [2, 2], [4, 4], 2)
    assert not torch.allclose(temb0.float(), temb.float())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 6e-2)])
def test_engine_first_frame_condition_concat(g, dtype, tol):
    """8-channel input built by fyc_unet_input mode 1 (latents | first-frame latents on every frame), conv_in / 2 folded into its weights"""
    kcfg = Fn.tiny_unet_config(use_first_frame_condition_concat=True, use_first_frame_mask_condition_concat=False)
    sd = W.make_weights(W.unet_state_shapes(kcfg), 0)
    ecfg = UNet3DConfig(use_first_frame_condition_concat=True, use_first_frame_mask_condition_concat=False, **TINY)
    eng = UNet3DEngine(pack_unet(sd, ecfg, dtype, "cpu"), ops=EmuOps())
    inp = W.seeded_inputs(kcfg, 1, 2, 8, 8, seed=int(g["concat_input_seed"]))
    x = eng.new(2 * 2 * 64, pad_channels(8))
    eng.ops.unet_input(inp["latents"].contiguous(), None, inp["first_image_latents"].reshape(1, 4, 64).contiguous(), x, B=1, F=2, HW=64, c_latent=4,
                       c_pad=pad_channels(8), cfg_dup=2, mode=1)
    eng.prepare_context(inp["text"])
    _, temb = eng.prepare_time_embeddings([int(g["concat_timestep"])], [2, 2], [4, 4], 2)
    out = eng.forward(x, temb, 2, 2, 8, 8).float().reshape(2, 2, 8, 8, 4).permute(0, 4, 1, 2, 3)
    assert rel(out, g["concat_out"]) < tol
    # the sampler takes the same path: one step = guidance over that forward + the DDIM update
    lat = DDIMSampler(eng, DDIMConfig()).sample(inp["latents"], inp["text"], 2, 8.0, inp["first_image_latents"], None, fps=[2], flow=[4])
    with torch.no_grad():
        ref = Fn.denoise(sd, kcfg, Fn.DDIMConfig(), inp["latents"].clone(), inp["text"], 2, 8.0, inp["first_image_latents"], None, torch.tensor([2]), torch.tensor([4]))
    assert rel(lat, ref) < (5e-4 if dtype == torch.float32 else 1e-1)


def test_concat_flag_wins_over_the_default_mask_flag():
    """both concat flags set (the mask flag DEFAULTS to True): the 8-channel concat wins everywhere - UNet3DConfig.conv_in_channels,
    pack_unet (conv_in / 2) and the sampler's input layout - as in the reference's constructor (unet.py:114-126).  Round-5 advisor:
    the sampler used to test the mask flag first and fed the halved 8-channel conv_in the 9-channel layout."""
    kcfg = Fn.tiny_unet_config(use_first_frame_condition_concat=True, use_first_frame_mask_condition_concat=False)
    sd = W.make_weights(W.unet_state_shapes(kcfg), 0)
    inp = W.seeded_inputs(kcfg, 1, 2, 8, 8, seed=5)
    lats = []
    for mask_flag in (False, True):
        ecfg = UNet3DConfig(use_first_frame_condition_concat=True, use_first_frame_mask_condition_concat=mask_flag, **TINY)
        assert ecfg.conv_in_channels == 8
        eng = UNet3DEngine(pack_unet(sd, ecfg, torch.float32, "cpu"), ops=EmuOps())
        lats.append(DDIMSampler(eng, DDIMConfig()).sample(inp["latents"], inp["text"], 2, 8.0, inp["first_image_latents"], None, fps=[2], flow=[4]))
    assert torch.equal(lats[0], lats[1])
    with torch.no_grad():
        ref = Fn.denoise(sd, kcfg, Fn.DDIMConfig(), inp["latents"].clone(), inp["text"], 2, 8.0, inp["first_image_latents"], None, torch.tensor([2]), torch.tensor([4]))
    assert rel(lats[1], ref) < 5e-4


# ---- the HIP kernels behind the two ops that changed ---------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
def test_hip_cfg_ddim_step_with_noise_and_unet_input_mode1(dt):
    from followyourclick_amd import ops
    h = ops.get()
    dev = torch.device("cuda:0")
    h.ensure_init(dev)
    emu = EmuOps()
    gen = torch.Generator().manual_seed(3)
    B, C, F, HW, ld = 2, 4, 3, 48, 64
    pred = torch.randn(2 * B * F * HW, ld, generator=gen).to(dt)
    lat, noise = torch.randn(B, C, F, HW, generator=gen), torch.randn(B, C, F, HW, generator=gen)
    for ptype, clip, reclip, sigma in ((1, False, False, 0.3), (1, True, True, 0.2), (0, False, False, 0.4), (1, True, True, 0.0)):
        coef = torch.tensor([0.8, 0.6, 0.9, 0.3, sigma])
        kw = dict(B=B, F=F, HW=HW, c_latent=C, ld=ld, cfg=True, guidance=7.5, pred_type=ptype, clip_sample=clip,
                  sigma=sigma, clipped_model_output=reclip)
        a, b = lat.clone(), lat.clone().to(dev)
        emu.cfg_ddim_step(pred, a, coef, variance_noise=noise if sigma > 0 else None, **kw)
        h.cfg_ddim_step(pred.to(dev), b, coef.to(dev), variance_noise=noise.to(dev) if sigma > 0 else None, **kw)
        assert rel(b.cpu(), a) < 1e-6, (ptype, clip, reclip, sigma)
    first = torch.randn(B, C, HW, generator=gen)
    cp = 64
    xe, xh = torch.zeros(2 * B * F * HW, cp, dtype=dt), torch.zeros(2 * B * F * HW, cp, dtype=dt, device=dev)
    emu.unet_input(lat, None, first, xe, B=B, F=F, HW=HW, c_latent=C, c_pad=cp, cfg_dup=2, mode=1)
    h.unet_input(lat.to(dev), None, first.to(dev), xh, B=B, F=F, HW=HW, c_latent=C, c_pad=cp, cfg_dup=2, mode=1)
    assert torch.equal(xh.cpu(), xe)


# ---- the drop-in API surface (reference class names and signatures) on the op emulator --------------------------------------------------
MM = dict(num_attention_heads=8, num_transformer_block=1, attention_block_types=["Temporal_Self", "Temporal_Self"],
          temporal_position_encoding=True, temporal_position_encoding_max_len=24, temporal_attention_dim_div=1, zero_initialize=True)
TINY_CTOR = dict(sample_size=8, in_channels=4, out_channels=4, block_out_channels=(64, 128, 256, 256), layers_per_block=2,
                 cross_attention_dim=64, attention_head_dim=8, use_motion_module=True, motion_module_resolutions=(1, 2, 4, 8),
                 unet_use_cross_frame_attention=False, unet_use_temporal_attention=False, use_fps_condition=True,
                 use_first_frame_mask_condition_concat=True, motion_module_type="Vanilla", motion_module_kwargs=MM)


@pytest.fixture()
def dropin_emulated(monkeypatch):
    import sys
    import followyourclick_amd
    from followyourclick_amd import ops as ops_mod
    followyourclick_amd.install_dropin(force=True)
    monkeypatch.setattr(ops_mod, "impl", EmuOps())              # CPU box: the drop-in's host orchestration on the op emulator (tests only)
    monkeypatch.setenv("FYC_COMPUTE_DTYPE", "f32")
    yield
    for name in [k for k in sys.modules if k.split(".")[0] in ("animatediff", "diffusers", "ip_adapter")]:
        del sys.modules[name]
    if followyourclick_amd.DROPIN_DIR in sys.path:
        sys.path.remove(followyourclick_amd.DROPIN_DIR)


def test_dropin_scheduler_step_with_eta(dropin_emulated, g):
    from diffusers import DDIMScheduler
    x, v, noise = g["sched_sample"], g["sched_model_output"], g["sched_noise"]
    base = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False,
                prediction_type="v_prediction", rescale_betas_zero_snr=True)
    for tag, ckw, skw in SCHED_CASES:
        s = DDIMScheduler(**dict(base, **ckw))
        s.set_timesteps(25)
        for t in (961, 481, 41):
            out = s.step(v, t, x, variance_noise=noise if skw.get("eta", 0) > 0 else None, **skw).prev_sample
            assert rel(out, g[f"sched_{tag}_{t}"]) < 2e-6, (tag, t)
    with pytest.raises(ValueError, match="Cannot pass both generator and variance_noise"):
        s.step(v, 961, x, eta=0.5, generator=torch.Generator(), variance_noise=noise)
    a = s.step(v, 961, x, eta=0.5, generator=torch.Generator().manual_seed(1)).prev_sample        # generator path: reproducible
    assert torch.equal(a, s.step(v, 961, x, eta=0.5, generator=torch.Generator().manual_seed(1)).prev_sample)


def test_dropin_unet_camera_and_first_frame_concat(dropin_emulated, g):
    from animatediff.models.unet import UNet3DConditionModel
    fps, flow = torch.tensor([2, 2]), torch.tensor([4, 4])
    ccfg = Fn.tiny_unet_config(use_camera_motion_condition=True)
    unet = UNet3DConditionModel(**dict(TINY_CTOR, use_camera_motion_condition=True), compute_dtype=torch.float32)
    unet.load_state_dict(W.make_weights(W.unet_state_shapes(ccfg), 0), strict=True)
    inp = W.seeded_inputs(ccfg, 1, 2, 8, 8, seed=int(g["camera_input_seed"]))
    x9 = torch.cat([Fn.build_model_input(inp["latents"], inp["first_image_latents"], inp["first_images_mask"])] * 2)
    y = unet(x9, torch.tensor(int(g["camera_timestep"])), inp["text"], use_fps_condition=True, fps_tensor=fps, flow_control=flow,
             use_camera_motion_condition=True, camera_movement_type_tensor=g["camera_type"]).sample
    assert rel(y, g["camera_out"]) < 2e-4
    kcfg = Fn.tiny_unet_config(use_first_frame_condition_concat=True, use_first_frame_mask_condition_concat=False)
    unet = UNet3DConditionModel(**dict(TINY_CTOR, use_first_frame_condition_concat=True, use_first_frame_mask_condition_concat=False), compute_dtype=torch.float32)
    unet.load_state_dict(W.make_weights(W.unet_state_shapes(kcfg), 0), strict=True)
    inp = W.seeded_inputs(kcfg, 1, 2, 8, 8, seed=int(g["concat_input_seed"]))
    y = unet(torch.cat([inp["latents"]] * 2), torch.tensor(int(g["concat_timestep"])), inp["text"], use_fps_condition=True, fps_tensor=fps, flow_control=flow,
             use_first_frame_condition_concat=True, reference_images_latent=torch.cat([inp["first_image_latents"]] * 2)).sample
    assert rel(y, g["concat_out"]) < 2e-4
    with pytest.raises(ValueError, match="use_first_frame_condition_concat=True"):           # the halved conv_in must not be used silently
        unet(torch.zeros(2, 8, 2, 8, 8), torch.tensor(1), inp["text"])


def test_dropin_pipeline_with_eta(dropin_emulated, g):
    """AnimationPipeline.__call__(eta=0.6, generator=...) through the drop-in classes = the real reference's stochastic trajectory"""
    from animatediff.models.unet import UNet3DConditionModel
    from animatediff.pipelines.pipeline_animation import AnimationPipeline
    from diffusers import AutoencoderKL, DDIMScheduler
    from oracle import stubs
    cfg = Fn.tiny_unet_config()
    unet = UNet3DConditionModel(**TINY_CTOR, compute_dtype=torch.float32)
    unet.load_state_dict(W.make_weights(W.unet_state_shapes(cfg), 0), strict=True)
    vae = AutoencoderKL(block_out_channels=(64, 128, 128, 128), layers_per_block=2, latent_channels=4)
    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False,
                          prediction_type="v_prediction", rescale_betas_zero_snr=True)
    pipe = AnimationPipeline(vae=vae, text_encoder=stubs.StubTextEncoder(64), tokenizer=stubs.FakeTokenizer(), unet=unet, scheduler=sched)
    frames, lat, steps = int(g["pipe_eta_frames"]), int(g["pipe_eta_lat"]), int(g["pipe_eta_steps"])
    pipe.decode_latents = lambda latents: np.zeros((1, 3, frames, 8, 8), dtype=np.float32)
    inp = W.seeded_inputs(cfg, 1, frames, lat, lat, seed=int(g["pipe_eta_input_seed"]))
    traj = {}
    pipe("a corgi waving its tail", video_length=frames, height=lat * 8, width=lat * 8, num_inference_steps=steps, guidance_scale=8.0,
         negative_prompt="blurry", latents=inp["latents"].clone(), first_image_latents=inp["first_image_latents"], first_images_mask=inp["first_images_mask"],
         use_first_frame_mask_condition_concat=True, use_fps_condition=True, fps_tensor=torch.tensor([2]), flow_control=torch.tensor([4]),
         eta=float(g["pipe_eta"]), generator=torch.Generator().manual_seed(int(g["pipe_eta_generator_seed"])),
         callback=lambda i, t, l: traj.__setitem__(i, l.clone().float()), callback_steps=1)
    for i in range(steps):
        assert rel(traj[i], g[f"pipe_eta_step{i}"]) < 5e-4, i


def test_dropin_unet_per_sample_timesteps(dropin_emulated):
    """a (B,) timestep tensor with different entries (reference unet.py:488-521): one time-embedding row per batch element"""
    from animatediff.models.unet import UNet3DConditionModel
    cfg = Fn.tiny_unet_config()
    sd = W.make_weights(W.unet_state_shapes(cfg), 0)
    unet = UNet3DConditionModel(**TINY_CTOR, compute_dtype=torch.float32)
    unet.load_state_dict(sd, strict=True)
    inp = W.seeded_inputs(cfg, 1, 2, 8, 8, seed=5)
    x9 = torch.cat([Fn.build_model_input(inp["latents"], inp["first_image_latents"], inp["first_images_mask"])] * 2)
    t, fps, flow = torch.tensor([900, 300]), torch.tensor([2, 6]), torch.tensor([4, 1])
    y = unet(x9, t, inp["text"], use_fps_condition=True, fps_tensor=fps, flow_control=flow).sample
    with torch.no_grad():
        ref = Fn.unet3d_forward(sd, cfg, x9, t, inp["text"], fps, flow)
    assert rel(y, ref) < 2e-4
    with pytest.raises(ValueError, match="timestep has 3 entries"):
        unet(x9, torch.tensor([1, 2, 3]), inp["text"])
