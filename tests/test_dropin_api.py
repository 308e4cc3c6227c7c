"""API surface of the drop-in packages (CPU): import paths, constructor / call signatures compared with the
reference's (recorded from the reference source), state-dict schema, error behaviour."""
import inspect
import os
import sys

import numpy as np
import pytest
import torch

import followyourclick_amd


@pytest.fixture(scope="module")
def dropin():
    followyourclick_amd.install_dropin(force=True)
    yield
    for name in [k for k in sys.modules if k.split(".")[0] in ("animatediff", "diffusers", "ip_adapter")]:
        del sys.modules[name]
    if followyourclick_amd.DROPIN_DIR in sys.path:
        sys.path.remove(followyourclick_amd.DROPIN_DIR)


# parameter names of the reference signatures (animatediff/pipelines/pipeline_animation.py:547-584,
# animatediff/models/unet.py:422-444)
PIPE_CALL = ["prompt", "video_length", "height", "width", "num_inference_steps", "guidance_scale", "negative_prompt",
             "num_videos_per_prompt", "eta", "generator", "latents", "output_type", "return_dict", "callback", "callback_steps",
             "use_first_frame_condition", "use_first_frame_condition_concat", "use_first_frame_mask_condition_concat",
             "use_first_frame_mask_condition_concat_image_partial_mask", "first_image_latents", "use_first_image_as_init_latents",
             "video_scale", "use_ip_cross_attention", "condition_images", "use_uncond_images", "use_camera_motion_condition",
             "camera_movement_type", "use_text_encoder_2", "use_uncond_text_2", "use_fps_condition", "fps_tensor",
             "use_interpolate_noise", "first_images_mask", "flow_control", "kwargs"]
UNET_FWD = ["sample", "timestep", "encoder_hidden_states", "class_labels", "attention_mask", "return_dict",
            "use_first_frame_condition", "use_first_frame_condition_concat", "use_ip_cross_attention", "reference_images_latent",
            "reference_images_clip_feat", "use_camera_motion_condition", "camera_movement_type_tensor", "use_image_concat_training",
            "use_text_encoder_2", "encoder_hidden_states_2", "use_fps_condition", "fps_tensor", "first_images_mask", "flow_control"]

MM = dict(num_attention_heads=8, num_transformer_block=1, attention_block_types=["Temporal_Self", "Temporal_Self"],
          temporal_position_encoding=True, temporal_position_encoding_max_len=24, temporal_attention_dim_div=1, zero_initialize=True)
TINY = dict(sample_size=8, in_channels=4, out_channels=4, block_out_channels=(64, 128, 256, 256), layers_per_block=2,
            cross_attention_dim=64, attention_head_dim=8, use_motion_module=True, motion_module_resolutions=(1, 2, 4, 8),
            unet_use_cross_frame_attention=False, unet_use_temporal_attention=False, use_fps_condition=True,
            use_first_frame_mask_condition_concat=True, motion_module_type="Vanilla", motion_module_kwargs=MM)


def test_signatures_match_reference(dropin):
    from animatediff.models.unet import UNet3DConditionModel
    from animatediff.pipelines.pipeline_animation import AnimationPipeline
    assert [p for p in inspect.signature(AnimationPipeline.__call__).parameters][1:] == PIPE_CALL
    assert [p for p in inspect.signature(UNet3DConditionModel.forward).parameters][1:] == UNET_FWD
    ctor = inspect.signature(AnimationPipeline.__init__).parameters
    assert list(ctor)[1:] == ["vae", "text_encoder", "tokenizer", "unet", "scheduler", "image_encoder", "text_encoder_2", "tokenizer_2", "ip_adapter"]
    from diffusers.utils.import_utils import is_xformers_available
    assert is_xformers_available()


def test_unet_state_dict_has_reference_keys(dropin, golden_dir):
    import json
    from animatediff.models.unet import UNet3DConditionModel
    unet = UNet3DConditionModel(**TINY)
    with open(os.path.join(golden_dir, "schema_unet_tiny.json")) as f:
        ref = {k: tuple(v) for k, v in json.load(f).items()}
    mine = {k: tuple(v.shape) for k, v in unet.state_dict().items()}
    assert mine == ref
    assert unet.in_channels == 4 and unet.config.sample_size == 8 and unet.config.cross_attention_dim == 64
    # zero-initialised tensors of the reference (motion proj_out, fps/motion linear_2)
    assert unet.state_dict()["down_blocks.0.motion_modules.0.temporal_transformer.proj_out.weight"].abs().sum() == 0
    assert unet.state_dict()["fps_embedding.linear_2.weight"].abs().sum() == 0
    # checkpoints with a `module.` prefix / partial key sets load the way scripts/inference.py:170-181 expects
    sd = {"down_blocks.0.motion_modules.0.temporal_transformer.proj_out.weight": torch.ones(64, 64), "bogus.key": torch.zeros(1)}
    missing, unexpected = unet.load_state_dict(sd, strict=False)
    assert unexpected == ["bogus.key"] and len(missing) == len(ref) - 1
    unet.enable_xformers_memory_efficient_attention()
    unet.set_attention_slice("auto")


def test_unsupported_options_fail_loudly(dropin):
    from animatediff.models.unet import UNet3DConditionModel
    with pytest.raises(NotImplementedError):
        UNet3DConditionModel(**dict(TINY, use_inflated_groupnorm=True))
    unet = UNet3DConditionModel(**TINY)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        unet(torch.zeros(2, 9, 4, 8, 8), torch.tensor(1), torch.zeros(2, 77, 64))


def test_scheduler_surface(dropin, golden_dir):
    import numpy as np
    from diffusers import DDIMScheduler
    g = np.load(os.path.join(golden_dir, "ddim.npz"))
    s = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1,
                      clip_sample=False, prediction_type="v_prediction", rescale_betas_zero_snr=True)
    assert s.order == 1 and s.init_noise_sigma == 1.0 and s.config.steps_offset == 1 and s.config.clip_sample is False
    with pytest.raises(ValueError):
        s.step(torch.zeros(1), 1, torch.zeros(1))
    s.set_timesteps(25)
    assert np.array_equal(s.timesteps.numpy(), g["timesteps_25"])
    x, v = torch.from_numpy(g["step_sample"]), torch.from_numpy(g["step_model_output"])
    assert torch.allclose(s.step(v, 961, x, eta=0.0).prev_sample, torch.from_numpy(g["step_out_t961"]), atol=1e-6)
    assert torch.equal(s.scale_model_input(x, 5), x)


def test_pipeline_argument_errors(dropin):
    """the reference's ValueErrors (pipeline_animation.py:432-445, 517-518) before any GPU work"""
    from animatediff.models.unet import UNet3DConditionModel
    from animatediff.pipelines.pipeline_animation import AnimationPipeline
    from diffusers import AutoencoderKL, DDIMScheduler
    from oracle import stubs
    vae = AutoencoderKL(block_out_channels=(64, 128, 128, 128), layers_per_block=2, latent_channels=4)
    pipe = AnimationPipeline(vae=vae, text_encoder=stubs.StubTextEncoder(64), tokenizer=stubs.FakeTokenizer(),
                             unet=UNet3DConditionModel(**TINY), scheduler=DDIMScheduler(steps_offset=0, clip_sample=True))
    assert pipe.scheduler.config.steps_offset == 1 and pipe.scheduler.config.clip_sample is False and pipe.vae_scale_factor == 8
    with pytest.raises(ValueError, match="divisible by 8"):
        pipe("x", video_length=4, height=60, width=64, use_first_frame_mask_condition_concat=True, first_image_latents=torch.zeros(1, 4, 8, 8))
    with pytest.raises(ValueError, match="prompt"):
        pipe(3, video_length=4, height=64, width=64)
    with pytest.raises(ValueError, match="callback_steps"):
        pipe("x", video_length=4, height=64, width=64, callback_steps=0)
    with pytest.raises(ValueError, match="Unexpected latents shape"):
        pipe("x", video_length=4, height=64, width=64, latents=torch.zeros(1, 4, 3, 8, 8), use_first_frame_mask_condition_concat=True,
             first_image_latents=torch.zeros(1, 4, 8, 8))
    with pytest.raises(ValueError, match="built"):        # ... nor the 8-channel first-frame concat (supported since round 5 on a model built for it)
        pipe("x", video_length=4, height=64, width=64, use_first_frame_condition_concat=True, first_image_latents=torch.zeros(1, 4, 8, 8))
    with pytest.raises(ValueError, match="use_camera_motion_condition"):
        pipe("x", video_length=4, height=64, width=64, use_first_frame_mask_condition_concat=True, first_image_latents=torch.zeros(1, 4, 8, 8),
             use_camera_motion_condition=True, camera_movement_type=torch.tensor([1]))
    with pytest.raises(ValueError, match="first_image_latents is required"):
        pipe("x", video_length=4, height=64, width=64, use_first_frame_mask_condition_concat=True)
    with pytest.raises(ValueError, match="built"):        # a 9-channel (concat) UNet cannot run the 4-channel first-frame mode
        pipe("x", video_length=4, height=64, width=64, use_first_frame_condition=True, first_image_latents=torch.zeros(1, 4, 8, 8))


def test_attention_processor_surface(dropin):
    import ip_adapter.attention_processor as ap
    for n in ("AttnProcessor", "IPAttnProcessor", "AttnProcessor2_0", "IPAttnProcessor2_0", "CNAttnProcessor", "CNAttnProcessor2_0"):
        assert hasattr(ap, n)
    p = ap.IPAttnProcessor(hidden_size=320, cross_attention_dim=768, scale=0.5, num_tokens=16)
    assert p.to_k_ip.weight.shape == (320, 768) and p.to_v_ip.bias is None and p.num_tokens == 16 and p.scale == 0.5
    assert list(inspect.signature(p.__call__).parameters) == ["attn", "hidden_states", "encoder_hidden_states", "attention_mask", "temb"]
    assert isinstance(p, torch.nn.Module) and not isinstance(ap.CNAttnProcessor(), torch.nn.Module)


# ---- 2-D first-image path: diffusers.UNet2DConditionModel / StableDiffusionPipeline / DDIMScheduler.from_pretrained ----
TINY_2D = dict(sample_size=8, block_out_channels=(64, 128, 256, 256), cross_attention_dim=64)


def test_unet2d_state_dict_and_loading(dropin, golden_dir, tmp_path):
    import json
    from diffusers import UNet2DConditionModel
    from diffusers.models import UNet2DConditionModel as M2
    assert M2 is UNet2DConditionModel
    unet = UNet2DConditionModel(**TINY_2D)
    with open(os.path.join(golden_dir, "schema_unet2d_tiny.json")) as f:
        ref = {k: tuple(v) for k, v in json.load(f).items()}
    assert {k: tuple(v.shape) for k, v in unet.state_dict().items()} == ref        # the REAL reference 2-D UNet's state dict
    assert unet.config.down_block_types[0] == "CrossAttnDownBlock2D" and unet.in_channels == 4
    assert [p for p in inspect.signature(UNet2DConditionModel.forward).parameters][1:] == \
        ["sample", "timestep", "encoder_hidden_states", "class_labels", "attention_mask", "return_dict"]
    # from_pretrained on a diffusers-style directory (config.json + diffusion_pytorch_model.bin)
    d = tmp_path / "unet"
    d.mkdir()
    (d / "config.json").write_text(json.dumps(dict(TINY_2D, _class_name="UNet2DConditionModel", _diffusers_version="0.11.1")))
    torch.save(unet.state_dict(), d / "diffusion_pytorch_model.bin")
    again = UNet2DConditionModel.from_pretrained(str(tmp_path), subfolder="unet")
    assert all(torch.equal(a, b) for a, b in zip(unet.state_dict().values(), again.state_dict().values()))
    with pytest.raises(EnvironmentError):
        UNet2DConditionModel.from_pretrained(str(tmp_path), subfolder="nope")
    with pytest.raises(NotImplementedError):
        UNet2DConditionModel(**TINY_2D, down_block_types=("AttnDownBlock2D",) * 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        unet(torch.zeros(1, 4, 8, 8), 1, torch.zeros(1, 77, 64))


def test_sd_pipeline_surface(dropin, tmp_path):
    import json
    from diffusers import AutoencoderKL, DDIMScheduler, StableDiffusionPipeline, UNet2DConditionModel
    from diffusers.pipelines import StableDiffusionPipeline as P2
    assert P2 is StableDiffusionPipeline
    assert [p for p in inspect.signature(StableDiffusionPipeline.__call__).parameters][1:] == \
        ["prompt", "height", "width", "num_inference_steps", "guidance_scale", "negative_prompt", "num_images_per_prompt", "eta",
         "generator", "latents", "output_type", "return_dict", "callback", "callback_steps"]
    # a checkpoint's scheduler_config.json written by another scheduler class (SD-1.5 ships PNDM's)
    d = tmp_path / "scheduler"
    d.mkdir()
    (d / "scheduler_config.json").write_text(json.dumps(dict(
        _class_name="PNDMScheduler", _diffusers_version="0.6.0", beta_end=0.012, beta_schedule="scaled_linear", beta_start=0.00085,
        num_train_timesteps=1000, set_alpha_to_one=False, skip_prk_steps=True, steps_offset=1, trained_betas=None, clip_sample=False)))
    sch = DDIMScheduler.from_pretrained(str(tmp_path), subfolder="scheduler")
    assert sch.config.beta_schedule == "scaled_linear" and sch.config.set_alpha_to_one is False and sch.config.prediction_type == "epsilon"
    assert float(sch.final_alpha_cumprod) == float(sch.alphas_cumprod[0])
    from oracle import stubs
    unet, vae = UNet2DConditionModel(**TINY_2D), AutoencoderKL(block_out_channels=(64, 128, 128, 128))
    with pytest.raises(ValueError, match="needs"):
        StableDiffusionPipeline.from_pretrained("unused", unet=unet)
    pipe = StableDiffusionPipeline.from_pretrained("unused", unet=unet, vae=vae, tokenizer=stubs.FakeTokenizer(),
                                                   text_encoder=stubs.StubTextEncoder(64), scheduler=sch, safety_checker=None)
    assert pipe.vae_scale_factor == 8
    with pytest.raises(ValueError, match="divisible by 8"):
        pipe("x", height=60, width=64)
    with pytest.raises(ValueError, match="`prompt` has to be"):
        pipe(3, height=64, width=64)
    with pytest.raises(ValueError, match="Unexpected latents shape"):
        pipe("x", height=64, width=64, latents=torch.zeros(1, 4, 4, 4))
    with pytest.raises(NotImplementedError):
        StableDiffusionPipeline(vae, stubs.StubTextEncoder(64), stubs.FakeTokenizer(), unet, sch, safety_checker=object())


# ---- conditioning front-end: ip_adapter.{Resampler, ImageProjModel, MyIPAdapter[Plus]}, CLIP fronts -------------
def test_ip_adapter_modules_surface(dropin, golden_dir, tmp_path):
    import ip_adapter
    from ip_adapter.my_ip_adapter import ImageProjModel, MyIPAdapter, MyIPAdapterPlus
    from ip_adapter.resampler import Resampler
    from oracle import encoders as E
    assert ip_adapter.Resampler is Resampler and ip_adapter.MyIPAdapterPlus is MyIPAdapterPlus
    rc = E.TINY_RESAMPLER
    res = Resampler(dim=rc.dim, depth=rc.depth, dim_head=rc.dim_head, heads=rc.heads, num_queries=rc.num_queries,
                    embedding_dim=rc.embedding_dim, output_dim=rc.output_dim, ff_mult=rc.ff_mult)
    # these shape tables were loaded with strict=True into the REAL reference classes by oracle/make_golden_encoders.py
    assert {k: tuple(v.shape) for k, v in res.state_dict().items()} == dict(E.resampler_shapes(rc))
    proj = ImageProjModel(cross_attention_dim=64, clip_embeddings_dim=64, clip_extra_context_tokens=4)
    assert {k: tuple(v.shape) for k, v in proj.state_dict().items()} == dict(E.image_proj_shapes(64, 64, 4))
    with pytest.raises(NotImplementedError):
        Resampler(apply_pos_emb=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        proj(torch.zeros(1, 64))
    assert list(inspect.signature(MyIPAdapter.__init__).parameters)[1:] == ["unet", "image_encoder_path", "ip_ckpt", "device", "num_tokens"]
    for name in ("init_proj", "get_ip_adapter_state_dict", "load_ip_adapter", "get_image_clip_feat", "get_image_embeds"):
        assert callable(getattr(MyIPAdapterPlus, name))


def test_load_ip_adapter_state_surgery(dropin, tmp_path):
    """checkpoint `ip_adapter` tensors are matched in order with the UNet's *_ip* parameters; image_proj goes to the projection"""
    from animatediff.models.unet import UNet3DConditionModel
    from followyourclick_amd.encoders import ClipVisionHip
    from ip_adapter.my_ip_adapter import MyIPAdapter
    from oracle import encoders as E
    unet = UNet3DConditionModel(**dict(TINY, use_ip_cross_attention=True, num_tokens=4))
    vis = ClipVisionHip(E.make_encoder_weights(E.clip_vision_shapes(E.TINY_VISION), 53), vars(E.TINY_VISION))
    ip_keys = [k for k in unet.state_dict() if "_ip" in k]
    assert len(ip_keys) == 32                                    # 16 cross-attention layers x (to_k_ip, to_v_ip)
    ck = {"image_proj": E.make_encoder_weights(E.image_proj_shapes(64, 64, 4), 55),
          "ip_adapter": {f"{i}.to_{'kv'[i % 2]}_ip.weight": torch.full_like(unet.state_dict()[k], float(i)) for i, k in enumerate(ip_keys)}}
    path = str(tmp_path / "ip.bin")
    torch.save(ck, path)
    ad = MyIPAdapter(unet, vis, path, "cpu", num_tokens=4)
    missing, unexpected = ad.load_ip_adapter()
    assert not unexpected
    sd = unet.state_dict()
    assert all(float(sd[k].flatten()[0]) == float(i) for i, k in enumerate(ip_keys))
    assert torch.equal(ad.image_proj_model.state_dict()["proj.weight"], ck["image_proj"]["proj.weight"])
    # use_unet_image_proj_model=True: the projection weights travel inside the UNet state dict (scripts/inference.py:167-168)
    unet.image_proj_model = ad.init_proj()
    ad.load_ip_adapter(unet, use_unet_image_proj_model=True)
    assert torch.equal(unet.state_dict()["image_proj_model.norm.bias"], ck["image_proj"]["norm.bias"])


def test_load_ip_adapter_plus_state_surgery(dropin, tmp_path):
    """MyIPAdapterPlus: the Resampler weights come from the checkpoint's `image_proj` entry, either into the adapter's own projection
    or into a fresh `unet.image_proj_model` (reference my_ip_adapter.py:252-262)"""
    from animatediff.models.unet import UNet3DConditionModel
    from followyourclick_amd.encoders import ClipVisionHip
    from ip_adapter.my_ip_adapter import MyIPAdapterPlus
    from ip_adapter.resampler import Resampler
    from oracle import encoders as E
    unet = UNet3DConditionModel(**dict(TINY, use_ip_cross_attention=True, num_tokens=4))
    vis = ClipVisionHip(E.make_encoder_weights(E.clip_vision_shapes(E.TINY_VISION), 53), vars(E.TINY_VISION))
    ad = MyIPAdapterPlus(unet, vis, None, "cpu", num_tokens=4)
    assert isinstance(ad.image_proj_model, Resampler)
    rcfg = E.ResamplerConfig(dim=64, depth=4, dim_head=64, heads=12, num_queries=4, embedding_dim=E.TINY_VISION.hidden_size, output_dim=64)
    proj_sd = E.make_encoder_weights(E.resampler_shapes(rcfg), 91)
    assert {k: tuple(v.shape) for k, v in ad.image_proj_model.state_dict().items()} == dict(E.resampler_shapes(rcfg))
    ip_keys = [k for k in unet.state_dict() if "_ip" in k]
    ck = {"image_proj": proj_sd, "ip_adapter": {f"{i}.w": torch.full_like(unet.state_dict()[k], float(i) + 0.5) for i, k in enumerate(ip_keys)}}
    path = str(tmp_path / "ip_plus.bin")
    torch.save(ck, path)
    ad.ip_ckpt = path
    ad.load_ip_adapter()
    assert torch.equal(ad.image_proj_model.state_dict()["layers.3.1.3.weight"], proj_sd["layers.3.1.3.weight"])
    assert all(float(unet.state_dict()[k].flatten()[0]) == i + 0.5 for i, k in enumerate(ip_keys))
    ad.load_ip_adapter(unet, use_unet_image_proj_model=True)
    assert isinstance(unet.image_proj_model, Resampler)
    assert torch.equal(unet.image_proj_model.state_dict()["latents"], proj_sd["latents"])
    assert "image_proj_model.latents" in unet.state_dict()        # travels with the UNet's state dict, as in the reference


def test_prepare_latents_vs_reference_golden(dropin, golden_dir):
    """AnimationPipeline.prepare_latents against the real reference's outputs (oracle/make_golden_full.py p2; reference
    pipeline_animation.py:448-536): seeded noise, use_interpolate_noise repeat, the init_latents blend in both branches,
    use_residual_noise, and the two error paths."""
    from animatediff.models.unet import UNet3DConditionModel
    from animatediff.pipelines.pipeline_animation import AnimationPipeline
    from diffusers import AutoencoderKL, DDIMScheduler
    from oracle import stubs
    g = dict(np.load(os.path.join(golden_dir, "prepare_latents.npz")))
    vae = AutoencoderKL(block_out_channels=(64, 128, 128, 128), layers_per_block=2, latent_channels=4)
    pipe = AnimationPipeline(vae=vae, text_encoder=stubs.StubTextEncoder(64), tokenizer=stubs.FakeTokenizer(),
                             unet=UNet3DConditionModel(**TINY), scheduler=DDIMScheduler())
    first, mask, given = (torch.from_numpy(g[k]) for k in ("first_image_latents", "first_images_mask", "given"))
    cases = dict(gen_interp=dict(), gen_plain=dict(use_interpolate_noise=False),
                 gen_init=dict(use_interpolate_noise=False, init_latents=first, first_images_mask=mask),
                 gen_init_interp=dict(init_latents=first, first_images_mask=mask),
                 gen_init_nomask=dict(init_latents=first),
                 gen_residual=dict(use_interpolate_noise=False, use_residual_noise=True, base_lambda=0.9),
                 given_init=dict(latents=given.clone(), init_latents=first, k=30), given_plain=dict(latents=given.clone()),
                 given_badshape=dict(latents=given[:, :, :5].clone()))
    errors = dict(TypeError=TypeError, ValueError=ValueError)
    for name, kw in cases.items():
        gen = torch.Generator().manual_seed(77)
        if name + "_error" in g:
            with pytest.raises(errors[str(g[name + "_error"])]):
                pipe.prepare_latents(1, 4, 6, 64, 64, torch.float32, torch.device("cpu"), gen, **kw)
            continue
        out = pipe.prepare_latents(1, 4, 6, 64, 64, torch.float32, torch.device("cpu"), gen, **kw)
        assert torch.equal(out, torch.from_numpy(g[name])), name


def test_install_dropin_evicts_a_namespace_package(tmp_path):
    """the reference's `animatediff` has no __init__.py (a namespace package: __file__ is None): install_dropin must report it
    as foreign, refuse without force and evict it with force=True"""
    import importlib
    ref = tmp_path / "refroot"
    (ref / "animatediff" / "models").mkdir(parents=True)
    (ref / "animatediff" / "models" / "__init__.py").write_text("")
    for k in [k for k in sys.modules if k.split(".")[0] in ("animatediff", "diffusers", "ip_adapter")]:
        del sys.modules[k]
    if followyourclick_amd.DROPIN_DIR in sys.path:
        sys.path.remove(followyourclick_amd.DROPIN_DIR)
    sys.path.insert(0, str(ref))
    try:
        mod = importlib.import_module("animatediff")
        assert getattr(mod, "__file__", None) is None
        with pytest.raises(RuntimeError, match="already imported"):
            followyourclick_amd.install_dropin()
        followyourclick_amd.install_dropin(force=True)
        mod = importlib.import_module("animatediff")
        assert any(str(p).startswith(followyourclick_amd.DROPIN_DIR) for p in list(getattr(mod, "__path__", [])) + [mod.__file__ or ""])
    finally:
        sys.path.remove(str(ref))
        for k in [k for k in sys.modules if k.split(".")[0] in ("animatediff", "diffusers", "ip_adapter")]:
            del sys.modules[k]
        if followyourclick_amd.DROPIN_DIR in sys.path:
            sys.path.remove(followyourclick_amd.DROPIN_DIR)
