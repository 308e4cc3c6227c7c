"""The drop-in API surface on a real MI355X: the same calls scripts/inference.py makes, compared with the
REAL reference's outputs (tests/golden/pipeline_tiny.npz was produced by the reference AnimationPipeline
with these very test doubles for tokenizer / text encoder)."""
import os
import sys

import numpy as np
import pytest
import torch

import followyourclick_amd
from oracle import functional as Fn
from oracle import stubs
from oracle import weights as W

pytestmark = pytest.mark.gpu

MM = dict(num_attention_heads=8, num_transformer_block=1, attention_block_types=["Temporal_Self", "Temporal_Self"],
          temporal_position_encoding=True, temporal_position_encoding_max_len=24, temporal_attention_dim_div=1, zero_initialize=True)
TINY = dict(sample_size=8, in_channels=4, out_channels=4, block_out_channels=(64, 128, 256, 256), layers_per_block=2,
            cross_attention_dim=64, attention_head_dim=8, use_motion_module=True, motion_module_resolutions=(1, 2, 4, 8),
            unet_use_cross_frame_attention=False, unet_use_temporal_attention=False, use_fps_condition=True,
            use_first_frame_mask_condition_concat=True, motion_module_type="Vanilla", motion_module_kwargs=MM)


@pytest.fixture(scope="module")
def dropin():
    followyourclick_amd.install_dropin(force=True)
    yield
    for name in [k for k in sys.modules if k.split(".")[0] in ("animatediff", "diffusers", "ip_adapter")]:
        del sys.modules[name]


def _load(golden_dir, name):
    return {k: torch.from_numpy(v) if v.shape else v for k, v in np.load(os.path.join(golden_dir, name)).items()}


@pytest.mark.parametrize("dtype,tol_lat,tol_vid", [(torch.float32, 1e-3, 1e-3), (torch.bfloat16, 1.05e-1, 5.5e-2)])      # bf16: 2 x the measured 5.2e-2 / 2.7e-2
def test_animation_pipeline_call(dropin, golden_dir, dtype, tol_lat, tol_vid):
    from animatediff.models.unet import UNet3DConditionModel
    from animatediff.pipelines.pipeline_animation import AnimationPipeline
    from diffusers import AutoencoderKL, DDIMScheduler
    g = _load(golden_dir, "pipeline_tiny.npz")
    unet = UNet3DConditionModel(**TINY, compute_dtype=dtype)
    m, u = unet.load_state_dict(W.make_weights(W.unet_state_shapes(Fn.tiny_unet_config()), int(g["unet_weight_seed"])), strict=False)
    assert not m and not u
    vae = AutoencoderKL(block_out_channels=(64, 128, 128, 128), layers_per_block=2, latent_channels=4, compute_dtype=dtype)
    vae.load_state_dict(W.make_weights(W.vae_decoder_state_shapes(Fn.VAEConfig(block_out_channels=(64, 128, 128, 128))), int(g["vae_weight_seed"])), strict=False)
    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1,
                          clip_sample=False, prediction_type="v_prediction", rescale_betas_zero_snr=True)
    pipe = AnimationPipeline(vae=vae, text_encoder=stubs.StubTextEncoder(64), tokenizer=stubs.FakeTokenizer(), unet=unet,
                             scheduler=sched).to("cuda")
    traj = []
    out = pipe("a corgi waving its tail", video_length=4, height=64, width=64, num_inference_steps=5, guidance_scale=8.0,
               negative_prompt="blurry", latents=g["latents"].clone(), first_image_latents=g["first_image_latents"].cuda(),
               first_images_mask=g["first_images_mask"].cuda(), use_first_frame_mask_condition_concat=True,
               use_fps_condition=True, fps_tensor=torch.tensor([2]), flow_control=torch.tensor([4]),
               callback=lambda i, t, l: traj.append(l.clone().cpu()), callback_steps=1,
               use_first_frame_mask_condition_concat_zero_padding=True)        # unknown kwarg swallowed like the reference
    traj = torch.stack(traj)
    err = (traj - g["trajectory"]).flatten(1).norm(dim=1) / g["trajectory"].flatten(1).norm(dim=1)
    assert err.max().item() < tol_lat, err
    assert isinstance(out.videos, torch.Tensor) and out.videos.dtype == torch.float32 and out.videos.device.type == "cpu"
    assert out.videos.shape == g["videos"].shape
    e = ((out.videos - g["videos"]).norm() / g["videos"].norm()).item()
    assert e < tol_vid, e


def test_unet_module_forward_and_reload(dropin, golden_dir):
    from animatediff.models.unet import UNet3DConditionModel
    g = _load(golden_dir, "unet_tiny_fwd.npz")
    unet = UNet3DConditionModel(**TINY, compute_dtype=torch.float32).to("cuda")
    unet.load_state_dict(W.make_weights(W.unet_state_shapes(Fn.tiny_unet_config()), int(g["weight_seed"])))
    out = unet(g["sample"].cuda(), torch.tensor(int(g["timestep"])), g["text"].cuda(), use_fps_condition=True,
               fps_tensor=g["fps"].cuda(), flow_control=g["flow"].cuda()).sample.cpu()
    assert ((out - g["out"]).norm() / g["out"].norm()).item() < 1e-3
    # changing a weight must invalidate the packed engine copy
    with torch.no_grad():
        unet.conv_out.bias.add_(1.0)
    out2 = unet(g["sample"].cuda(), torch.tensor(int(g["timestep"])), g["text"].cuda(), use_fps_condition=True,
                fps_tensor=g["fps"].cuda(), flow_control=g["flow"].cuda()).sample.cpu()
    assert torch.allclose(out2, out + 1.0, atol=2e-3)


class _HostAttn(torch.nn.Module):
    """the host attention module a diffusers>=0.17 UNet hands to a processor"""

    def __init__(self, C, ctx, heads):
        super().__init__()
        self.heads, self.scale = heads, (C // heads) ** -0.5
        self.to_q, self.to_k, self.to_v = (torch.nn.Linear(C, C, bias=False), torch.nn.Linear(ctx, C, bias=False), torch.nn.Linear(ctx, C, bias=False))
        self.to_out = torch.nn.ModuleList([torch.nn.Linear(C, C), torch.nn.Dropout(0.0)])
        self.spatial_norm = self.group_norm = None
        self.norm_cross, self.residual_connection, self.rescale_output_factor = False, True, 1.0


def _ref_core(q, k, v, heads, scale):
    B, N, C = q.shape
    d = C // heads
    sp = lambda t: t.reshape(B, t.shape[1], heads, d).permute(0, 2, 1, 3)
    o = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * scale, -1) @ sp(v)
    return o.permute(0, 2, 1, 3).reshape(B, N, C)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 2e-2)])
def test_ip_attention_processors(dropin, dtype, tol):
    import ip_adapter.attention_processor as ap
    torch.manual_seed(0)
    C, ctx, heads, ntok = 320, 768, 8, 4
    attn = _HostAttn(C, ctx, heads).cuda().to(dtype)
    proc = ap.IPAttnProcessor2_0(hidden_size=C, cross_attention_dim=ctx, scale=0.6, num_tokens=ntok).cuda().to(dtype)
    x = torch.randn(2, C, 8, 8, device="cuda", dtype=dtype)          # 4-D input form
    enc = torch.randn(2, 77 + ntok, ctx, device="cuda", dtype=dtype)
    out = proc(attn, x, enc)
    with torch.no_grad():                                            # the reference's math in torch fp32
        import copy
        a32, p32 = copy.deepcopy(attn).float(), copy.deepcopy(proc).float()
        h = x.float().view(2, C, 64).transpose(1, 2)
        q = a32.to_q(h)
        t, ip = enc.float()[:, :77], enc.float()[:, 77:]
        o = _ref_core(q, a32.to_k(t), a32.to_v(t), heads, a32.scale) + 0.6 * _ref_core(q, p32.to_k_ip(ip), p32.to_v_ip(ip), heads, a32.scale)
        ref = a32.to_out[0](o).transpose(-1, -2).reshape(2, C, 8, 8) + x.float()
    assert out.shape == x.shape
    assert ((out.float() - ref).norm() / ref.norm()).item() < tol
    attn_self = _HostAttn(C, C, heads).cuda().to(dtype)
    plain = ap.AttnProcessor()(attn_self, x.view(2, C, 64).transpose(1, 2).contiguous(), None)       # self-attention, 3-D form
    assert plain.shape == (2, 64, C) and torch.isfinite(plain.float()).all()
    cn = ap.CNAttnProcessor(num_tokens=ntok)(attn, x, enc)
    assert cn.shape == x.shape


def test_autoencoder_encode_then_decode_surface(dropin, golden_dir):
    from diffusers import AutoencoderKL
    g = _load(golden_dir, "vae_enc_tiny.npz")
    vae = AutoencoderKL(block_out_channels=(64, 128, 128, 128), layers_per_block=2, latent_channels=4, compute_dtype=torch.float32).to("cuda")
    sd = W.make_weights(W.vae_encoder_state_shapes(Fn.VAEConfig(block_out_channels=(64, 128, 128, 128))), int(g["weight_seed"]))
    missing, unexpected = vae.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith(("decoder.", "post_quant_conv.")) for k in missing)
    dist = vae.encode(g["x"].cuda()).latent_dist
    assert (dist.parameters.cpu() - g["moments"]).abs().max().item() < 1e-4
    z = dist.sample(torch.Generator(device="cuda").manual_seed(0)) * 0.18215          # scripts/inference.py:356-358
    assert z.shape == (2, 4, 8, 6) and torch.isfinite(z).all()
    img = vae.decode(z / 0.18215).sample
    assert img.shape == (2, 3, 64, 48)


# ---- 2-D first-image path against the REAL reference's UNet2DConditionModel / StableDiffusionPipeline ----------
TINY_2D = dict(sample_size=8, block_out_channels=(64, 128, 256, 256), cross_attention_dim=64)
CFG_2D = dict(use_motion_module=False, use_fps_condition=False, use_first_frame_mask_condition_concat=False)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 6e-2)])
def test_unet2d_forward(dropin, golden_dir, dtype, tol):
    from diffusers import UNet2DConditionModel
    g = _load(golden_dir, "sd2d_unet_fwd.npz")
    unet = UNet2DConditionModel(**TINY_2D, compute_dtype=dtype).to("cuda")
    unet.load_state_dict(W.make_weights(W.unet_state_shapes(Fn.tiny_unet_config(**CFG_2D)), int(g["weight_seed"])))
    for s, t, o in (("sample", "timestep", "out"), ("sample_odd", "timestep_odd", "out_odd")):
        out = unet(g[s].cuda(), torch.tensor(int(g[t])), g["text"].cuda()).sample.cpu()
        assert out.shape == g[o].shape
        assert ((out - g[o]).norm() / g[o].norm()).item() < tol


@pytest.mark.parametrize("dtype,tol_lat,tol_img", [(torch.float32, 1e-3, 2e-3), (torch.bfloat16, 1e-1, 6e-2)])
def test_stable_diffusion_pipeline_call(dropin, golden_dir, dtype, tol_lat, tol_img):
    from diffusers import AutoencoderKL, DDIMScheduler, StableDiffusionPipeline, UNet2DConditionModel
    g = _load(golden_dir, "sd2d_pipeline.npz")
    unet = UNet2DConditionModel(**TINY_2D, compute_dtype=dtype)
    unet.load_state_dict(W.make_weights(W.unet_state_shapes(Fn.tiny_unet_config(**CFG_2D)), int(g["unet_weight_seed"])))
    vae = AutoencoderKL(block_out_channels=(64, 128, 128, 128), layers_per_block=2, latent_channels=4, compute_dtype=dtype)
    vae.load_state_dict(W.make_weights(W.vae_decoder_state_shapes(Fn.VAEConfig(block_out_channels=(64, 128, 128, 128))), int(g["vae_weight_seed"])), strict=False)
    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                          set_alpha_to_one=False, steps_offset=1, prediction_type="epsilon")
    pipe = StableDiffusionPipeline.from_pretrained("unused", vae=vae, text_encoder=stubs.StubTextEncoder(64), tokenizer=stubs.FakeTokenizer(),
                                                   unet=unet, scheduler=sched, safety_checker=None).to("cuda")
    pipe.enable_vae_slicing()
    traj = []
    out = pipe("a corgi on the beach", height=64, width=64, num_inference_steps=4, guidance_scale=8.0, negative_prompt="blurry",
               latents=g["latents"].clone(), output_type="np", callback=lambda i, t, l: traj.append(l.clone().cpu()), callback_steps=1)
    traj = torch.stack(traj)
    err = (traj - g["trajectory"]).flatten(1).norm(dim=1) / g["trajectory"].flatten(1).norm(dim=1)
    assert err.max().item() < tol_lat, err
    assert out.images.shape == g["images"].shape and out.nsfw_content_detected is None
    ref = g["images"].numpy()                                   # images in [0,1]: relative L2 (max-abs is dominated by bf16 rounding)
    e = float(np.linalg.norm(out.images - ref) / np.linalg.norm(ref))
    assert e < tol_img, e
    pil = pipe("a corgi on the beach", height=64, width=64, num_inference_steps=2, latents=g["latents"].clone()).images
    assert len(pil) == 1 and pil[0].size == (64, 64)


# ---- conditioning front-end ---------------------------------------------------------------------------------------
def _vision_model(dtype):
    from followyourclick_amd.encoders import ClipVisionHip
    from oracle import encoders as E
    return ClipVisionHip(E.make_encoder_weights(E.clip_vision_shapes(E.TINY_VISION), 53), vars(E.TINY_VISION), compute_dtype=dtype)


class _Unet64:
    config = type("C", (), {"cross_attention_dim": 64})()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 5e-5), (torch.bfloat16, 3e-2)])
def test_my_ip_adapter_front_end(dropin, golden_dir, dtype, tol):
    """MyIPAdapter.get_image_clip_feat / get_image_embeds == transformers CLIP vision + the reference's ImageProjModel"""
    from ip_adapter.my_ip_adapter import MyIPAdapter, MyIPAdapterPlus
    from oracle import encoders as E
    gv, gi = _load(golden_dir, "enc_clip_vision.npz"), _load(golden_dir, "enc_ip_adapter.npz")
    ad = MyIPAdapter(_Unet64(), _vision_model(dtype), None, "cuda", num_tokens=4)
    ad.image_proj_model.compute_dtype = dtype
    ad.image_proj_model.load_state_dict(E.make_encoder_weights(E.image_proj_shapes(64, 64, 4), int(gi["proj_seed"])))
    cond, uncond = ad.get_image_clip_feat(gv["pixel_values"])
    assert cond.shape == gv["image_embeds"].shape and float(uncond.abs().sum()) == 0.0
    rel = lambda a, b: ((a.float().cpu() - b).norm() / b.norm()).item()
    assert rel(cond, gv["image_embeds"]) < tol
    tok, utok = ad.get_image_embeds(clip_image_embeds=gi["image_embeds"])
    assert rel(tok, gi["proj_tokens"]) < tol and rel(utok, gi["proj_tokens_uncond"]) < tol
    # Plus: penultimate hidden states of the image and of an all-zero image, then the Resampler (depth 4, 12 heads x 64)
    plus = MyIPAdapterPlus(_Unet64(), _vision_model(dtype), None, "cuda", num_tokens=4)
    c, u = plus.get_image_clip_feat(gv["pixel_values"])
    assert rel(c, gv["penultimate"]) < tol
    sd_v = E.make_encoder_weights(E.clip_vision_shapes(E.TINY_VISION), 53)
    u_ref = E.clip_vision_forward(sd_v, E.TINY_VISION, torch.zeros_like(gv["pixel_values"]))[0][-2]
    assert rel(u, u_ref) < tol
    rcfg = E.ResamplerConfig(dim=64, depth=4, dim_head=64, heads=12, num_queries=4, embedding_dim=128, output_dim=64, ff_mult=4)
    sd_r = E.make_encoder_weights(E.resampler_shapes(rcfg), 91)
    plus.image_proj_model.compute_dtype = dtype
    plus.image_proj_model.load_state_dict(sd_r)
    toks = plus.image_proj_model(gv["penultimate"].cuda())
    assert rel(toks, E.resampler_forward(sd_r, rcfg, gv["penultimate"])) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2)])
def test_clip_text_front_in_pipeline(dropin, golden_dir, dtype, tol):
    """pipeline._encode_prompt with the HIP text encoder == transformers.CLIPTextModel on the same token ids"""
    from animatediff.pipelines.pipeline_animation import AnimationPipeline
    from diffusers import AutoencoderKL, DDIMScheduler
    from followyourclick_amd.encoders import ClipTextHip
    from oracle import encoders as E
    g = _load(golden_dir, "enc_clip_text.npz")
    sd = E.make_encoder_weights(E.clip_text_shapes(E.TINY_TEXT), int(g["weight_seed"]))
    enc = ClipTextHip({"text_model." + k: v for k, v in sd.items()}, dict(vars(E.TINY_TEXT), eos_token_id=2), compute_dtype=dtype)
    out = enc.cuda()(g["input_ids"].cuda(), attention_mask=None)
    assert ((out[0].cpu() - g["last_hidden_state"]).norm() / g["last_hidden_state"].norm()).item() < tol
    assert out.pooler_output.shape == (2, 64)
    tok = stubs.FakeTokenizer()
    pipe = AnimationPipeline(vae=AutoencoderKL(block_out_channels=(64, 128, 128, 128)), text_encoder=enc, tokenizer=tok, unet=None,
                             scheduler=DDIMScheduler()).to("cuda")
    emb = pipe._encode_prompt(["a corgi waving its tail"], "cuda", 1, True, ["blurry"])
    ids = torch.cat([tok(["blurry"], max_length=77).input_ids, tok(["a corgi waving its tail"], max_length=77).input_ids])
    ref = E.clip_text_forward(sd, E.TINY_TEXT, ids)
    assert emb.shape == ref.shape and ((emb.cpu() - ref).norm() / ref.norm()).item() < tol
    with pytest.raises(NotImplementedError):
        enc(g["input_ids"].cuda(), attention_mask=torch.ones(2, 77))


def test_pipeline_with_ip_adapter_end_to_end(dropin, golden_dir):
    """cfg5-shaped run through the public API only: condition image -> CLIP vision tower -> ImageProjModel -> IP cross-attention
    UNet -> DDIM loop, all on the engine, against the oracle fed with the same weights (f32 parity mode)."""
    from animatediff.models.unet import UNet3DConditionModel
    from animatediff.pipelines.pipeline_animation import AnimationPipeline
    from diffusers import AutoencoderKL, DDIMScheduler
    from ip_adapter.my_ip_adapter import MyIPAdapter
    from oracle import encoders as E
    gv, g = _load(golden_dir, "enc_clip_vision.npz"), _load(golden_dir, "pipeline_tiny.npz")
    f32 = torch.float32
    ocfg = Fn.tiny_unet_config(use_ip_cross_attention=True, ip_scale=0.7, ip_num_tokens=4)
    sd = W.make_weights(W.unet_state_shapes(ocfg), 5)
    unet = UNet3DConditionModel(**dict(TINY, use_ip_cross_attention=True, num_tokens=4, scale=0.7), compute_dtype=f32)
    assert not any(unet.load_state_dict(sd, strict=False))
    ad = MyIPAdapter(unet, _vision_model(f32), None, "cuda", num_tokens=4)
    sd_p = E.make_encoder_weights(E.image_proj_shapes(64, 64, 4), 55)
    ad.image_proj_model.compute_dtype = f32
    ad.image_proj_model.load_state_dict(sd_p)
    unet.image_proj_model = ad.image_proj_model                          # scripts/inference.py:167
    vae = AutoencoderKL(block_out_channels=(64, 128, 128, 128), layers_per_block=2, latent_channels=4, compute_dtype=f32)
    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1,
                          clip_sample=False, prediction_type="v_prediction", rescale_betas_zero_snr=True)
    pipe = AnimationPipeline(vae=vae, text_encoder=stubs.StubTextEncoder(64), tokenizer=stubs.FakeTokenizer(), unet=unet,
                             scheduler=sched, ip_adapter=ad).to("cuda")
    image = gv["pixel_values"][:1]
    traj = []
    pipe("a corgi waving its tail", video_length=4, height=64, width=64, num_inference_steps=3, guidance_scale=8.0,
         negative_prompt="blurry", latents=g["latents"].clone(), first_image_latents=g["first_image_latents"].cuda(),
         first_images_mask=g["first_images_mask"].cuda(), use_first_frame_mask_condition_concat=True, use_fps_condition=True,
         fps_tensor=torch.tensor([2]), flow_control=torch.tensor([4]), use_ip_cross_attention=True, condition_images=image,
         callback=lambda i, t, l: traj.append(l.clone().cpu()), callback_steps=1)
    # oracle: same text states (the stub encoder is deterministic), image tokens from the restated CLIP tower + projection
    emb = E.clip_vision_forward(E.make_encoder_weights(E.clip_vision_shapes(E.TINY_VISION), 53), E.TINY_VISION, image)[1]
    ip_tokens = E.image_proj_forward(sd_p, torch.cat([torch.zeros_like(emb), emb]), 4, 64)
    ref = []
    Fn.denoise(sd, ocfg, Fn.DDIMConfig(), g["latents"], g["text_embeddings"], 3, 8.0, g["first_image_latents"], g["first_images_mask"],
               fps=torch.tensor([2]), flow=torch.tensor([4]), ip_tokens=ip_tokens, callback=lambda i, t, l: ref.append(l.clone()))
    traj, ref = torch.stack(traj), torch.stack(ref)
    err = (traj - ref).flatten(1).norm(dim=1) / ref.flatten(1).norm(dim=1)
    assert err.max().item() < 1e-3, err


@pytest.mark.parametrize("dtype,tol_lat,tol_vid", [(torch.float32, 1e-3, 1e-3), (torch.bfloat16, 1.5e-1, 1e-1)])
def test_animation_pipeline_plain_text_to_video(dropin, golden_dir, dtype, tol_lat, tol_vid):
    """the call scripts/inference_org.py makes: no concat conditioning, 4-channel UNet3D with motion modules"""
    from animatediff.models.unet import UNet3DConditionModel
    from animatediff.pipelines.pipeline_animation import AnimationPipeline
    from diffusers import AutoencoderKL, DDIMScheduler
    g = _load(golden_dir, "pipeline_tiny_t2v.npz")
    kw = dict(TINY, use_fps_condition=False, use_first_frame_mask_condition_concat=False)
    unet = UNet3DConditionModel(**kw, compute_dtype=dtype)
    ocfg = Fn.tiny_unet_config(use_fps_condition=False, use_first_frame_mask_condition_concat=False)
    assert not any(unet.load_state_dict(W.make_weights(W.unet_state_shapes(ocfg), int(g["unet_weight_seed"])), strict=False))
    vae = AutoencoderKL(block_out_channels=(64, 128, 128, 128), layers_per_block=2, latent_channels=4, compute_dtype=dtype)
    vae.load_state_dict(W.make_weights(W.vae_decoder_state_shapes(Fn.VAEConfig(block_out_channels=(64, 128, 128, 128))), int(g["vae_weight_seed"])), strict=False)
    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False)
    pipe = AnimationPipeline(vae=vae, text_encoder=stubs.StubTextEncoder(64), tokenizer=stubs.FakeTokenizer(), unet=unet, scheduler=sched).to("cuda")
    traj = []
    out = pipe("a corgi waving its tail", video_length=4, height=64, width=64, num_inference_steps=4, guidance_scale=7.5,
               negative_prompt="blurry", latents=g["latents"].clone(), callback=lambda i, t, l: traj.append(l.clone().cpu()), callback_steps=1)
    traj = torch.stack(traj)
    err = (traj - g["trajectory"]).flatten(1).norm(dim=1) / g["trajectory"].flatten(1).norm(dim=1)
    assert err.max().item() < tol_lat, err
    assert ((out.videos - g["videos"]).norm() / g["videos"].norm()).item() < tol_vid
    with pytest.raises(ValueError, match="built"):
        pipe("x", video_length=4, height=64, width=64, use_first_frame_mask_condition_concat=True, first_image_latents=torch.zeros(1, 4, 8, 8))
    # --video_scale of scripts/inference_org.py: per-frame unconditional pass + three-way guidance
    traj = []
    pipe("a corgi waving its tail", video_length=4, height=64, width=64, num_inference_steps=3, guidance_scale=7.5, negative_prompt="blurry",
         latents=g["latents"].clone(), video_scale=float(g["video_scale"]), callback=lambda i, t, l: traj.append(l.clone().cpu()), callback_steps=1)
    traj, ref = torch.stack(traj), g["trajectory_video_scale"]
    err = (traj - ref).flatten(1).norm(dim=1) / ref.flatten(1).norm(dim=1)
    assert err.max().item() < tol_lat, err
    # use_first_frame_condition: frame 0 pinned to the first-frame latents, timestep-0 embedding on that frame
    traj = []
    pipe("a corgi waving its tail", video_length=4, height=64, width=64, num_inference_steps=3, guidance_scale=7.5, negative_prompt="blurry",
         latents=g["latents"].clone(), use_first_frame_condition=True, first_image_latents=g["first_image_latents"].cuda(),
         callback=lambda i, t, l: traj.append(l.clone().cpu()), callback_steps=1)
    traj, ref = torch.stack(traj), g["trajectory_first_frame"]
    err = (traj - ref).flatten(1).norm(dim=1) / ref.flatten(1).norm(dim=1)
    assert err.max().item() < tol_lat, err


def test_partial_mask_on_first_frame_block(dropin, golden_dir):
    """use_first_frame_mask_condition_concat_image_partial_mask: the first-frame latents block is multiplied by the mask before the
    concat (reference pipeline_animation.py:698-699) == handing in pre-multiplied first-frame latents"""
    from animatediff.models.unet import UNet3DConditionModel
    from animatediff.pipelines.pipeline_animation import AnimationPipeline
    from diffusers import AutoencoderKL, DDIMScheduler
    g = _load(golden_dir, "pipeline_tiny.npz")
    unet = UNet3DConditionModel(**TINY, compute_dtype=torch.float32)
    unet.load_state_dict(W.make_weights(W.unet_state_shapes(Fn.tiny_unet_config()), int(g["unet_weight_seed"])), strict=False)
    vae = AutoencoderKL(block_out_channels=(64, 128, 128, 128), layers_per_block=2, latent_channels=4, compute_dtype=torch.float32)
    pipe = AnimationPipeline(vae=vae, text_encoder=stubs.StubTextEncoder(64), tokenizer=stubs.FakeTokenizer(), unet=unet,
                             scheduler=DDIMScheduler(steps_offset=1, clip_sample=False, prediction_type="v_prediction")).to("cuda")
    pm = (torch.rand(1, 1, 8, 8, generator=torch.Generator().manual_seed(3)) > 0.5).float().cuda()
    kw = dict(video_length=4, height=64, width=64, num_inference_steps=2, guidance_scale=8.0, latents=g["latents"].clone(),
              first_images_mask=g["first_images_mask"].cuda(), use_first_frame_mask_condition_concat=True, output_type="tensor")
    a = pipe("x", first_image_latents=g["first_image_latents"].cuda(), use_first_frame_mask_condition_concat_image_partial_mask=pm, **kw).videos
    b = pipe("x", first_image_latents=g["first_image_latents"].cuda() * pm, **kw).videos
    c = pipe("x", first_image_latents=g["first_image_latents"].cuda(), **kw).videos
    # run-to-run differences come from the summation order of the GroupNorm atomics only
    assert (a - b).abs().max().item() < 2e-3 and (a - c).abs().max().item() > 2e-2
