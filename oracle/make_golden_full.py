"""Golden vectors at SD-1.5 WIDTHS (320/640/1280/1280, ctx 768, 8 heads => d = 40/80/160) and at the
production precision, by executing the REAL reference (container only; TEST INFRASTRUCTURE).

    python -m oracle.make_golden_full small      # F=4, 16x16 latent, one forward            (~1 min)
    python -m oracle.make_golden_full ip         # the same forward with the IP-Adapter branch (16 image tokens)
    python -m oracle.make_golden_full vae        # AutoencoderKL.decode, (128,256,512,512), 16x16 latent
    python -m oracle.make_golden_full p2         # AnimationPipeline.prepare_latents (init-latents blend / interpolate noise)
    python -m oracle.make_golden_full cfg1       # BASELINE configs[0]: 8 frames 256x256, 5 DDIM steps, full pipeline
    python -m oracle.make_golden_full cfg2       # BASELINE configs[1]: 16 frames 512x512, 25 DDIM steps (~1.5 h on 8 cores)
    python -m oracle.make_golden_full cfg4       # BASELINE configs[3] paths: F=32 forward (24x24 latent) and a 96x96-latent forward, max_len 32
    python -m oracle.make_golden_full cfg5       # BASELINE configs[4]: 5-step IP-Adapter + rectangle-mask trajectory (4 frames 128x128)
    python -m oracle.make_golden_full cfg4full   # BASELINE configs[3] at its REAL shape: one forward, F=32, 96x96 latent, max_len 32 (~25 min on 8 cores)
    python -m oracle.make_golden_full cfg5full   # BASELINE configs[4] at its REAL shape: one CFG-pair forward, 16f@512^2, 16 IP tokens + rectangle mask (~10 min)
    python -m oracle.make_golden_full cfg5yard   # bf16-autocast drift of the NO-quirk oracle along the cfg5 trajectory (the engine's yardstick)
    python -m oracle.make_golden_full f16yard    # fp16-autocast drift of the REAL reference along the cfg1 and cfg2 trajectories and of the small
                                                 # full-width forward: the yardstick of the engine's FYC_F16 mode (the precision the reference deploys)

Every UNet golden exists twice: `*_f32` = the reference as the CPU runs it (fp32), `*_bf16` = the same
reference code under the CUDA-autocast cast policy with bfloat16 (oracle/autocast_emul.py), i.e. the
precision the engine's production mode computes in.  `drift` = rel-L2(bf16 reference, f32 reference) is
stored next to them: it is the yardstick for the engine's own bf16-vs-f32 distance.
Weights/inputs are re-derived from seeds (oracle/weights.py); only outputs are stored.
"""
import os
import sys
import time

import numpy as np
import torch

from . import functional as Fn
from . import refshim, stubs
from . import weights as W
from .autocast_emul import CudaAutocastOnCpu
from .make_golden import OUT, ref_unet, ref_vae

SKW = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1,
           clip_sample=False, prediction_type="v_prediction", rescale_betas_zero_snr=True)


def rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


def log(*a):
    print(time.strftime("%H:%M:%S"), *a, flush=True)


def full_unet(max_len=24):
    cfg = Fn.UNetConfig(temporal_position_encoding_max_len=max_len) if max_len != 24 else Fn.UNetConfig()
    t0 = time.time()
    unet = ref_unet(cfg).eval()
    sd = W.make_weights(W.unet_state_shapes(cfg), seed=0)
    unet.load_state_dict(sd, strict=True)
    for m in unet.modules():                       # score-tensor valve on the 16 spatial attn1 only: bit-identical math
        if m.__class__.__name__ == "BasicTransformerBlock" and hasattr(m, "attn1"):
            m.attn1._slice_size = 8
    log(f"reference UNet3D built + seeded in {time.time() - t0:.0f}s")
    return cfg, unet


def small():
    cfg, unet = full_unet()
    inp = W.seeded_inputs(cfg, 1, 4, 16, 16, seed=31)
    x9 = torch.cat([Fn.build_model_input(inp["latents"], inp["first_image_latents"], inp["first_images_mask"])] * 2)
    fps, flow = torch.tensor([2, 2]), torch.tensor([4, 4])
    kw = dict(use_fps_condition=True, fps_tensor=fps, flow_control=flow)
    with torch.no_grad():
        y32 = unet(x9, torch.tensor(961), inp["text"], **kw).sample
        with CudaAutocastOnCpu(torch.bfloat16):
            y16 = unet(x9, torch.tensor(961), inp["text"], **kw).sample.float()
    d = rel(y16, y32)
    log(f"small: drift bf16-autocast vs f32 = {d:.3e}")
    np.savez_compressed(os.path.join(OUT, "unet_full_small_fwd.npz"), out_f32=y32.numpy(), out_bf16=y16.numpy(), drift=np.float64(d),
                        timestep=np.int64(961), fps=fps.numpy(), flow=flow.numpy(), weight_seed=np.int64(0), input_seed=np.int64(31),
                        frames=np.int64(4), h=np.int64(16), w=np.int64(16))


def ip():
    """full-width UNet3D forward WITH the IP-Adapter branch (BASELINE configs[4]: 16 image tokens).  The reference's CPU code path
    (no xformers) runs attn2 with softmax temperature = the IP weight (IPCrossAttention.__init__ overwrites `scale`, SURVEY.md
    headline 6): the golden pins oracle.functional with `ip_reference_cpu_scale_quirk=True`; the engine implements the deployed
    (xformers) semantics and is held to the oracle without the quirk (tests/test_fullwidth_gpu.py)."""
    cfg = Fn.UNetConfig(use_ip_cross_attention=True, ip_num_tokens=16, ip_scale=0.7, ip_reference_cpu_scale_quirk=True)
    unet = ref_unet(cfg).eval()
    sd = W.make_weights(W.unet_state_shapes(cfg), seed=0)
    unet.load_state_dict(sd, strict=True)

    class Proj(torch.nn.Module):
        def forward(self, feat):
            return feat
    unet.image_proj_model = Proj()
    inp = W.seeded_inputs(cfg, 1, 4, 16, 16, seed=33)
    x9 = torch.cat([Fn.build_model_input(inp["latents"], inp["first_image_latents"], inp["first_images_mask"])] * 2)
    fps, flow = torch.tensor([2, 2]), torch.tensor([4, 4])
    kw = dict(use_fps_condition=True, fps_tensor=fps, flow_control=flow, use_ip_cross_attention=True, reference_images_clip_feat=inp["ip_tokens"])
    with torch.no_grad():
        y32 = unet(x9, torch.tensor(961), inp["text"], **kw).sample
        with CudaAutocastOnCpu(torch.bfloat16):
            y16 = unet(x9, torch.tensor(961), inp["text"], **kw).sample.float()
    d = rel(y16, y32)
    log(f"ip: drift bf16-autocast vs f32 = {d:.3e}")
    # round 6: the oracle's two runs are stored beside the reference's - WITH the CPU-path quirk (pinned to the reference's output here, at
    # generation time, and again in tests/test_oracle_golden.py) and WITHOUT it = the deployed semantics the engine is held to
    # (tests/test_fullwidth_gpu.py used to run this 15-s CPU forward inside the GPU test)
    with torch.no_grad():
        o_q = Fn.unet3d_forward(sd, cfg, x9, torch.tensor(961), inp["text"], fps, flow, inp["ip_tokens"])
        ncfg = Fn.UNetConfig(use_ip_cross_attention=True, ip_num_tokens=16, ip_scale=0.7)
        o_n = Fn.unet3d_forward(sd, ncfg, x9, torch.tensor(961), inp["text"], fps, flow, inp["ip_tokens"])
    log(f"ip: oracle with the quirk vs reference {rel(o_q, y32):.3e}; oracle without it vs reference {rel(o_n, y32):.3e}")
    assert rel(o_q, y32) < 1e-4
    np.savez_compressed(os.path.join(OUT, "unet_full_ip_fwd.npz"), out_f32=y32.numpy(), out_bf16=y16.numpy(), drift=np.float64(d),
                        out_oracle_noquirk=o_n.numpy(), oracle_quirk_vs_ref=np.float64(rel(o_q, y32)),
                        timestep=np.int64(961), fps=fps.numpy(), flow=flow.numpy(), weight_seed=np.int64(0), input_seed=np.int64(33),
                        frames=np.int64(4), h=np.int64(16), w=np.int64(16), ip_scale=np.float64(0.7), ip_num_tokens=np.int64(16))


def vae():
    vcfg = Fn.VAEConfig()
    v = ref_vae(vcfg).eval()
    sdv = W.make_weights(W.vae_decoder_state_shapes(vcfg), seed=3)
    v.load_state_dict(sdv, strict=False)
    z = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        img = v.decode(z).sample
        with CudaAutocastOnCpu(torch.bfloat16):
            img16 = v.decode(z).sample.float()
    log(f"vae: max|bf16-f32| = {(img16 - img).abs().max():.3e}")
    np.savez_compressed(os.path.join(OUT, "vae_full.npz"), z=z.numpy(), out=img.numpy(), out_bf16=img16.numpy(), weight_seed=np.int64(3))


def _pipeline(unet, cfg, vae_mod=None):
    from animatediff.pipelines.pipeline_animation import AnimationPipeline
    from diffusers.schedulers.scheduling_ddim import DDIMScheduler
    if vae_mod is None:
        vcfg = Fn.VAEConfig(block_out_channels=(64, 128, 128, 128))
        vae_mod = ref_vae(vcfg).eval()
        vae_mod.load_state_dict(W.make_weights(W.vae_decoder_state_shapes(vcfg), seed=3), strict=False)
    tok, txt = stubs.FakeTokenizer(), stubs.StubTextEncoder(cfg.cross_attention_dim)
    return AnimationPipeline(vae=vae_mod, text_encoder=txt, tokenizer=tok, unet=unet, scheduler=DDIMScheduler(**SKW))


def p2():
    """prepare_latents (reference pipeline_animation.py:448-536): generated noise with / without use_interpolate_noise, the
    init_latents blend in both branches (:501-508, :526-532), use_residual_noise (:509-513); the generator is seeded so the
    host restatement must draw the same noise."""
    cfg = Fn.tiny_unet_config()
    unet = ref_unet(cfg).eval()
    pipe = _pipeline(unet, cfg)
    d = {}
    first = 0.18215 * torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(41))
    mask = (torch.rand(1, 1, 6, 8, 8, generator=torch.Generator().manual_seed(42)) > 0.5).float()
    given = torch.randn(1, 4, 6, 8, 8, generator=torch.Generator().manual_seed(43))
    cases = (("gen_interp", dict()),
             ("gen_plain", dict(use_interpolate_noise=False)),
             ("gen_init", dict(use_interpolate_noise=False, init_latents=first, first_images_mask=mask)),
             ("gen_init_interp", dict(init_latents=first, first_images_mask=mask)),
             ("gen_init_nomask", dict(init_latents=first)),
             ("gen_residual", dict(use_interpolate_noise=False, use_residual_noise=True, base_lambda=0.9)),
             ("given_init", dict(latents=given.clone(), init_latents=first, k=30)),
             ("given_plain", dict(latents=given.clone())),
             ("given_badshape", dict(latents=given[:, :, :5].clone())))
    import contextlib
    import io
    for name, kw in cases:
        g = torch.Generator().manual_seed(77)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                lat = pipe.prepare_latents(1, 4, 6, 64, 64, torch.float32, torch.device("cpu"), g, **kw)
            d[name] = lat.numpy()
            log("p2", name, tuple(lat.shape), float(lat.std()))
        except Exception as e:       # recorded: the host restatement must raise as well
            d[name + "_error"] = np.array(type(e).__name__)
            log("p2", name, "raises", repr(e)[:120])
    d["first_image_latents"], d["first_images_mask"], d["given"] = first.numpy(), mask.numpy(), given.numpy()
    np.savez_compressed(os.path.join(OUT, "prepare_latents.npz"), **d)


def _run_pipeline(pipe, inp, frames, size, steps, keep):
    traj = {}

    def cb(i, t, l):
        log(f"  step {i} t={int(t)}")
        if i in keep:
            traj[i] = l.clone().float()

    out = pipe("a corgi waving its tail", video_length=frames, height=size, width=size, num_inference_steps=steps, guidance_scale=8.0,
               negative_prompt="blurry", latents=inp["latents"].clone(), first_image_latents=inp["first_image_latents"],
               first_images_mask=inp["first_images_mask"], use_first_frame_mask_condition_concat=True,
               use_fps_condition=True, fps_tensor=torch.tensor([2]), flow_control=torch.tensor([4]),
               callback=cb, callback_steps=1, output_type="latent")
    return traj, out


def _trajectory(tag, frames, lat, steps, keep, seed):
    cfg, unet = full_unet()
    pipe = _pipeline(unet, cfg)
    pipe.decode_latents = lambda latents: np.zeros((1, 3, frames, 8, 8), dtype=np.float32)   # VAE decode is pinned by vae_full / pipeline_tiny
    inp = W.seeded_inputs(cfg, 1, frames, lat, lat, seed=seed)
    with torch.no_grad():
        text_emb = pipe._encode_prompt(["a corgi waving its tail"], "cpu", 1, True, ["blurry"])
    res = {}
    for mode in ("bf16", "f32"):
        t0 = time.time()
        log(f"{tag} {mode}: {steps} steps")
        if mode == "bf16":
            with CudaAutocastOnCpu(torch.bfloat16):
                traj, _ = _run_pipeline(pipe, inp, frames, lat * 8, steps, keep)
        else:
            traj, _ = _run_pipeline(pipe, inp, frames, lat * 8, steps, keep)
        res[mode] = traj
        log(f"{tag} {mode}: {time.time() - t0:.0f}s")
        # partial save after each mode so a long run leaves something behind
        d = dict(text_embeddings=text_emb.numpy(), steps=np.int64(steps), frames=np.int64(frames), lat=np.int64(lat),
                 weight_seed=np.int64(0), input_seed=np.int64(seed), keep=np.array(sorted(keep)))
        for m, tr in res.items():
            for i, l in tr.items():
                d[f"step{i}_{m}"] = l.numpy()
        if "f32" in res:
            for i in sorted(keep):
                d[f"drift{i}"] = np.float64(rel(res["bf16"][i], res["f32"][i]))
                log(f"{tag}: drift bf16-autocast vs f32 after step {i}: {d[f'drift{i}']:.3e}")
        np.savez_compressed(os.path.join(OUT, f"{tag}_trajectory.npz"), **d)


def cfg4():
    """BASELINE configs[3] (32 frames 768x768, temporal_position_encoding_max_len 32): the two paths it adds to cfg2 -
    (a) F = 32: the 32 x 32-score temporal attention at d = 40 / 80 / 160 and the 32-row positional table (motion_module.py:286-304,
        371-464), one forward at a reduced 24x24 latent;
    (b) a 96x96 latent: spatial attention over N = 9216 tokens (F = 2 keeps the CPU run to minutes).
    Both with the max_len = 32 UNet, f32 and bf16-autocast."""
    cfg, unet = full_unet(max_len=32)
    fps, flow = torch.tensor([2, 2]), torch.tensor([4, 4])
    kw = dict(use_fps_condition=True, fps_tensor=fps, flow_control=flow)
    d = dict(timestep=np.int64(961), fps=fps.numpy(), flow=flow.numpy(), weight_seed=np.int64(0), max_len=np.int64(32))
    for tag, F, lat, seed in (("f32x24", 32, 24, 61), ("f2x96", 2, 96, 62)):
        inp = W.seeded_inputs(cfg, 1, F, lat, lat, seed=seed)
        x9 = torch.cat([Fn.build_model_input(inp["latents"], inp["first_image_latents"], inp["first_images_mask"])] * 2)
        t0 = time.time()
        with torch.no_grad():
            y32 = unet(x9, torch.tensor(961), inp["text"], **kw).sample
            with CudaAutocastOnCpu(torch.bfloat16):
                y16 = unet(x9, torch.tensor(961), inp["text"], **kw).sample.float()
        dr = rel(y16, y32)
        log(f"cfg4 {tag}: F={F} latent {lat}x{lat}: drift bf16-autocast vs f32 = {dr:.3e} ({time.time() - t0:.0f}s)")
        d.update({f"{tag}_out_f32": y32.numpy(), f"{tag}_out_bf16": y16.numpy(), f"{tag}_drift": np.float64(dr), f"{tag}_frames": np.int64(F),
                  f"{tag}_lat": np.int64(lat), f"{tag}_input_seed": np.int64(seed)})
        np.savez_compressed(os.path.join(OUT, "cfg4_forwards.npz"), **d)


def cfg5():
    """BASELINE configs[4] as a TRAJECTORY: AnimationPipeline.__call__ with the IP-Adapter branch (16 image tokens, scale 0.7), the
    rectangle region mask + first-frame latent concat, CFG 8, 5 DDIM steps, full SD-1.5 widths, 4 frames 128x128 (pipeline_animation.py
    :676-680, 716-723; animatediff/models/attention.py:49-127).  The reference's CPU code path runs attn2 at the IP weight as softmax
    temperature (SURVEY headline 6), so three trajectories are stored: the real reference (f32, and bf16-autocast for the drift),
    and oracle.functional WITHOUT that quirk (= the deployed xformers semantics the engine implements).  tests/test_oracle_golden.py
    holds the oracle WITH the quirk to the real reference; tests/test_fullwidth_gpu.py holds the engine to the oracle without it."""
    frames, lat, steps, seed = 4, 16, 5, 63
    cfg = Fn.UNetConfig(use_ip_cross_attention=True, ip_num_tokens=16, ip_scale=0.7, ip_reference_cpu_scale_quirk=True)
    unet = ref_unet(cfg).eval()
    sd = W.make_weights(W.unet_state_shapes(cfg), seed=0)
    unet.load_state_dict(sd, strict=True)

    class Proj(torch.nn.Module):
        def forward(self, feat):
            return feat
    unet.image_proj_model = Proj()
    pipe = _pipeline(unet, cfg)
    pipe.decode_latents = lambda latents: np.zeros((1, 3, frames, 8, 8), dtype=np.float32)
    inp = W.seeded_inputs(cfg, 1, frames, lat, lat, seed=seed)
    mask = torch.zeros(1, 1, 1, lat, lat)
    mask[..., lat // 4: 3 * lat // 4, lat // 4: 3 * lat // 4] = 1.0          # the rectangle of SURVEY.md 8d (a SAM box stand-in)
    inp["first_images_mask"] = mask

    class IP:                                                   # MyIPAdapter.get_image_clip_feat stand-in: seeded "CLIP features"
        def get_image_clip_feat(self, input_image=None):
            return inp["ip_tokens"][1:2], inp["ip_tokens"][0:1]
    pipe.ip_adapter = IP()
    with torch.no_grad():
        text_emb = pipe._encode_prompt(["a corgi waving its tail"], "cpu", 1, True, ["blurry"])

    def run():
        traj = {}

        def cb(i, t, l):
            traj[i] = l.clone().float()
        pipe("a corgi waving its tail", video_length=frames, height=lat * 8, width=lat * 8, num_inference_steps=steps, guidance_scale=8.0,
             negative_prompt="blurry", latents=inp["latents"].clone(), first_image_latents=inp["first_image_latents"],
             first_images_mask=inp["first_images_mask"], use_first_frame_mask_condition_concat=True, use_fps_condition=True,
             fps_tensor=torch.tensor([2]), flow_control=torch.tensor([4]), use_ip_cross_attention=True, condition_images=[None],
             callback=cb, callback_steps=1, output_type="latent")
        return traj
    with torch.no_grad():
        t32 = run()
        with CudaAutocastOnCpu(torch.bfloat16):
            t16 = run()
        ocfg = Fn.UNetConfig(use_ip_cross_attention=True, ip_num_tokens=16, ip_scale=0.7)
        tor = {}
        Fn.denoise(sd, ocfg, Fn.DDIMConfig(), inp["latents"].clone(), text_emb, steps, 8.0, inp["first_image_latents"], inp["first_images_mask"],
                   torch.tensor([2]), torch.tensor([4]), ip_tokens=inp["ip_tokens"], callback=lambda i, t, l: tor.__setitem__(i, l.clone().float()))
    d = dict(text_embeddings=text_emb.numpy(), steps=np.int64(steps), frames=np.int64(frames), lat=np.int64(lat), weight_seed=np.int64(0),
             input_seed=np.int64(seed), ip_scale=np.float64(0.7), ip_num_tokens=np.int64(16), first_images_mask=mask.numpy())
    for i in range(steps):
        d[f"step{i}_ref_f32"], d[f"step{i}_ref_bf16"], d[f"step{i}_oracle_noquirk"] = t32[i].numpy(), t16[i].numpy(), tor[i].numpy()
        d[f"drift{i}"] = np.float64(rel(t16[i], t32[i]))
        log(f"cfg5 step {i}: ref drift bf16 vs f32 {d[f'drift{i}']:.3e}; oracle(no quirk) vs reference(CPU quirk) {rel(tor[i], t32[i]):.3e}")
    np.savez_compressed(os.path.join(OUT, "cfg5_trajectory.npz"), **d)


CFG4_KEEP = (0, 5, 10, 15, 20, 25, 31)     # frames of the full-shape forward that are stored whole (every frame's norm is stored)


def cfg4full():
    """BASELINE configs[3] at the shape the bench line runs: ONE forward of the real reference at F = 32 frames on a 96x96 latent with the
    32-row positional table - 18 432 pixels x 8 heads of 32x32 temporal scores together with 64 frames of 9 216-token spatial attention
    (motion_module.py:286-304, 371-464; diffusers/models/attention.py:649-678).  181 TFLOP per forward: f32, then bf16-autocast.
    Stored: the frames CFG4_KEEP of both outputs (the f32 one in f32, the bf16 one in f16 - 5e-4 relative, 25x below the drift), the
    per-frame L2 norms of every frame of the f32 output, and the drift over the whole tensor."""
    cfg, unet = full_unet(max_len=32)
    F, lat, seed = 32, 96, 64
    fps, flow = torch.tensor([2, 2]), torch.tensor([4, 4])
    kw = dict(use_fps_condition=True, fps_tensor=fps, flow_control=flow)
    inp = W.seeded_inputs(cfg, 1, F, lat, lat, seed=seed)
    x9 = torch.cat([Fn.build_model_input(inp["latents"], inp["first_image_latents"], inp["first_images_mask"])] * 2)
    keep = list(CFG4_KEEP)
    with torch.no_grad():
        t0 = time.time()
        y32 = unet(x9, torch.tensor(961), inp["text"], **kw).sample
        log(f"cfg4full f32 forward {time.time() - t0:.0f}s")
        np.savez_compressed(os.path.join(OUT, "cfg4_full_shape.partial.npz"), out_f32=y32[:, :, keep].numpy())
        t0 = time.time()
        with CudaAutocastOnCpu(torch.bfloat16):
            y16 = unet(x9, torch.tensor(961), inp["text"], **kw).sample.float()
        log(f"cfg4full bf16-autocast forward {time.time() - t0:.0f}s")
    dr = rel(y16, y32)
    dr_keep = rel(y16[:, :, keep], y32[:, :, keep])
    log(f"cfg4full: drift bf16-autocast vs f32 = {dr:.3e} (on the stored frames {dr_keep:.3e})")
    np.savez_compressed(os.path.join(OUT, "cfg4_full_shape.npz"), out_f32=y32[:, :, keep].numpy(), out_bf16=y16[:, :, keep].numpy().astype(np.float16),
                        frame_norms_f32=y32.pow(2).sum(dim=(1, 3, 4)).sqrt().numpy(), keep=np.array(keep), drift=np.float64(dr), drift_keep=np.float64(dr_keep),
                        timestep=np.int64(961), fps=fps.numpy(), flow=flow.numpy(), weight_seed=np.int64(0), input_seed=np.int64(seed), max_len=np.int64(32),
                        frames=np.int64(F), lat=np.int64(lat))
    os.remove(os.path.join(OUT, "cfg4_full_shape.partial.npz"))


def _ip_setup(quirk: bool):
    cfg = Fn.UNetConfig(use_ip_cross_attention=True, ip_num_tokens=16, ip_scale=0.7, ip_reference_cpu_scale_quirk=quirk)
    sd = W.make_weights(W.unet_state_shapes(cfg), seed=0)
    return cfg, sd


def cfg5full():
    """BASELINE configs[4] at its REAL shape: one CFG-pair forward at 16 frames on a 64x64 latent with 16 IP tokens, the rectangle region mask
    and the first-frame latent concat (pipeline_animation.py:676-680, 716-723; animatediff/models/attention.py:49-127).  Four forwards:
    the real reference (its CPU path carries the attn2 temperature quirk), oracle.functional WITH the quirk (must reproduce it - asserted
    here, the value is stored), oracle.functional WITHOUT it (= the deployed xformers semantics the engine implements) in f32 and under
    bf16-autocast: that last pair's distance is the yardstick for the engine's bf16 mode."""
    F, lat, seed = 16, 64, 65
    cfg, sd = _ip_setup(True)
    unet = ref_unet(cfg).eval()
    unet.load_state_dict(sd, strict=True)
    for m in unet.modules():
        if m.__class__.__name__ == "BasicTransformerBlock" and hasattr(m, "attn1"):
            m.attn1._slice_size = 8

    class Proj(torch.nn.Module):
        def forward(self, feat):
            return feat
    unet.image_proj_model = Proj()
    inp = W.seeded_inputs(cfg, 1, F, lat, lat, seed=seed)
    mask = torch.zeros(1, 1, 1, lat, lat)
    mask[..., lat // 4: 3 * lat // 4, lat // 4: 3 * lat // 4] = 1.0
    x9 = torch.cat([Fn.build_model_input(inp["latents"], inp["first_image_latents"], mask)] * 2)
    fps, flow = torch.tensor([2, 2]), torch.tensor([4, 4])
    kw = dict(use_fps_condition=True, fps_tensor=fps, flow_control=flow, use_ip_cross_attention=True, reference_images_clip_feat=inp["ip_tokens"])
    with torch.no_grad():
        t0 = time.time()
        yref = unet(x9, torch.tensor(961), inp["text"], **kw).sample
        log(f"cfg5full reference f32 forward {time.time() - t0:.0f}s")
        del unet
        t0 = time.time()
        yq = Fn.unet3d_forward(sd, cfg, x9, torch.tensor(961), inp["text"], fps, flow, ip_tokens=inp["ip_tokens"])
        log(f"cfg5full oracle (quirk) forward {time.time() - t0:.0f}s: vs reference {rel(yq, yref):.3e}")
        assert rel(yq, yref) < 1e-4
        ocfg, _ = _ip_setup(False)
        yo = Fn.unet3d_forward(sd, ocfg, x9, torch.tensor(961), inp["text"], fps, flow, ip_tokens=inp["ip_tokens"])
        with CudaAutocastOnCpu(torch.bfloat16):
            yo16 = Fn.unet3d_forward(sd, ocfg, x9, torch.tensor(961), inp["text"], fps, flow, ip_tokens=inp["ip_tokens"]).float()
    dr = rel(yo16, yo)
    log(f"cfg5full: oracle(no quirk) vs reference(CPU quirk) {rel(yo, yref):.3e}; drift of the no-quirk oracle bf16-autocast vs f32 = {dr:.3e}")
    np.savez_compressed(os.path.join(OUT, "cfg5_full_shape.npz"), out_ref_f32=yref.numpy(), out_oracle_noquirk=yo.numpy(), out_oracle_noquirk_bf16=yo16.numpy().astype(np.float16),
                        oracle_quirk_vs_ref=np.float64(rel(yq, yref)), drift_noquirk=np.float64(dr), first_images_mask=mask.numpy(), timestep=np.int64(961),
                        fps=fps.numpy(), flow=flow.numpy(), weight_seed=np.int64(0), input_seed=np.int64(seed), frames=np.int64(F), lat=np.int64(lat),
                        ip_scale=np.float64(0.7), ip_num_tokens=np.int64(16))


def cfg5yard():
    """The yardstick the cfg5 TRAJECTORY's bf16 bound was missing: the reference's stored drift is of its CPU path (wrong attn2 temperature),
    so here the NO-quirk oracle runs the same 5 steps under bf16-autocast; drift_noquirk{i} = rel-L2 to its own f32 run (which must
    reproduce cfg5_trajectory.npz's `step{i}_oracle_noquirk` - asserted)."""
    g = np.load(os.path.join(OUT, "cfg5_trajectory.npz"))
    frames, lat, steps, seed = int(g["frames"]), int(g["lat"]), int(g["steps"]), int(g["input_seed"])
    ocfg, sd = _ip_setup(False)
    inp = W.seeded_inputs(ocfg, 1, frames, lat, lat, seed=seed)
    text_emb, mask = torch.from_numpy(g["text_embeddings"]), torch.from_numpy(g["first_images_mask"])

    def run():
        tr = {}
        Fn.denoise(sd, ocfg, Fn.DDIMConfig(), inp["latents"].clone(), text_emb, steps, 8.0, inp["first_image_latents"], mask,
                   torch.tensor([2]), torch.tensor([4]), ip_tokens=inp["ip_tokens"], callback=lambda i, t, l: tr.__setitem__(i, l.clone().float()))
        return tr
    with torch.no_grad():
        t32 = run()
        with CudaAutocastOnCpu(torch.bfloat16):
            t16 = run()
    d = dict(steps=np.int64(steps))
    for i in range(steps):
        assert rel(t32[i], torch.from_numpy(g[f"step{i}_oracle_noquirk"])) < 1e-5
        d[f"drift_noquirk{i}"] = np.float64(rel(t16[i], t32[i]))
        log(f"cfg5yard step {i}: no-quirk oracle bf16-autocast vs f32 {d[f'drift_noquirk{i}']:.3e} (reference CPU-path drift {float(g[f'drift{i}']):.3e})")
    np.savez_compressed(os.path.join(OUT, "cfg5_yardstick.npz"), **d)


def f16yard():
    """What the reference itself loses when it runs as deployed - `torch.autocast("cuda")` = float16 (scripts/inference.py:294) -
    measured against its own f32 run: the yardstick for the engine's FYC_F16 mode, next to the bf16 `drift` values the trajectory
    goldens already hold.  The f32 runs are NOT repeated: the stored `step{i}_f32` tensors of cfg1 / cfg2_trajectory.npz and the stored
    f32 output of unet_full_small_fwd.npz are the reference side (the fp16 run starts from the same seeded weights and inputs).
    Only scalars are stored (tests/golden/f16_yardstick.npz); partial results are saved after every part."""
    d = {}
    path = os.path.join(OUT, "f16_yardstick.npz")
    # (1) the small full-width forward (F = 4, 16 x 16 latent)
    g = np.load(os.path.join(OUT, "unet_full_small_fwd.npz"))
    cfg, unet = full_unet()
    inp = W.seeded_inputs(cfg, 1, int(g["frames"]), int(g["h"]), int(g["w"]), seed=int(g["input_seed"]))
    x9 = torch.cat([Fn.build_model_input(inp["latents"], inp["first_image_latents"], inp["first_images_mask"])] * 2)
    kw = dict(use_fps_condition=True, fps_tensor=torch.from_numpy(g["fps"]), flow_control=torch.from_numpy(g["flow"]))
    with torch.no_grad():
        with CudaAutocastOnCpu(torch.float16):
            y16 = unet(x9, torch.tensor(int(g["timestep"])), inp["text"], **kw).sample.float()
    d["small_fwd_drift_f16"] = np.float64(rel(y16, torch.from_numpy(g["out_f32"])))
    d["small_fwd_drift_bf16"] = np.float64(g["drift"])
    log(f"f16yard small forward: fp16-autocast vs f32 {d['small_fwd_drift_f16']:.3e} (bf16-autocast: {float(g['drift']):.3e})")
    np.savez_compressed(path, **d)
    pipe = _pipeline(unet, cfg)
    for tag in ("cfg1", "cfg2"):
        t = np.load(os.path.join(OUT, f"{tag}_trajectory.npz"))
        frames, lat, steps, seed = int(t["frames"]), int(t["lat"]), int(t["steps"]), int(t["input_seed"])
        keep = set(int(i) for i in t["keep"])
        pipe.decode_latents = lambda latents, _f=frames: np.zeros((1, 3, _f, 8, 8), dtype=np.float32)
        inp = W.seeded_inputs(cfg, 1, frames, lat, lat, seed=seed)
        t0 = time.time()
        log(f"f16yard {tag}: {steps} steps under fp16 autocast")
        with CudaAutocastOnCpu(torch.float16):
            traj, _ = _run_pipeline(pipe, inp, frames, lat * 8, steps, keep)
        for i in sorted(keep):
            d[f"{tag}_drift_f16_{i}"] = np.float64(rel(traj[i], torch.from_numpy(t[f"step{i}_f32"])))
            d[f"{tag}_drift_bf16_{i}"] = np.float64(t[f"drift{i}"])
            log(f"f16yard {tag} step {i}: fp16-autocast vs f32 {d[f'{tag}_drift_f16_{i}']:.3e} (bf16-autocast: {float(t[f'drift{i}']):.3e})")
        log(f"f16yard {tag}: {time.time() - t0:.0f}s")
        np.savez_compressed(path, **d)
    np.savez_compressed(path, **d)


def cfg1():
    _trajectory("cfg1", 8, 32, 5, {0, 1, 2, 3, 4}, seed=51)


def cfg2():
    _trajectory("cfg2", 16, 64, 25, {0, 4, 24}, seed=52)


if __name__ == "__main__":
    refshim.install()
    torch.manual_seed(0)
    for part in sys.argv[1:]:
        log("==", part)
        dict(small=small, ip=ip, vae=vae, p2=p2, cfg1=cfg1, cfg2=cfg2, cfg4=cfg4, cfg5=cfg5, cfg4full=cfg4full, cfg5full=cfg5full, cfg5yard=cfg5yard, f16yard=f16yard)[part]()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
