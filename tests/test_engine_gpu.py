"""End-to-end parity of the HIP engine on a real MI355X against golden vectors produced by the REAL
reference (tests/golden, see oracle/make_golden.py) and against the CPU oracle.

Tolerances (BASELINE.json north_star asks for 1e-3 rel-err; SURVEY.md 7 explains why that is only
meaningful at equal precision):
  * f32 parity mode  (v_mfma_f32_16x16x4_f32 GEMMs, f32 norms/softmax): rel-L2 <= 1e-3 vs the f32
    reference after all DDIM steps (measured ~1e-5);
  * bf16 production mode (bf16 MFMA operands, f32 accumulation/statistics): rel-L2 <= 5e-2 vs the f32
    reference for one UNet forward with random weights (the reference's own bf16-autocast drifts
    1.4e-2 from its f32 run on this model, SURVEY.md headline 5).
"""
import os

import numpy as np
import pytest
import torch

from followyourclick_amd.engine import DDIMConfig, UNet3DConfig, VAEDecoderConfig
from followyourclick_amd.engine.sampler import DDIMSampler
from followyourclick_amd.engine.unet3d import UNet3DEngine
from followyourclick_amd.engine.vae import VAEDecoderEngine
from followyourclick_amd.engine.weights import pack_unet, pack_vae_decoder
from oracle import functional as Fn
from oracle import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def tiny_cfg(**kw):
    return UNet3DConfig(block_out_channels=(64, 128, 256, 256), cross_attention_dim=64, sample_size=8, **kw)


def _load(golden_dir, name):
    return {k: torch.from_numpy(v) if v.shape else v for k, v in np.load(os.path.join(golden_dir, name)).items()}


def _nhwc(x9, dtype):
    B, C9, F, H, Wd = x9.shape
    x = torch.zeros(B * F * H * Wd, 64)
    x[:, :C9] = x9.permute(0, 2, 3, 4, 1).reshape(-1, C9)
    return x.to(dtype).to(DEV)


def rel(a, b):
    return ((a.float().cpu() - b).norm() / b.norm()).item()


def report(line):
    """measured parity numbers are appended to gpurun_out/parity_report.txt (copied into DESIGN.md)"""
    print(line)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "parity_report.txt"), "a") as f:
        f.write(line + "\n")


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 5e-2)])
def test_unet_forward_vs_reference_golden(golden_dir, dtype, tol):
    g = _load(golden_dir, "unet_tiny_fwd.npz")
    sd = W.make_weights(W.unet_state_shapes(Fn.tiny_unet_config()), int(g["weight_seed"]))
    eng = UNet3DEngine(pack_unet(sd, tiny_cfg(), dtype, DEV))
    assert eng.ops.name == "hip"
    B, _, F, H, Wd = g["sample"].shape
    eng.prepare_context(g["text"])
    _, temb = eng.prepare_time_embeddings([int(g["timestep"])], g["fps"].tolist(), g["flow"].tolist(), B)
    out = eng.forward(_nhwc(g["sample"], dtype), temb, B, F, H, Wd)
    torch.cuda.synchronize()
    out = out.float().cpu().reshape(B, F, H, Wd, 4).permute(0, 4, 1, 2, 3)
    assert torch.isfinite(out).all()
    r = rel(out, g["out"])
    report(f"unet fwd {dtype}: rel-L2 {r:.3e}")
    assert r < tol, r


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 5e-2)])
def test_unet_forward_odd_latent_size(golden_dir, dtype, tol):
    """10x12 latent (not a multiple of 8): ceil-halving downsamples, upsampling to the skip's size"""
    g = _load(golden_dir, "unet_tiny_odd_fwd.npz")
    sd = W.make_weights(W.unet_state_shapes(Fn.tiny_unet_config()), int(g["weight_seed"]))
    eng = UNet3DEngine(pack_unet(sd, tiny_cfg(), dtype, DEV))
    B, _, F, H, Wd = g["sample"].shape
    eng.prepare_context(g["text"])
    _, temb = eng.prepare_time_embeddings([int(g["timestep"])], g["fps"].tolist(), g["flow"].tolist(), B)
    out = eng.forward(_nhwc(g["sample"], dtype), temb, B, F, H, Wd).float().cpu().reshape(B, F, H, Wd, 4).permute(0, 4, 1, 2, 3)
    r = rel(out, g["out"])
    report(f"unet fwd odd size {dtype}: rel-L2 {r:.3e}")
    assert r < tol, r


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 5e-2)])
def test_unet_forward_ip_adapter_vs_oracle(dtype, tol):
    ocfg = Fn.tiny_unet_config(use_ip_cross_attention=True, ip_scale=0.7)
    sd = W.make_weights(W.unet_state_shapes(ocfg), 0)
    inp = W.seeded_inputs(ocfg, 1, 4, 8, 8, seed=7)
    x9 = torch.cat([Fn.build_model_input(inp["latents"], inp["first_image_latents"], inp["first_images_mask"])] * 2)
    with torch.no_grad():
        ref = Fn.unet3d_forward(sd, ocfg, x9, torch.tensor(961), inp["text"], torch.tensor([2, 2]), torch.tensor([4, 4]), inp["ip_tokens"])
    eng = UNet3DEngine(pack_unet(sd, tiny_cfg(use_ip_cross_attention=True, ip_scale=0.7), dtype, DEV))
    B, _, F, H, Wd = x9.shape
    eng.prepare_context(inp["text"], inp["ip_tokens"])
    _, temb = eng.prepare_time_embeddings([961], [2, 2], [4, 4], B)
    out = eng.forward(_nhwc(x9, dtype), temb, B, F, H, Wd).float().cpu().reshape(B, F, H, Wd, 4).permute(0, 4, 1, 2, 3)
    r = rel(out, ref)
    report(f"unet ip fwd {dtype}: rel-L2 {r:.3e}")
    assert r < tol, r


@pytest.mark.parametrize("dtype,tol_lat,tol_vid", [(torch.float32, 1e-3, 1e-3), (torch.bfloat16, 1.05e-1, 5.5e-2)])      # bf16: 2 x the measured 5.2e-2 / 2.7e-2
def test_sampling_loop_vs_reference_pipeline(golden_dir, dtype, tol_lat, tol_vid):
    """AnimationPipeline.__call__ of the real reference: 5 DDIM steps, CFG 8, mask + first frame,
    fps/flow conditioning; per-step latents and the decoded video."""
    g = _load(golden_dir, "pipeline_tiny.npz")
    sd = W.make_weights(W.unet_state_shapes(Fn.tiny_unet_config()), int(g["unet_weight_seed"]))
    eng = UNet3DEngine(pack_unet(sd, tiny_cfg(), dtype, DEV))
    traj = []
    lat = DDIMSampler(eng, DDIMConfig()).sample(g["latents"], g["text_embeddings"], 5, 8.0, g["first_image_latents"],
                                                g["first_images_mask"], fps=[2], flow=[4],
                                                callback=lambda i, t, l: traj.append(l.clone().cpu()))
    torch.cuda.synchronize()
    traj = torch.stack(traj)
    err = (traj - g["trajectory"]).flatten(1).norm(dim=1) / g["trajectory"].flatten(1).norm(dim=1)
    report(f"sampling {dtype}: per-step rel-L2 {[f'{e:.2e}' for e in err.tolist()]}")
    assert err.max().item() < tol_lat, err
    vcfg = VAEDecoderConfig(block_out_channels=(64, 128, 128, 128))
    sdv = W.make_weights(W.vae_decoder_state_shapes(Fn.VAEConfig(block_out_channels=(64, 128, 128, 128))), int(g["vae_weight_seed"]))
    vae = VAEDecoderEngine(pack_vae_decoder(sdv, vcfg, dtype, DEV))
    vid = vae.decode_video(lat).cpu()
    assert vid.shape == g["videos"].shape
    e = rel(vid, g["videos"])
    report(f"video {dtype}: rel-L2 {e:.3e}, max abs err {(vid - g['videos']).abs().max().item():.3e}")
    assert e < tol_vid, e


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sampling_loop_graph_replay_equals_eager(golden_dir, dtype):
    """steps 1..n-1 replayed from one captured hipGraph give the same BITS as eager launches, and so do two eager runs: every
    reduction on the path has a fixed order (round 3: float atomics in the GroupNorm statistics, bound 1.5e-1 in bf16)"""
    g = _load(golden_dir, "pipeline_tiny.npz")
    sd = W.make_weights(W.unet_state_shapes(Fn.tiny_unet_config()), int(g["unet_weight_seed"]))
    smp = DDIMSampler(UNet3DEngine(pack_unet(sd, tiny_cfg(), dtype, DEV)), DDIMConfig())
    args = (g["latents"], g["text_embeddings"], 5, 8.0, g["first_image_latents"], g["first_images_mask"])
    seen = []
    eager = smp.sample(*args, fps=[2], flow=[4], use_graph=False).cpu()
    again = smp.sample(*args, fps=[2], flow=[4], use_graph=False).cpu()
    graph = smp.sample(*args, fps=[2], flow=[4], use_graph=True, callback=lambda i, t, l: seen.append(i)).cpu()
    assert seen == [0, 1, 2, 3, 4]
    assert torch.equal(eager, again), rel(eager, again)
    assert torch.equal(graph, eager), rel(graph, eager)


def test_graph_replay_honours_use_clipped_model_output(golden_dir):
    """round-5 advisor: the hipGraph path of DDIMSampler.sample dropped `use_clipped_model_output`.  With clip_sample on, the option
    changes the update (scheduling_ddim.py:336-340): eager and replayed runs must agree WITH it and differ from the run without it"""
    g = _load(golden_dir, "pipeline_tiny.npz")
    sd = W.make_weights(W.unet_state_shapes(Fn.tiny_unet_config()), int(g["unet_weight_seed"]))
    smp = DDIMSampler(UNet3DEngine(pack_unet(sd, tiny_cfg(), torch.float32, DEV)), DDIMConfig(clip_sample=True))
    args = (g["latents"] * 2.0, g["text_embeddings"], 4, 8.0, g["first_image_latents"], g["first_images_mask"])
    eager = smp.sample(*args, fps=[2], flow=[4], use_graph=False, use_clipped_model_output=True).cpu()
    graph = smp.sample(*args, fps=[2], flow=[4], use_graph=True, use_clipped_model_output=True).cpu()
    plain = smp.sample(*args, fps=[2], flow=[4], use_graph=True, use_clipped_model_output=False).cpu()
    assert torch.equal(graph, eager), rel(graph, eager)
    assert not torch.equal(plain, eager)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 6e-2)])
def test_vae_decode_vs_reference_golden(golden_dir, dtype, tol):
    g = _load(golden_dir, "vae_tiny.npz")
    vcfg = VAEDecoderConfig(block_out_channels=(64, 128, 128, 128))
    sdv = W.make_weights(W.vae_decoder_state_shapes(Fn.VAEConfig(block_out_channels=(64, 128, 128, 128))), int(g["weight_seed"]))
    vae = VAEDecoderEngine(pack_vae_decoder(sdv, vcfg, dtype, DEV))
    out = vae.decode(g["z"] * vcfg.scaling_factor).cpu()
    ref = (g["out"] / 2 + 0.5).clamp(0, 1)
    e = (out - ref).abs().max().item()
    report(f"vae {dtype}: max abs err {e:.3e}")
    assert e < tol, e


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 8e-2)])
def test_vae_encode_vs_reference_golden(golden_dir, dtype, tol):
    """first-frame conditioning front-end: moments of AutoencoderKL.encode (64x48 image, asymmetric-pad downsamples)"""
    from followyourclick_amd.engine.vae import VAEEncoderEngine
    from followyourclick_amd.engine.weights import pack_vae_encoder
    g = _load(golden_dir, "vae_enc_tiny.npz")
    vcfg = VAEDecoderConfig(block_out_channels=(64, 128, 128, 128))
    sde = W.make_weights(W.vae_encoder_state_shapes(Fn.VAEConfig(block_out_channels=(64, 128, 128, 128))), int(g["weight_seed"]))
    enc = VAEEncoderEngine(pack_vae_encoder(sde, vcfg, dtype, DEV))
    out = enc.encode_moments(g["x"]).cpu()
    e = (out - g["moments"]).abs().max().item()
    report(f"vae encode {dtype}: max abs err {e:.3e} (moments max {g['moments'].abs().max().item():.2f})")
    assert e < tol, e
