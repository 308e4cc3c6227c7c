"""Host orchestration of the engine (weight packing, layouts, block wiring, time-embedding tables,
cross-attention K/V cache, DDIM coefficients) checked on CPU against the oracle by running the engine
on the op emulator (tests/emu_ops.py).  The HIP kernels themselves are covered by the -m gpu tests."""
import os

import numpy as np
import pytest
import torch

from emu_ops import EmuOps
from followyourclick_amd.engine import DDIMConfig, UNet3DConfig, VAEDecoderConfig
from followyourclick_amd.engine.sampler import DDIMSampler
from followyourclick_amd.engine.scheduler import DDIMTables
from followyourclick_amd.engine.unet3d import UNet3DEngine
from followyourclick_amd.engine.vae import VAEDecoderEngine
from followyourclick_amd.engine.weights import pack_unet, pack_vae_decoder
from oracle import functional as Fn
from oracle import weights as W


def tiny_cfg(**kw):
    return UNet3DConfig(block_out_channels=(64, 128, 256, 256), cross_attention_dim=64, sample_size=8, **kw)


def _load(golden_dir, name):
    return {k: torch.from_numpy(v) if v.shape else v for k, v in np.load(os.path.join(golden_dir, name)).items()}


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 6e-2), (torch.float16, 8e-3)])
def test_unet_forward_matches_golden(golden_dir, dtype, tol):
    g = _load(golden_dir, "unet_tiny_fwd.npz")
    sd = W.make_weights(W.unet_state_shapes(Fn.tiny_unet_config()), int(g["weight_seed"]))
    eng = UNet3DEngine(pack_unet(sd, tiny_cfg(), dtype, "cpu"), ops=EmuOps())
    x9 = g["sample"]                                    # (2, 9, F, h, w): CFG pair of identical inputs
    B, C9, F, H, Wd = x9.shape
    x = torch.zeros(B * F * H * Wd, 64)
    x[:, :C9] = x9.permute(0, 2, 3, 4, 1).reshape(-1, C9)
    eng.prepare_context(g["text"])
    _, temb = eng.prepare_time_embeddings([int(g["timestep"])], g["fps"].tolist(), g["flow"].tolist(), B)
    out = eng.forward(x.to(dtype), temb, B, F, H, Wd)
    out = out.float().reshape(B, F, H, Wd, 4).permute(0, 4, 1, 2, 3)
    ref = g["out"]
    rel = ((out - ref).norm() / ref.norm()).item()
    assert rel < tol, rel


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 6e-2), (torch.float16, 8e-3)])
def test_unet_forward_with_fused_ff_blocks(golden_dir, dtype, tol, monkeypatch):
    """FYC_FUSE_FF: every feed-forward (spatial and temporal transformer blocks) goes through `ff_block` with the weight stream of
    weights.pack_ff_block (projection stages, FF1 chunks + constants, W2' in k-slot order) and its tile statistics feed the next
    GroupNorm - same golden as the unfused schedule"""
    from followyourclick_amd.engine import unet3d
    monkeypatch.setattr(unet3d, "FUSE_FF", True)
    calls = []

    class Spy(EmuOps):
        def ff_block_supported(self, dtype, *, rows, C_, hidden, cs_rows=0):      # the tiny widths are outside the kernel's shapes: force the path
            return rows % 128 == 0 and (cs_rows == 0 or (cs_rows % 128 == 0 and rows % cs_rows == 0))

        def ff_block(self, *a, **kw):
            calls.append((kw["rows"], kw["C_"], kw["chan_parts"] is not None))
            return super().ff_block(*a, **kw)
    g = _load(golden_dir, "unet_tiny_fwd.npz")
    sd = W.make_weights(W.unet_state_shapes(Fn.tiny_unet_config()), int(g["weight_seed"]))
    eng = UNet3DEngine(pack_unet(sd, tiny_cfg(), dtype, "cpu"), ops=Spy())
    x9 = g["sample"]
    B, C9, F, H, Wd = x9.shape
    x = torch.zeros(B * F * H * Wd, 64)
    x[:, :C9] = x9.permute(0, 2, 3, 4, 1).reshape(-1, C9)
    eng.prepare_context(g["text"])
    _, temb = eng.prepare_time_embeddings([int(g["timestep"])], g["fps"].tolist(), g["flow"].tolist(), B)
    out = eng.forward(x.to(dtype), temb, B, F, H, Wd).float().reshape(B, F, H, Wd, 4).permute(0, 4, 1, 2, 3)
    rel = ((out - g["out"]).norm() / g["out"].norm()).item()
    assert rel < tol, rel
    assert len(calls) >= 5, calls                      # the 64-channel level (rows = B*F*H*W is a multiple of 128 there)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 6e-2), (torch.float16, 8e-3)])
def test_unet_forward_with_panel_linears(golden_dir, dtype, tol, monkeypatch):
    """FYC_FUSE_PANEL: proj_in of every transformer / motion module goes through `panel_linear` WITH its GroupNorm (from the
    producer's channel sums: no gn_apply pass), the attention output projections through it with their residual - same golden"""
    from followyourclick_amd.engine import unet3d
    monkeypatch.setattr(unet3d, "FUSE_PANEL", True)
    monkeypatch.setattr(unet3d, "PANEL_ALL", True)
    calls = []

    class Spy(EmuOps):
        def panel_linear_supported(self, dtype, *, rows, N, K, gn_rows_per_sample=0, gn_groups=32):   # the tiny widths are outside the kernel's shapes: force the path
            return N == K

        def panel_linear(self, x, out, **kw):
            calls.append((kw["N"], kw.get("gn_cs") is not None, kw.get("residual") is not None))
            return super().panel_linear(x, out, **kw)
    g = _load(golden_dir, "unet_tiny_fwd.npz")
    sd = W.make_weights(W.unet_state_shapes(Fn.tiny_unet_config()), int(g["weight_seed"]))
    eng = UNet3DEngine(pack_unet(sd, tiny_cfg(), dtype, "cpu"), ops=Spy())
    x9 = g["sample"]
    B, C9, F, H, Wd = x9.shape
    x = torch.zeros(B * F * H * Wd, 64)
    x[:, :C9] = x9.permute(0, 2, 3, 4, 1).reshape(-1, C9)
    eng.prepare_context(g["text"])
    _, temb = eng.prepare_time_embeddings([int(g["timestep"])], g["fps"].tolist(), g["flow"].tolist(), B)
    out = eng.forward(x.to(dtype), temb, B, F, H, Wd).float().reshape(B, F, H, Wd, 4).permute(0, 4, 1, 2, 3)
    rel = ((out - g["out"]).norm() / g["out"].norm()).item()
    assert rel < tol, rel
    # proj_in with its GroupNorm (wherever the producer could deliver channel sums: frames of >= 16 rows); to_out with residual
    assert sum(1 for c in calls if c[1]) >= 8 and sum(1 for c in calls if c[2]) >= 16, calls


def test_panel_linear_stream_round_trip():
    from followyourclick_amd.engine.weights import pack_panel_linear
    for N, K, T in [(320, 320, torch.bfloat16), (640, 640, torch.bfloat16), (320, 640, torch.float32), (128, 64, torch.float32), (640, 320, torch.float16)]:
        w = torch.randn(N, K, generator=torch.Generator().manual_seed(N + K)).to(T)
        st = pack_panel_linear(w)
        assert st.numel() == N * K
        assert torch.equal(EmuOps._panel_unpack(st, N, K), w)


def test_ff_block_stream_round_trip():
    """weights.pack_ff_block against the layout description of include/fyc.h (the emulator's independent unpacker), at the
    kernel's widths and at a small one"""
    from followyourclick_amd.engine.weights import Packed, ff_block_layout, pack_ff_block
    assert ff_block_layout(320, 1280) == (92, 32, 14)
    for C, hid, T in [(320, 1280, torch.bfloat16), (64, 256, torch.float32), (96, 384, torch.bfloat16), (320, 1280, torch.float16)]:
        g = torch.Generator().manual_seed(C)
        ff = Packed(w1=torch.randn(2 * hid, C, generator=g).to(T), b1=torch.randn(2 * hid, generator=g), cs1=torch.randn(2 * hid, generator=g),
                    po_w=torch.randn(C, C + hid, generator=g).to(T), po_b=torch.randn(C, generator=g))
        st = pack_ff_block(ff)
        nst, npc, _ = ff_block_layout(C, hid)
        assert st.numel() == nst * npc * 512
        Wp, W1, bi, W2 = EmuOps._ff_unpack(st, C, hid)
        assert torch.equal(Wp, ff.po_w[:, :C]) and torch.equal(W1, ff.w1) and torch.equal(W2, ff.po_w[:, C:]) and torch.equal(bi, ff.b1)


@pytest.mark.parametrize("with_pe", [True, False])
def test_temporal_block_stream_round_trip(with_pe):
    """weights.pack_temporal_stream against the unpacker written from include/fyc.h's layout description, and the two
    specifications of the sub-block (per-head operands with the LayerNorm folded / packed stream with normalised tokens)
    against each other"""
    from followyourclick_amd.engine.weights import Packed, pack_temporal_block, temporal_block_layout
    T, H, d, F, P, clips = torch.bfloat16, 8, 40, 16, 3, 2
    C = H * d
    g = lambda shape, seed, scale=1.0: torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale     # noqa: E731
    att = Packed(qkv_f=((g((3 * C, C), 1) * C ** -0.5).to(T), g((3 * C,), 2, 0.1), None), pe_w=g((24, 3 * C), 3, 0.5) if with_pe else None,
                 o_w=(g((C, C), 4) * C ** -0.5).to(T), o_b=g((C,), 5, 0.1))
    att["qkv_f"] = (att.qkv_f[0], att.qkv_f[1], att.qkv_f[0].float().sum(dim=1))
    ops = pack_temporal_block(att, H, F)
    lay = temporal_block_layout(C)
    assert (lay["a_pieces"], lay["b_pieces"], lay["stride"]) == (66, 73, 73) and ops["wstream"].numel() == 2 * H * 73 * 512
    w, tab, wo = EmuOps._tblock_unpack(ops["wstream"], C, H, d)
    for i in range(3):
        assert torch.equal(w[:, i, :d], ops["w_qkv"][:, i * d:(i + 1) * d]) and not w[:, i, d:].float().any()
        ref = ops["bias"][:, None, i * d:(i + 1) * d] + (ops["pe_bias"].permute(1, 0, 2)[:, :, i * d:(i + 1) * d] if with_pe else 0.0)
        assert torch.equal(tab[:, :, i, :d], ref.expand(H, F, d)) and not tab[:, :, i, d:].any()
    assert torch.equal(wo, ops["w_out"])
    x = (g((clips * F * P, C), 6) * 1.5 + 0.3).to(T)
    kw = dict(clips=clips, frames=F, pixels=P, heads=H, d=d, scale=d ** -0.5)
    emu = EmuOps(acc=torch.float64)
    o_a, o_b = torch.zeros_like(x), torch.zeros_like(x)
    emu.temporal_block(x, o_a, **ops, **kw)
    emu.temporal_block(x, o_b, **{k: v for k, v in ops.items() if k != "wstream"}, **kw)
    rel = ((o_a.double() - o_b.double()).norm() / o_b.double().norm()).item()
    assert rel < 4e-3, rel


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 6e-2), (torch.float16, 8e-3)])
def test_unet_forward_with_fused_temporal_blocks(golden_dir, dtype, tol, monkeypatch):
    """FYC_FUSE_TEMPORAL: every temporal attention sub-block goes through `temporal_block` with the per-head operands of
    weights.pack_temporal_block (q|k|v gather per head, positional bias per frame, output projection slices) - same golden"""
    from followyourclick_amd.engine import unet3d
    monkeypatch.setattr(unet3d, "FUSE_TEMPORAL", True)
    calls = []

    class Spy(EmuOps):
        def temporal_block_supported(self, dtype, **kw):      # the tiny widths are outside the kernel's shapes: force the path
            return True

        def temporal_block(self, *a, **kw):
            calls.append(kw["pixels"])
            return super().temporal_block(*a, **kw)

        def temporal_attention(self, *a, **kw):
            raise AssertionError("the unfused temporal attention must not run")
    g = _load(golden_dir, "unet_tiny_fwd.npz")
    sd = W.make_weights(W.unet_state_shapes(Fn.tiny_unet_config()), int(g["weight_seed"]))
    eng = UNet3DEngine(pack_unet(sd, tiny_cfg(), dtype, "cpu"), ops=Spy())
    x9 = g["sample"]
    B, C9, F, H, Wd = x9.shape
    x = torch.zeros(B * F * H * Wd, 64)
    x[:, :C9] = x9.permute(0, 2, 3, 4, 1).reshape(-1, C9)
    eng.prepare_context(g["text"])
    _, temb = eng.prepare_time_embeddings([int(g["timestep"])], g["fps"].tolist(), g["flow"].tolist(), B)
    out = eng.forward(x.to(dtype), temb, B, F, H, Wd).float().reshape(B, F, H, Wd, 4).permute(0, 4, 1, 2, 3)
    rel = ((out - g["out"]).norm() / g["out"].norm()).item()
    assert rel < tol, rel
    assert len(calls) >= 2 * 9                         # two attention blocks per motion module


def test_unet_forward_odd_size(golden_dir):
    g = _load(golden_dir, "unet_tiny_odd_fwd.npz")
    sd = W.make_weights(W.unet_state_shapes(Fn.tiny_unet_config()), int(g["weight_seed"]))
    eng = UNet3DEngine(pack_unet(sd, tiny_cfg(), torch.float32, "cpu"), ops=EmuOps())
    B, C9, F, H, Wd = g["sample"].shape
    x = torch.zeros(B * F * H * Wd, 64)
    x[:, :C9] = g["sample"].permute(0, 2, 3, 4, 1).reshape(-1, C9)
    eng.prepare_context(g["text"])
    _, temb = eng.prepare_time_embeddings([int(g["timestep"])], g["fps"].tolist(), g["flow"].tolist(), B)
    out = eng.forward(x, temb, B, F, H, Wd).reshape(B, F, H, Wd, 4).permute(0, 4, 1, 2, 3)
    assert ((out - g["out"]).norm() / g["out"].norm()).item() < 2e-4


def test_unet_forward_ip_matches_oracle():
    """IP-Adapter decoupled cross-attention with the deployed (xformers) softmax temperature."""
    ocfg = Fn.tiny_unet_config(use_ip_cross_attention=True, ip_scale=0.7)
    sd = W.make_weights(W.unet_state_shapes(ocfg), 0)
    inp = W.seeded_inputs(ocfg, 1, 4, 8, 8, seed=7)
    x9 = torch.cat([Fn.build_model_input(inp["latents"], inp["first_image_latents"], inp["first_images_mask"])] * 2)
    fps, flow = torch.tensor([2, 2]), torch.tensor([4, 4])
    with torch.no_grad():
        ref = Fn.unet3d_forward(sd, ocfg, x9, torch.tensor(961), inp["text"], fps, flow, inp["ip_tokens"])
    eng = UNet3DEngine(pack_unet(sd, tiny_cfg(use_ip_cross_attention=True, ip_scale=0.7), torch.float32, "cpu"), ops=EmuOps())
    B, C9, F, H, Wd = x9.shape
    x = torch.zeros(B * F * H * Wd, 64)
    x[:, :C9] = x9.permute(0, 2, 3, 4, 1).reshape(-1, C9)
    eng.prepare_context(inp["text"], inp["ip_tokens"])
    _, temb = eng.prepare_time_embeddings([961], [2, 2], [4, 4], B)
    out = eng.forward(x, temb, B, F, H, Wd).reshape(B, F, H, Wd, 4).permute(0, 4, 1, 2, 3)
    assert ((out - ref).norm() / ref.norm()).item() < 2e-4


def test_ddim_tables_bit_exact(golden_dir):
    g = _load(golden_dir, "ddim.npz")
    tb = DDIMTables(DDIMConfig())
    assert torch.equal(tb.alphas_cumprod, g["alphas_cumprod"])
    for n in (5, 25, 50):
        assert torch.equal(tb.timesteps(n), g[f"timesteps_{n}"])
    c = tb.step_coefficients(1, 25)
    assert c[2] == 1.0 and c[3] == 0.0          # last step returns x0 (final_alpha_cumprod = 1)


def test_sampler_trajectory_matches_reference_pipeline(golden_dir):
    """5 DDIM steps, CFG 8, mask + first-frame concat, fps/flow conditioning (the cfg1-shaped run)."""
    g = _load(golden_dir, "pipeline_tiny.npz")
    sd = W.make_weights(W.unet_state_shapes(Fn.tiny_unet_config()), int(g["unet_weight_seed"]))
    eng = UNet3DEngine(pack_unet(sd, tiny_cfg(), torch.float32, "cpu"), ops=EmuOps())
    traj = []
    lat = DDIMSampler(eng, DDIMConfig()).sample(g["latents"], g["text_embeddings"], 5, 8.0, g["first_image_latents"],
                                                g["first_images_mask"], fps=[2], flow=[4],
                                                callback=lambda i, t, l: traj.append(l.clone()))
    traj = torch.stack(traj)
    err = (traj - g["trajectory"]).flatten(1).norm(dim=1) / g["trajectory"].flatten(1).norm(dim=1)
    assert err.max().item() < 5e-4, err
    vcfg = VAEDecoderConfig(block_out_channels=(64, 128, 128, 128))
    sdv = W.make_weights(W.vae_decoder_state_shapes(Fn.VAEConfig(block_out_channels=(64, 128, 128, 128))), int(g["vae_weight_seed"]))
    vae = VAEDecoderEngine(pack_vae_decoder(sdv, vcfg, torch.float32, "cpu"), ops=EmuOps())
    vid = vae.decode_video(lat)
    assert vid.shape == g["videos"].shape
    assert (vid - g["videos"]).abs().max().item() < 2e-3


@pytest.mark.parametrize("dtype,tol_traj,tol_video", [(torch.bfloat16, 7e-2, 1.5e-1), (torch.float16, 9e-3, 2e-2)])
def test_sampler_and_vae_in_16_bit_modes(golden_dir, dtype, tol_traj, tol_video):
    """the same 5-step run + decode with bf16 / f16 storage on the op emulator: the host side of the 16-bit modes (packing, the sampler's
    f32 latents against 16-bit UNet I/O, the VAE engine) and the ~8x between the two formats (measured: bf16 1.4e-2 ... 5.1e-2 per step,
    video 1.0e-1; f16 1.7e-3 ... 6.0e-3, video 1.3e-2)"""
    g = _load(golden_dir, "pipeline_tiny.npz")
    sd = W.make_weights(W.unet_state_shapes(Fn.tiny_unet_config()), int(g["unet_weight_seed"]))
    eng = UNet3DEngine(pack_unet(sd, tiny_cfg(), dtype, "cpu"), ops=EmuOps())
    traj = []
    lat = DDIMSampler(eng, DDIMConfig()).sample(g["latents"], g["text_embeddings"], 5, 8.0, g["first_image_latents"],
                                                g["first_images_mask"], fps=[2], flow=[4],
                                                callback=lambda i, t, l: traj.append(l.clone()))
    assert lat.dtype == torch.float32
    traj = torch.stack(traj)
    err = (traj - g["trajectory"]).flatten(1).norm(dim=1) / g["trajectory"].flatten(1).norm(dim=1)
    assert err.max().item() < tol_traj, err
    vcfg = VAEDecoderConfig(block_out_channels=(64, 128, 128, 128))
    sdv = W.make_weights(W.vae_decoder_state_shapes(Fn.VAEConfig(block_out_channels=(64, 128, 128, 128))), int(g["vae_weight_seed"]))
    vae = VAEDecoderEngine(pack_vae_decoder(sdv, vcfg, dtype, "cpu"), ops=EmuOps())
    assert (vae.decode_video(lat) - g["videos"]).abs().max().item() < tol_video


def test_vae_decode_matches_golden(golden_dir):
    g = _load(golden_dir, "vae_tiny.npz")
    vcfg = VAEDecoderConfig(block_out_channels=(64, 128, 128, 128))
    sdv = W.make_weights(W.vae_decoder_state_shapes(Fn.VAEConfig(block_out_channels=(64, 128, 128, 128))), int(g["weight_seed"]))
    vae = VAEDecoderEngine(pack_vae_decoder(sdv, vcfg, torch.float32, "cpu"), ops=EmuOps())
    out = vae.decode(g["z"] * vcfg.scaling_factor)
    ref = (g["out"] / 2 + 0.5).clamp(0, 1)
    assert (out - ref).abs().max().item() < 1e-4


def test_vae_encode_matches_golden(golden_dir):
    from followyourclick_amd.engine.schema import vae_encoder_schema
    from followyourclick_amd.engine.vae import VAEEncoderEngine
    from followyourclick_amd.engine.weights import pack_vae_encoder
    g = _load(golden_dir, "vae_enc_tiny.npz")
    vcfg = VAEDecoderConfig(block_out_channels=(64, 128, 128, 128))
    sde = W.make_weights(W.vae_encoder_state_shapes(Fn.VAEConfig(block_out_channels=(64, 128, 128, 128))), int(g["weight_seed"]))
    assert set(vae_encoder_schema(vcfg)) == set(sde)
    enc = VAEEncoderEngine(pack_vae_encoder(sde, vcfg, torch.float32, "cpu"), ops=EmuOps())
    assert (enc.encode_moments(g["x"]) - g["moments"]).abs().max().item() < 1e-4


def test_two_transformer_blocks_per_motion_module():
    """num_transformer_block=2 (VanillaTemporalModule default): inner block keeps a plain FF residual"""
    ocfg = Fn.tiny_unet_config(motion_num_transformer_block=2)
    sd = W.make_weights(W.unet_state_shapes(ocfg), 2)
    inp = W.seeded_inputs(ocfg, 1, 3, 8, 8, seed=5)
    x9 = torch.cat([Fn.build_model_input(inp["latents"], inp["first_image_latents"], inp["first_images_mask"])] * 2)
    with torch.no_grad():
        ref = Fn.unet3d_forward(sd, ocfg, x9, torch.tensor(321), inp["text"], torch.tensor([2, 2]), torch.tensor([4, 4]))
    eng = UNet3DEngine(pack_unet(sd, tiny_cfg(motion_num_transformer_block=2), torch.float32, "cpu"), ops=EmuOps())
    B, C9, F, H, Wd = x9.shape
    x = torch.zeros(B * F * H * Wd, 64)
    x[:, :C9] = x9.permute(0, 2, 3, 4, 1).reshape(-1, C9)
    eng.prepare_context(inp["text"])
    _, temb = eng.prepare_time_embeddings([321], [2, 2], [4, 4], B)
    out = eng.forward(x, temb, B, F, H, Wd).reshape(B, F, H, Wd, 4).permute(0, 4, 1, 2, 3)
    assert ((out - ref).norm() / ref.norm()).item() < 2e-4


# ---- 2-D Stable Diffusion first-image path -------------------------------------------------------------
def cfg_2d():
    return tiny_cfg(use_motion_module=False, use_fps_condition=False, use_first_frame_mask_condition_concat=False)


def oracle_cfg_2d():
    return Fn.tiny_unet_config(use_motion_module=False, use_fps_condition=False, use_first_frame_mask_condition_concat=False)


SCHED_2D = dict(beta_schedule="scaled_linear", set_alpha_to_one=False, prediction_type="epsilon", rescale_betas_zero_snr=False)


def test_unet2d_forward_matches_reference(golden_dir):
    """UNet2DConditionModel.forward of the real reference == the engine run as a one-frame clip without motion modules."""
    g = _load(golden_dir, "sd2d_unet_fwd.npz")
    sd = W.make_weights(W.unet_state_shapes(oracle_cfg_2d()), int(g["weight_seed"]))
    eng = UNet3DEngine(pack_unet(sd, cfg_2d(), torch.float32, "cpu"), ops=EmuOps())
    eng.prepare_context(g["text"])
    for s, t, o in (("sample", "timestep", "out"), ("sample_odd", "timestep_odd", "out_odd")):
        B, C, H, Wd = g[s].shape
        x = torch.zeros(B * H * Wd, 64)
        x[:, :C] = g[s].permute(0, 2, 3, 1).reshape(-1, C)
        _, temb = eng.prepare_time_embeddings([int(g[t])], None, None, B)
        out = eng.forward(x, temb, B, 1, H, Wd).reshape(B, H, Wd, 4).permute(0, 3, 1, 2)
        assert ((out - g[o]).norm() / g[o].norm()).item() < 2e-4


def test_sampler_plain_latents_matches_reference_sd_pipeline(golden_dir):
    """StableDiffusionPipeline.__call__ of the real reference: 4 DDIM steps, epsilon prediction, scaled_linear betas."""
    g = _load(golden_dir, "sd2d_pipeline.npz")
    sd = W.make_weights(W.unet_state_shapes(oracle_cfg_2d()), int(g["unet_weight_seed"]))
    eng = UNet3DEngine(pack_unet(sd, cfg_2d(), torch.float32, "cpu"), ops=EmuOps())
    traj = []
    lat = DDIMSampler(eng, DDIMConfig(**SCHED_2D)).sample(g["latents"][:, :, None], g["text_embeddings"], 4, 8.0,
                                                          callback=lambda i, t, l: traj.append(l[:, :, 0].clone()))
    traj = torch.stack(traj)
    err = (traj - g["trajectory"]).flatten(1).norm(dim=1) / g["trajectory"].flatten(1).norm(dim=1)
    assert err.max().item() < 5e-4, err
    vcfg = VAEDecoderConfig(block_out_channels=(64, 128, 128, 128))
    sdv = W.make_weights(W.vae_decoder_state_shapes(Fn.VAEConfig(block_out_channels=(64, 128, 128, 128))), int(g["vae_weight_seed"]))
    vae = VAEDecoderEngine(pack_vae_decoder(sdv, vcfg, torch.float32, "cpu"), ops=EmuOps())
    img = vae.decode_video(lat)[:, :, 0].permute(0, 2, 3, 1)
    assert (img - g["images"]).abs().max().item() < 2e-3


def test_sampler_plain_text_to_video_matches_reference_pipeline(golden_dir):
    g = _load(golden_dir, "pipeline_tiny_t2v.npz")
    ocfg = Fn.tiny_unet_config(use_fps_condition=False, use_first_frame_mask_condition_concat=False)
    sd = W.make_weights(W.unet_state_shapes(ocfg), int(g["unet_weight_seed"]))
    eng = UNet3DEngine(pack_unet(sd, tiny_cfg(use_fps_condition=False, use_first_frame_mask_condition_concat=False), torch.float32, "cpu"),
                       ops=EmuOps())
    traj = []
    DDIMSampler(eng, DDIMConfig(prediction_type="epsilon", rescale_betas_zero_snr=False)).sample(
        g["latents"], g["text_embeddings"], 4, 7.5, callback=lambda i, t, l: traj.append(l.clone()))
    traj = torch.stack(traj)
    err = (traj - g["trajectory"]).flatten(1).norm(dim=1) / g["trajectory"].flatten(1).norm(dim=1)
    assert err.max().item() < 5e-4, err


def test_sampler_video_scale_matches_reference_pipeline(golden_dir):
    g = _load(golden_dir, "pipeline_tiny_t2v.npz")
    ocfg = Fn.tiny_unet_config(use_fps_condition=False, use_first_frame_mask_condition_concat=False)
    sd = W.make_weights(W.unet_state_shapes(ocfg), int(g["unet_weight_seed"]))
    eng = UNet3DEngine(pack_unet(sd, tiny_cfg(use_fps_condition=False, use_first_frame_mask_condition_concat=False), torch.float32, "cpu"),
                       ops=EmuOps())
    traj = []
    DDIMSampler(eng, DDIMConfig(prediction_type="epsilon", rescale_betas_zero_snr=False)).sample(
        g["latents"], g["text_embeddings"], 3, 7.5, callback=lambda i, t, l: traj.append(l.clone()), video_scale=float(g["video_scale"]))
    traj = torch.stack(traj)
    ref = g["trajectory_video_scale"]
    err = (traj - ref).flatten(1).norm(dim=1) / ref.flatten(1).norm(dim=1)
    assert err.max().item() < 5e-4, err


def test_sampler_first_frame_condition_matches_reference_pipeline(golden_dir):
    g = _load(golden_dir, "pipeline_tiny_t2v.npz")
    ocfg = Fn.tiny_unet_config(use_fps_condition=False, use_first_frame_mask_condition_concat=False)
    sd = W.make_weights(W.unet_state_shapes(ocfg), int(g["unet_weight_seed"]))
    eng = UNet3DEngine(pack_unet(sd, tiny_cfg(use_fps_condition=False, use_first_frame_mask_condition_concat=False), torch.float32, "cpu"),
                       ops=EmuOps())
    traj = []
    DDIMSampler(eng, DDIMConfig(prediction_type="epsilon", rescale_betas_zero_snr=False)).sample(
        g["latents"], g["text_embeddings"], 3, 7.5, first_image_latents=g["first_image_latents"],
        callback=lambda i, t, l: traj.append(l.clone()), first_frame_condition=True)
    traj, ref = torch.stack(traj), g["trajectory_first_frame"]
    err = (traj - ref).flatten(1).norm(dim=1) / ref.flatten(1).norm(dim=1)
    assert err.max().item() < 5e-4, err


def test_ip_model_without_ip_tokens_splits_the_text_states():
    """On a model built with IP cross-attention the reference always treats the LAST num_tokens tokens of
    encoder_hidden_states as image tokens (animatediff/models/attention.py:52-53, 104-120), also when the caller passed plain
    text states: prepare_context(ctx) must equal prepare_context(ctx[:, :-n], ctx[:, -n:])."""
    cfg = tiny_cfg(use_ip_cross_attention=True, ip_scale=0.7)
    ocfg = Fn.tiny_unet_config(use_ip_cross_attention=True, ip_scale=0.7)
    sd = W.make_weights(W.unet_state_shapes(ocfg), 0)
    eng = UNet3DEngine(pack_unet(sd, cfg, torch.float32, "cpu"), ops=EmuOps())
    ctx = torch.randn(2, 77, 64, generator=torch.Generator().manual_seed(1))
    n = cfg.ip_num_tokens
    eng.prepare_context(ctx)
    a = eng.ctx_cache
    eng.prepare_context(ctx[:, :-n], ctx[:, -n:])
    b = eng.ctx_cache
    assert a[0]["n_text"] == 77 - n and a[0]["n_ip"] == n
    for ea, eb in zip(a, b):
        for ta, tb in zip(ea["text"][:2] + ea["ip"][:2], eb["text"][:2] + eb["ip"][:2]):
            assert torch.equal(ta, tb)
