"""Parity where the benchmark runs: SD-1.5 WIDTHS (320/640/1280/1280, ctx 768, head dims 40/80/160) and the production
precision, against golden vectors from the REAL reference (oracle/make_golden_full.py, container only):

  unet_full_small_fwd.npz   one UNet3D forward, F=4, 16x16 latent
  cfg1_trajectory.npz       BASELINE configs[0]: AnimationPipeline.__call__, 8 frames 256x256, 5 DDIM steps, every step's latents
  cfg2_trajectory.npz       BASELINE configs[1]: 16 frames 512x512, 25 DDIM steps, latents after steps 0 / 4 / 24
  vae_full.npz              AutoencoderKL.decode at (128, 256, 512, 512)
  cfg4_forwards.npz         BASELINE configs[3] paths: one F = 32 forward (24x24 latent) and one 96x96-latent forward, max_len 32
  cfg5_trajectory.npz       BASELINE configs[4]: IP-Adapter + rectangle mask, 5 DDIM steps (reference + oracle without the CPU-path quirk)

Each UNet golden holds the reference twice: as the CPU runs it (f32) and under the CUDA-autocast cast policy in bfloat16
(oracle/autocast_emul.py) = the precision the engine's production mode computes in.  Tiered tolerance:

  * f32 parity mode         rel-L2 <= 1e-3 vs the f32 reference (BASELINE.json north_star's figure; measured ~1e-5);
  * bf16 production mode    two different bf16 roundings of a random-weight UNet are NOT within 1e-3 of each other - the
                            reference's own bf16 run is `drift` (1.2e-2 per forward, 4.3e-2 after 5 steps) away from its
                            own f32 run.  The engine's bf16 mode must be no further from the f32 reference than
                            BF16_FACTOR (1.1) x that drift - it is measured at 0.88-0.95 x, i.e. CLOSER to the f32 reference than
                            the reference's own bf16 run - and no further from the bf16 reference than BF16_VS_BF16 (1.5) x drift
                            (measured 1.21-1.28 x); both numbers are written to gpurun_out/parity_report.txt.
"""
import os

import pytest
import torch

from followyourclick_amd.engine import DDIMConfig, UNet3DConfig, VAEDecoderConfig
from followyourclick_amd.engine.sampler import DDIMSampler
from followyourclick_amd.engine.unet3d import UNet3DEngine
from followyourclick_amd.engine.vae import VAEDecoderEngine
from followyourclick_amd.engine.weights import pack_unet, pack_vae_decoder
from oracle import functional as Fn
from oracle import weights as W

from test_engine_gpu import _load, _nhwc, rel, report

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# engine-bf16 vs reference-f32 may be at most BF16_FACTOR x drift (drift = reference-bf16-autocast vs reference-f32); measured over
# every checkpoint below: 0.88 .. 0.95 x (gpurun_out/parity_report.txt), so 1.1 leaves ~15 %: a 30 % loss of accuracy fails.
# engine-bf16 vs reference-bf16 (two independent bf16 roundings): measured 1.21 .. 1.28 x drift, bound BF16_VS_BF16 = 1.5.
BF16_FACTOR = 1.1
BF16_VS_BF16 = 1.5


@pytest.fixture(scope="module")
def engines(fullwidth):
    """the precisions of the full-width engine, packed once per SESSION (tests/conftest.py::fullwidth: 2.6 GB bf16 / f16, 5.2 GB f32)"""
    def get(dtype):
        return fullwidth.engine(Fn.UNetConfig(), UNet3DConfig(), dtype, seed=0)
    return get


IP_OCFG = dict(use_ip_cross_attention=True, ip_num_tokens=16, ip_scale=0.7)


def test_full_width_forward_vs_reference(golden_dir, engines):
    g = _load(golden_dir, "unet_full_small_fwd.npz")
    cfg = Fn.UNetConfig()
    F, H, Wd = int(g["frames"]), int(g["h"]), int(g["w"])
    inp = W.seeded_inputs(cfg, 1, F, H, Wd, seed=int(g["input_seed"]))
    x9 = torch.cat([Fn.build_model_input(inp["latents"], inp["first_image_latents"], inp["first_images_mask"])] * 2)
    drift = float(g["drift"])
    for dtype in (torch.float32, torch.bfloat16):
        eng = engines(dtype)
        eng.prepare_context(inp["text"])
        _, temb = eng.prepare_time_embeddings([int(g["timestep"])], g["fps"].tolist(), g["flow"].tolist(), 2)
        out = eng.forward(_nhwc(x9, dtype), temb, 2, F, H, Wd).float().cpu().reshape(2, F, H, Wd, 4).permute(0, 4, 1, 2, 3)
        assert torch.isfinite(out).all()
        r32, r16 = rel(out, g["out_f32"]), rel(out, g["out_bf16"])
        report(f"full-width fwd (F=4, 16x16) {dtype}: vs ref-f32 {r32:.3e}, vs ref-bf16-autocast {r16:.3e} (ref-bf16 vs ref-f32 {drift:.3e})")
        if dtype == torch.float32:
            assert r32 < 1e-3, r32
        else:
            if not r32 < BF16_FACTOR * drift:
                # round 6: this bound failed once (2.36e-2, a library built from a half-edited source) inside a full-suite run and could not be
                # reproduced alone - a failure now says what kind it is: a second forward of the same engine, and a freshly packed engine
                # from the same weights, on the same inputs
                out2 = eng.forward(_nhwc(x9, dtype), temb, 2, F, H, Wd).float().cpu().reshape(2, F, H, Wd, 4).permute(0, 4, 1, 2, 3)
                from followyourclick_amd import ops as _ops
                zero_ok = not bool(_ops.get()._zero.any())
                fresh = UNet3DEngine(pack_unet(W.make_weights(W.unet_state_shapes(cfg), 0), UNet3DConfig(), dtype, DEV))
                fresh.prepare_context(inp["text"])
                _, temb3 = fresh.prepare_time_embeddings([int(g["timestep"])], g["fps"].tolist(), g["flow"].tolist(), 2)
                out3 = fresh.forward(_nhwc(x9, dtype), temb3, 2, F, H, Wd).float().cpu().reshape(2, F, H, Wd, 4).permute(0, 4, 1, 2, 3)
                report(f"full-width fwd bf16 FAILED its bound: second forward of the same engine {rel(out2, g['out_f32']):.3e} (identical to the first: "
                       f"{torch.equal(out, out2)}), freshly packed engine {rel(out3, g['out_f32']):.3e}, zero page intact: {zero_ok}")
            assert r32 < BF16_FACTOR * drift, (r32, drift)
            assert r16 < BF16_VS_BF16 * drift, (r16, drift)


def test_full_width_ip_adapter_forward_vs_oracle(golden_dir, fullwidth):
    """BASELINE configs[4] (16 IP tokens) at SD-1.5 widths.  The reference's CPU path runs attn2 at the IP weight as softmax
    temperature (SURVEY headline 6), the deployed xformers path does not: tests/test_oracle_golden.py pins the oracle WITH that
    quirk to the real reference's output (unet_full_ip_fwd.npz); here the engine (deployed semantics) is held to the same oracle
    without it.  bf16 bound: 1.5 x the reference's own bf16-autocast drift on this forward."""
    g = _load(golden_dir, "unet_full_ip_fwd.npz")
    ocfg = Fn.UNetConfig(use_ip_cross_attention=True, ip_num_tokens=int(g["ip_num_tokens"]), ip_scale=float(g["ip_scale"]))
    assert (ocfg.ip_num_tokens, ocfg.ip_scale) == (IP_OCFG["ip_num_tokens"], IP_OCFG["ip_scale"])
    F, H, Wd = int(g["frames"]), int(g["h"]), int(g["w"])
    inp = W.seeded_inputs(ocfg, 1, F, H, Wd, seed=int(g["input_seed"]))
    x9 = torch.cat([Fn.build_model_input(inp["latents"], inp["first_image_latents"], inp["first_images_mask"])] * 2)
    # the oracle WITHOUT the quirk on these inputs, stored by the golden's recipe (oracle/make_golden_full.py ip) next to the proof that
    # the oracle WITH it reproduces the real reference (round 5 ran that 15-s CPU forward inside this test)
    assert float(g["oracle_quirk_vs_ref"]) < 1e-4
    ref = g["out_oracle_noquirk"].float()
    drift = float(g["drift"])
    for dtype in (torch.float32, torch.bfloat16):
        eng = fullwidth.engine(ocfg, UNet3DConfig(use_ip_cross_attention=True, ip_num_tokens=ocfg.ip_num_tokens, ip_scale=ocfg.ip_scale), dtype, seed=int(g["weight_seed"]))
        eng.prepare_context(inp["text"], inp["ip_tokens"])
        _, temb = eng.prepare_time_embeddings([int(g["timestep"])], g["fps"].tolist(), g["flow"].tolist(), 2)
        out = eng.forward(_nhwc(x9, dtype), temb, 2, F, H, Wd).float().cpu().reshape(2, F, H, Wd, 4).permute(0, 4, 1, 2, 3)
        r = rel(out, ref)
        report(f"full-width IP fwd (16 tokens) {dtype}: vs oracle-f32 {r:.3e} (reference bf16-autocast drift on its CPU path {drift:.3e})")
        assert torch.isfinite(out).all()
        assert r < (1e-3 if dtype == torch.float32 else 2.1e-2), r      # bf16 measured 1.77e-2 (the drift of the CPU-quirk path, 3.3e-2, is no yardstick here)


def _trajectory(golden_dir, engines, name, dtype, num_steps_run):
    g = _load(golden_dir, name)
    cfg = Fn.UNetConfig()
    F, lat, steps = int(g["frames"]), int(g["lat"]), int(g["steps"])
    inp = W.seeded_inputs(cfg, 1, F, lat, lat, seed=int(g["input_seed"]))
    got = {}

    class Stop(Exception):
        pass

    def cb(i, t, l):
        got[i] = l.clone().cpu()
        if i + 1 >= num_steps_run:
            raise Stop

    smp = DDIMSampler(engines(dtype), DDIMConfig())
    try:
        smp.sample(inp["latents"], g["text_embeddings"], steps, 8.0, inp["first_image_latents"], inp["first_images_mask"],
                   fps=[2], flow=[4], callback=cb)
    except Stop:
        pass
    torch.cuda.synchronize()
    return g, got


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_cfg1_trajectory_vs_reference_pipeline(golden_dir, engines, dtype):
    """BASELINE configs[0] (8 frames 256x256, 5 DDIM steps) at full widths: per-step latents of the real AnimationPipeline"""
    g, got = _trajectory(golden_dir, engines, "cfg1_trajectory.npz", dtype, 5)
    for i in sorted(got):
        r32, r16, drift = rel(got[i], g[f"step{i}_f32"]), rel(got[i], g[f"step{i}_bf16"]), float(g[f"drift{i}"])
        report(f"cfg1 step {i} {dtype}: vs ref-f32 {r32:.3e}, vs ref-bf16-autocast {r16:.3e} (ref-bf16 vs ref-f32 {drift:.3e})")
        if dtype == torch.float32:
            assert r32 < 1e-3, (i, r32)
        else:
            assert r32 < BF16_FACTOR * drift, (i, r32, drift)
            assert r16 < BF16_VS_BF16 * drift, (i, r16, drift)


@pytest.mark.parametrize("dtype,run_steps", [(torch.float32, 25), (torch.bfloat16, 25)])
def test_cfg2_trajectory_vs_reference_pipeline(golden_dir, engines, dtype, run_steps):
    """BASELINE configs[1] = the benchmarked workload (16 frames 512x512, 25 DDIM steps): latents after steps 0, 4 and 24 - the
    whole benchmarked trajectory in BOTH precisions (round 2 stopped the f32 parity mode, the one held to north_star's 1e-3,
    after step 4)"""
    if not os.path.exists(os.path.join(golden_dir, "cfg2_trajectory.npz")):
        pytest.skip("cfg2_trajectory.npz not generated (oracle/make_golden_full.py cfg2, ~1.5 h of CPU)")
    g, got = _trajectory(golden_dir, engines, "cfg2_trajectory.npz", dtype, run_steps)
    checked = 0
    for i in (0, 4, 24):
        if i not in got or f"step{i}_f32" not in g:
            continue
        r32, r16, drift = rel(got[i], g[f"step{i}_f32"]), rel(got[i], g[f"step{i}_bf16"]), float(g[f"drift{i}"])
        report(f"cfg2 step {i} {dtype}: vs ref-f32 {r32:.3e}, vs ref-bf16-autocast {r16:.3e} (ref-bf16 vs ref-f32 {drift:.3e})")
        checked += 1
        if dtype == torch.float32:
            assert r32 < 1e-3, (i, r32)
        else:
            assert r32 < BF16_FACTOR * drift, (i, r32, drift)
            assert r16 < BF16_VS_BF16 * drift, (i, r16, drift)
    assert checked == 3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_cfg2_shape_sampling_is_bitwise_repeatable(golden_dir, engines, dtype):
    """The product path is run-to-run deterministic: two eager runs and a hipGraph-replayed run of the sampling loop at the
    benchmarked shape (16 frames, 64x64 latents, full widths; 3 DDIM steps) give the same BITS.  Round 3 had LDS / device float
    atomics in the GroupNorm statistics (fyc_gn_stats, the fused column sums of fyc_gemm), whose arrival order moved the last bits
    of every normalisation and, amplified over the steps, the latents (bounded then by 1.5e-1, now by equality)."""
    g = _load(golden_dir, "cfg1_trajectory.npz")
    cfg = Fn.UNetConfig()
    inp = W.seeded_inputs(cfg, 1, 16, 64, 64, seed=int(g["input_seed"]))
    smp = DDIMSampler(engines(dtype), DDIMConfig())

    def run(use_graph):
        steps = []
        out = smp.sample(inp["latents"], g["text_embeddings"], 3, 8.0, inp["first_image_latents"], inp["first_images_mask"],
                         fps=[2], flow=[4], use_graph=use_graph, callback=lambda i, t, l: steps.append(l.clone()))
        torch.cuda.synchronize()
        return [s.cpu() for s in steps] + [out.cpu()]

    a, b, c = run(False), run(False), run(True)
    for i, (x, y, z) in enumerate(zip(a, b, c)):
        assert torch.isfinite(x).all()
        assert torch.equal(x, y), f"eager vs eager differ at checkpoint {i}: rel {rel(x, y):.3e}"
        assert torch.equal(x, z), f"eager vs hipGraph replay differ at checkpoint {i}: rel {rel(x, z):.3e}"
    report(f"cfg2-shape sampling {dtype}: eager == eager == hipGraph replay, bitwise, over 3 DDIM steps")


@pytest.mark.parametrize("tag", ["f32x24", "f2x96"])
def test_cfg4_forwards_vs_reference(golden_dir, fullwidth, tag):
    """BASELINE configs[3] (32 frames 768x768, temporal_position_encoding_max_len = 32): the two paths it adds to the benchmarked
    config, each as one forward of the REAL reference built with max_len 32 (oracle/make_golden_full.py cfg4) -
    f32x24: F = 32 -> 32 x 32-score temporal attention at d = 40 / 80 / 160 and a 32-row positional table (motion_module.py:286-304,
    371-464; F = 32 is outside the fused temporal block's shapes: the unfused kernels run), 24x24 latent;
    f2x96: a 96x96 latent -> spatial attention over N = 9216 keys (d = 40) and 2304 / 576 / 144 at the deeper levels."""
    g = _load(golden_dir, "cfg4_forwards.npz")
    ocfg = Fn.UNetConfig(temporal_position_encoding_max_len=int(g["max_len"]))
    F, lat = int(g[f"{tag}_frames"]), int(g[f"{tag}_lat"])
    inp = W.seeded_inputs(ocfg, 1, F, lat, lat, seed=int(g[f"{tag}_input_seed"]))
    x9 = torch.cat([Fn.build_model_input(inp["latents"], inp["first_image_latents"], inp["first_images_mask"])] * 2)
    drift = float(g[f"{tag}_drift"])
    for dtype in (torch.float32, torch.bfloat16):
        eng = fullwidth.engine(ocfg, UNet3DConfig(temporal_position_encoding_max_len=int(g["max_len"])), dtype, seed=0)
        eng.prepare_context(inp["text"])
        _, temb = eng.prepare_time_embeddings([int(g["timestep"])], g["fps"].tolist(), g["flow"].tolist(), 2)
        out = eng.forward(_nhwc(x9, dtype), temb, 2, F, lat, lat).float().cpu().reshape(2, F, lat, lat, 4).permute(0, 4, 1, 2, 3)
        assert torch.isfinite(out).all()
        r32, r16 = rel(out, g[f"{tag}_out_f32"]), rel(out, g[f"{tag}_out_bf16"])
        report(f"cfg4 {tag} (F={F}, {lat}x{lat} latent, max_len 32) {dtype}: vs ref-f32 {r32:.3e}, vs ref-bf16-autocast {r16:.3e} (ref-bf16 vs ref-f32 {drift:.3e})")
        if dtype == torch.float32:
            assert r32 < 1e-3, r32
        else:
            assert r32 < BF16_FACTOR * drift, (r32, drift)
            assert r16 < BF16_VS_BF16 * drift, (r16, drift)
        del eng
        torch.cuda.empty_cache()


def test_cfg4_full_shape_forward_vs_reference(golden_dir, fullwidth):
    """BASELINE configs[3] at the shape the bench line runs (VERDICT r3 missing 2): ONE forward of the real reference at F = 32 frames
    on a 96x96 latent with the 32-row positional table - 18 432 pixels x 8 heads of 32x32 temporal scores TOGETHER WITH 64 frames of
    9 216-token spatial attention and the grid sizes that go with them (motion_module.py:286-304, 371-464; diffusers/models/
    attention.py:649-678).  The golden (oracle/make_golden_full.py cfg4full, 25 min of CPU) stores 7 of the 32 frames whole plus the
    L2 norm of every frame of the f32 output; both are checked."""
    if not os.path.exists(os.path.join(golden_dir, "cfg4_full_shape.npz")):
        pytest.skip("cfg4_full_shape.npz not generated (oracle/make_golden_full.py cfg4full)")
    g = _load(golden_dir, "cfg4_full_shape.npz")
    ocfg = Fn.UNetConfig(temporal_position_encoding_max_len=int(g["max_len"]))
    F, lat = int(g["frames"]), int(g["lat"])
    inp = W.seeded_inputs(ocfg, 1, F, lat, lat, seed=int(g["input_seed"]))
    x9 = torch.cat([Fn.build_model_input(inp["latents"], inp["first_image_latents"], inp["first_images_mask"])] * 2)
    keep = [int(k) for k in g["keep"]]
    ref32, ref16, drift = g["out_f32"].float(), g["out_bf16"].float(), float(g["drift_keep"])
    for dtype in (torch.float32, torch.bfloat16):
        eng = fullwidth.engine(ocfg, UNet3DConfig(temporal_position_encoding_max_len=int(g["max_len"])), dtype, seed=0)
        eng.prepare_context(inp["text"])
        _, temb = eng.prepare_time_embeddings([int(g["timestep"])], g["fps"].tolist(), g["flow"].tolist(), 2)
        out = eng.forward(_nhwc(x9, dtype), temb, 2, F, lat, lat).float().cpu().reshape(2, F, lat, lat, 4).permute(0, 4, 1, 2, 3)
        assert torch.isfinite(out).all()
        r32, r16 = rel(out[:, :, keep], ref32), rel(out[:, :, keep], ref16)
        norms = out.pow(2).sum(dim=(1, 3, 4)).sqrt()
        rn = ((norms - g["frame_norms_f32"]).abs() / g["frame_norms_f32"]).max().item()
        report(f"cfg4 FULL shape (F={F}, {lat}x{lat} latent, max_len 32) {dtype}: vs ref-f32 {r32:.3e}, vs ref-bf16-autocast {r16:.3e} "
               f"(ref-bf16 vs ref-f32 {drift:.3e} on the stored frames, {float(g['drift']):.3e} on all); worst per-frame norm error {rn:.3e}")
        if dtype == torch.float32:
            assert r32 < 1e-3 and rn < 1e-4, (r32, rn)
        else:
            assert r32 < BF16_FACTOR * drift and r16 < BF16_VS_BF16 * drift and rn < 2e-3, (r32, r16, drift, rn)
        del eng
        torch.cuda.empty_cache()


def test_cfg5_full_shape_forward_vs_oracle(golden_dir, fullwidth):
    """BASELINE configs[4] at its real shape (VERDICT r3 missing 3): one CFG-pair forward at 16 frames on a 64x64 latent with 16 IP
    tokens, the rectangle region mask and the first-frame latent concat (pipeline_animation.py:676-680, 716-723; animatediff/models/
    attention.py:49-127).  The golden pins oracle.functional WITH the reference's CPU-path temperature quirk to the real reference
    (`oracle_quirk_vs_ref`, asserted at generation time) and holds the engine to the oracle WITHOUT it = the deployed semantics;
    the bf16 yardstick is that same oracle under bf16-autocast (`drift_noquirk`)."""
    g = _load(golden_dir, "cfg5_full_shape.npz")
    assert float(g["oracle_quirk_vs_ref"]) < 1e-4
    ocfg = Fn.UNetConfig(use_ip_cross_attention=True, ip_num_tokens=int(g["ip_num_tokens"]), ip_scale=float(g["ip_scale"]))
    F, lat = int(g["frames"]), int(g["lat"])
    inp = W.seeded_inputs(ocfg, 1, F, lat, lat, seed=int(g["input_seed"]))
    x9 = torch.cat([Fn.build_model_input(inp["latents"], inp["first_image_latents"], g["first_images_mask"])] * 2)
    ref, ref16, drift = g["out_oracle_noquirk"].float(), g["out_oracle_noquirk_bf16"].float(), float(g["drift_noquirk"])
    for dtype in (torch.float32, torch.bfloat16):
        eng = fullwidth.engine(ocfg, UNet3DConfig(use_ip_cross_attention=True, ip_num_tokens=ocfg.ip_num_tokens, ip_scale=ocfg.ip_scale), dtype, seed=int(g["weight_seed"]))
        eng.prepare_context(inp["text"], inp["ip_tokens"])
        _, temb = eng.prepare_time_embeddings([int(g["timestep"])], g["fps"].tolist(), g["flow"].tolist(), 2)
        out = eng.forward(_nhwc(x9, dtype), temb, 2, F, lat, lat).float().cpu().reshape(2, F, lat, lat, 4).permute(0, 4, 1, 2, 3)
        assert torch.isfinite(out).all()
        r, r16 = rel(out, ref), rel(out, ref16)
        report(f"cfg5 FULL shape (16f@512^2, 16 IP tokens + rectangle mask) {dtype}: vs oracle-f32 (deployed semantics) {r:.3e}, vs that oracle under "
               f"bf16-autocast {r16:.3e} (its own drift {drift:.3e})")
        if dtype == torch.float32:
            assert r < 1e-3, r
        else:
            assert r < BF16_FACTOR * drift and r16 < BF16_VS_BF16 * drift, (r, r16, drift)
        del eng
        torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_cfg5_ip_mask_trajectory_vs_oracle(golden_dir, fullwidth, dtype):
    """BASELINE configs[4] as a trajectory: IP-Adapter branch (16 image tokens, scale 0.7) + rectangle region mask + first-frame
    latent concat, CFG 8, 5 DDIM steps at full widths (pipeline_animation.py:676-680, 716-723; attention.py:49-127).  The golden holds the
    REAL reference's trajectory (its CPU path uses the IP weight as attn2's softmax temperature - tests/test_oracle_golden.py pins
    the oracle WITH that quirk to it at every step) and the oracle WITHOUT the quirk = the deployed semantics the engine implements."""
    g = _load(golden_dir, "cfg5_trajectory.npz")
    ocfg = Fn.UNetConfig(use_ip_cross_attention=True, ip_num_tokens=int(g["ip_num_tokens"]), ip_scale=float(g["ip_scale"]))
    F, lat, steps = int(g["frames"]), int(g["lat"]), int(g["steps"])
    inp = W.seeded_inputs(ocfg, 1, F, lat, lat, seed=int(g["input_seed"]))
    eng = fullwidth.engine(ocfg, UNet3DConfig(use_ip_cross_attention=True, ip_num_tokens=ocfg.ip_num_tokens, ip_scale=ocfg.ip_scale), dtype, seed=int(g["weight_seed"]))
    got = {}
    DDIMSampler(eng, DDIMConfig()).sample(inp["latents"], g["text_embeddings"], steps, 8.0, inp["first_image_latents"], g["first_images_mask"],
                                          fps=[2], flow=[4], ip_tokens=inp["ip_tokens"], callback=lambda i, t, l: got.__setitem__(i, l.clone().cpu()))
    torch.cuda.synchronize()
    assert sorted(got) == list(range(steps))
    # the yardstick of the bf16 mode: the SAME no-quirk oracle under bf16-autocast (cfg5_yardstick.npz, oracle/make_golden_full.py
    # cfg5yard) - the reference's own stored drift is of its CPU path with the wrong attn2 temperature and is 2.5x larger (round 3
    # used it: a bound with 3x slack)
    y = _load(golden_dir, "cfg5_yardstick.npz")
    for i in range(steps):
        r, drift, drift_cpu_path = rel(got[i], g[f"step{i}_oracle_noquirk"]), float(y[f"drift_noquirk{i}"]), float(g[f"drift{i}"])
        report(f"cfg5 IP + mask trajectory step {i} {dtype}: vs oracle-f32 (deployed semantics) {r:.3e} (that oracle under bf16-autocast {drift:.3e}; "
               f"the reference's CPU path under bf16-autocast {drift_cpu_path:.3e})")
        assert r < (1e-3 if dtype == torch.float32 else BF16_FACTOR * drift), (i, r, drift)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_vae_decode_full_width_vs_reference(golden_dir, dtype):
    g = _load(golden_dir, "vae_full.npz")
    vcfg = VAEDecoderConfig()
    sdv = W.make_weights(W.vae_decoder_state_shapes(Fn.VAEConfig()), int(g["weight_seed"]))
    vae = VAEDecoderEngine(pack_vae_decoder(sdv, vcfg, dtype, DEV))
    out = vae.decode(g["z"] * vcfg.scaling_factor).cpu()
    ref = (g["out"] / 2 + 0.5).clamp(0, 1)
    ref16 = (g["out_bf16"] / 2 + 0.5).clamp(0, 1)
    e, drift = (out - ref).abs().max().item(), (ref16 - ref).abs().max().item()
    report(f"vae full width {dtype}: max abs err {e:.3e} (ref-bf16 vs ref-f32 {drift:.3e})")
    # bf16 measured 1.29 x the reference's own bf16 error; f16 (three more mantissa bits) must be well inside it
    assert e < (1e-4 if dtype == torch.float32 else 1.5 * drift if dtype == torch.bfloat16 else 0.4 * drift), e


# ---- FYC_F16: the precision class the reference deploys (torch.autocast("cuda") = float16, scripts/inference.py:294) ----------------------
# engine-f16 vs reference-f32 may be at most F16_FACTOR x the reference's OWN fp16-autocast drift (tests/golden/f16_yardstick.npz,
# oracle/make_golden_full.py f16yard), and must in any case be well inside the bf16 drift of the same checkpoint (F16_VS_BF16_DRIFT):
# an f16 mode that lost its three extra mantissa bits somewhere (a bf16 rounding left in a kernel) fails the second bound even where
# the yardstick file is missing a checkpoint.
F16_FACTOR = 1.1           # measured 0.88 .. 0.93 x over the small forward, cfg1 (5 steps) and cfg2 (steps 0 / 4 / 24)
F16_VS_BF16_DRIFT = 0.5


def _f16_bounds(yard, key, bf16_drift):
    bound = F16_VS_BF16_DRIFT * bf16_drift
    d16 = float(yard[key]) if yard is not None and key in yard else None
    if d16 is not None:
        bound = min(bound, F16_FACTOR * d16)
    return bound, d16


def test_f16_mode_vs_reference(golden_dir, engines):
    """the engine's float16 mode (every kernel on v_mfma_*_f16, 16-bit storage in IEEE half) against the REAL reference's f32 outputs: the
    small full-width forward, the whole cfg1 trajectory and the benchmarked cfg2 trajectory (steps 0 / 4 / 24)"""
    path = os.path.join(golden_dir, "f16_yardstick.npz")
    yard = _load(golden_dir, "f16_yardstick.npz") if os.path.exists(path) else None
    dtype = torch.float16
    g = _load(golden_dir, "unet_full_small_fwd.npz")
    cfg = Fn.UNetConfig()
    F, H, Wd = int(g["frames"]), int(g["h"]), int(g["w"])
    inp = W.seeded_inputs(cfg, 1, F, H, Wd, seed=int(g["input_seed"]))
    x9 = torch.cat([Fn.build_model_input(inp["latents"], inp["first_image_latents"], inp["first_images_mask"])] * 2)
    eng = engines(dtype)
    eng.prepare_context(inp["text"])
    _, temb = eng.prepare_time_embeddings([int(g["timestep"])], g["fps"].tolist(), g["flow"].tolist(), 2)
    out = eng.forward(_nhwc(x9, dtype), temb, 2, F, H, Wd).float().cpu().reshape(2, F, H, Wd, 4).permute(0, 4, 1, 2, 3)
    assert torch.isfinite(out).all()
    r32 = rel(out, g["out_f32"])
    bound, d16 = _f16_bounds(yard, "small_fwd_drift_f16", float(g["drift"]))
    report(f"full-width fwd (F=4, 16x16) {dtype}: vs ref-f32 {r32:.3e} (ref-fp16-autocast vs ref-f32 {d16}, ref-bf16-autocast {float(g['drift']):.3e})")
    assert r32 < bound, (r32, bound)
    for name, tag, keep, run in (("cfg1_trajectory.npz", "cfg1", (0, 1, 2, 3, 4), 5), ("cfg2_trajectory.npz", "cfg2", (0, 4, 24), 25)):
        if not os.path.exists(os.path.join(golden_dir, name)):
            continue
        g, got = _trajectory(golden_dir, engines, name, dtype, run)
        for i in keep:
            r32 = rel(got[i], g[f"step{i}_f32"])
            bound, d16 = _f16_bounds(yard, f"{tag}_drift_f16_{i}", float(g[f"drift{i}"]))
            report(f"{tag} step {i} {dtype}: vs ref-f32 {r32:.3e} (ref-fp16-autocast vs ref-f32 {d16}, ref-bf16-autocast {float(g[f'drift{i}']):.3e})")
            assert torch.isfinite(got[i]).all()
            assert r32 < bound, (tag, i, r32, bound)


def test_f16_sampling_is_bitwise_repeatable(golden_dir, engines):
    """same property as the bf16 / f32 modes: two eager runs of the sampling loop at the benchmarked shape give the same bits"""
    g = _load(golden_dir, "cfg1_trajectory.npz")
    inp = W.seeded_inputs(Fn.UNetConfig(), 1, 16, 64, 64, seed=int(g["input_seed"]))
    smp = DDIMSampler(engines(torch.float16), DDIMConfig())

    def run():
        out = smp.sample(inp["latents"], g["text_embeddings"], 2, 8.0, inp["first_image_latents"], inp["first_images_mask"], fps=[2], flow=[4])
        torch.cuda.synchronize()
        return out.cpu()
    a, b = run(), run()
    assert torch.isfinite(a).all() and torch.equal(a, b), rel(a, b)
