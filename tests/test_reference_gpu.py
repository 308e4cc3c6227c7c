"""The same-device, same-precision yardstick: the UNMODIFIED reference UNet3DConditionModel on `cuda:0` (eager PyTorch-ROCm) under
the real `torch.autocast("cuda")` (/root/reference/scripts/inference.py:294), imported from the git-ignored byte copies under
oracle/_ref/ (oracle/stage_ref_scripts.py, oracle/gpu_reference.py - test infrastructure).

  (i)  validates oracle/autocast_emul.py: every bf16 / f16 bound of tests/test_fullwidth_gpu.py is relative to a CPU EMULATION of
       autocast's cast lists; here the real thing runs on the chip and its drift (autocast vs f32, same device) is compared with the
       emulated drift the goldens store;
  (ii) holds the engine to the MEASURED same-device drift: engine-16-bit vs the device's own f32 reference run must be no further than
       SAME_DEVICE_FACTOR x (device autocast run vs device f32 run).

Numbers go to gpurun_out/parity_report.txt (copied to profiles/).
"""
import os

import pytest
import torch

from followyourclick_amd.engine import UNet3DConfig
from oracle import functional as Fn
from oracle import weights as W

from test_engine_gpu import _load, _nhwc, rel, report

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# engine 16-bit mode vs the device's f32 reference run <= SAME_DEVICE_FACTOR x (the reference's own autocast run vs its f32 run, same
# device).  The CPU-emulated rule is 1.1 x (tests/test_fullwidth_gpu.py, measured 0.88-0.95 x); the real autocast also keeps softmax /
# norms in f32 but runs its GEMMs through other kernels (hipBLASLt / MIOpen, other accumulation orders), so the ratio is measured here
# and the bound leaves the same ~15 % the emulated rule leaves.
SAME_DEVICE_FACTOR = 1.15
# the emulation is accepted as a yardstick when the drift it predicts is within this band of the drift measured on the device.
# Measured (round 5, profiles/r05_parity_report.txt): bf16 1.10 x the emulated prediction, f16 1.52 x - the deployed branch keeps the
# attention probabilities in half precision inside the fused kernel, which the cast-list emulation (softmax in f32, then a cast for the
# P V product) does not model; the emulated f16 yardstick is therefore the STRICTER of the two for the engine.
EMUL_BAND = (0.6, 1.7)


@pytest.fixture(scope="module")
def small_case(golden_dir, device_reference):
    """the unet_full_small_fwd inputs, and the reference's three runs of them on the device"""
    g = _load(golden_dir, "unet_full_small_fwd.npz")
    F, H, Wd = int(g["frames"]), int(g["h"]), int(g["w"])
    inp = W.seeded_inputs(Fn.UNetConfig(), 1, F, H, Wd, seed=int(g["input_seed"]))
    x9 = torch.cat([Fn.build_model_input(inp["latents"], inp["first_image_latents"], inp["first_images_mask"])] * 2)
    runs = device_reference("small")
    for name in ("f32", "bf16", "f16"):
        assert torch.isfinite(runs[name]).all(), name
    return g, inp, x9, runs


def test_device_reference_reproduces_the_cpu_golden(small_case):
    """the staged byte copies on PyTorch-ROCm compute what /root/reference computed on the CPU: f32 run vs the committed golden"""
    g, inp, x9, runs = small_case
    r = rel(runs["f32"], g["out_f32"])
    report(f"same-device reference (F=4, 16x16) f32 on cuda vs CPU golden: {r:.3e}")
    assert r < 1e-3, r       # (MIOpen / hipBLASLt f32 kernels may split K differently from oneDNN; measured value in the report)


def test_real_autocast_validates_the_cpu_emulation(golden_dir, small_case):
    g, inp, x9, runs = small_case
    yard = _load(golden_dir, "f16_yardstick.npz") if os.path.exists(os.path.join(golden_dir, "f16_yardstick.npz")) else {}
    d_bf16 = rel(runs["bf16"], runs["f32"])
    d_f16 = rel(runs["f16"], runs["f32"])
    e_bf16 = float(g["drift"])
    report(f"same-device reference drift: torch.autocast(cuda, bf16) vs f32 {d_bf16:.3e} (CPU emulation predicted {e_bf16:.3e}, ratio {d_bf16 / e_bf16:.2f}); "
           f"bf16 device run vs emulated bf16 run {rel(runs['bf16'], g['out_bf16']):.3e}")
    assert EMUL_BAND[0] < d_bf16 / e_bf16 < EMUL_BAND[1], (d_bf16, e_bf16)
    if "small_fwd_drift_f16" in yard:
        e_f16 = float(yard["small_fwd_drift_f16"])
        report(f"same-device reference drift: torch.autocast(cuda, f16) vs f32 {d_f16:.3e} (CPU emulation predicted {e_f16:.3e}, ratio {d_f16 / e_f16:.2f})")
        assert EMUL_BAND[0] < d_f16 / e_f16 < EMUL_BAND[1], (d_f16, e_f16)


@pytest.mark.parametrize("mode", ["bf16", "f16"])
def test_engine_vs_same_device_reference(small_case, fullwidth, mode):
    """the engine's 16-bit modes against the reference's runs ON THE SAME CHIP: no further from the f32 run than the reference's own
    autocast run of that precision (x SAME_DEVICE_FACTOR); the distance between the two 16-bit runs is reported"""
    g, inp, x9, runs = small_case
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16}[mode]
    F, H, Wd = int(g["frames"]), int(g["h"]), int(g["w"])
    eng = fullwidth.engine(Fn.UNetConfig(), UNet3DConfig(), dtype, seed=int(g["weight_seed"]))
    eng.prepare_context(inp["text"])
    _, temb = eng.prepare_time_embeddings([int(g["timestep"])], g["fps"].tolist(), g["flow"].tolist(), 2)
    out = eng.forward(_nhwc(x9, dtype), temb, 2, F, H, Wd).float().cpu().reshape(2, F, H, Wd, 4).permute(0, 4, 1, 2, 3)
    assert torch.isfinite(out).all()
    drift = rel(runs[mode], runs["f32"])
    r32, r16 = rel(out, runs["f32"]), rel(out, runs[mode])
    report(f"engine {mode} vs same-device reference (F=4, 16x16): vs ref-f32-on-cuda {r32:.3e} = {r32 / drift:.2f} x the reference's own "
           f"{mode}-autocast drift on this chip ({drift:.3e}); vs ref-{mode}-autocast-on-cuda {r16:.3e}")
    assert r32 < SAME_DEVICE_FACTOR * drift, (r32, drift)


# ---- BASELINE configs[3] / configs[4] as multi-step trajectories at FULL shape (round-4 review, item 10) -------------------------------
# The CPU cannot afford them (one F = 32, 96x96 CFG-pair forward of the reference is 12 minutes on 8 cores), the chip under test can: the
# reference's own UNet3DConditionModel + DDIMScheduler run the first steps of the schedule on `cuda:0` in f32 and under the real autocast
# (oracle/gpu_reference.py::reference_trajectory; its f32 forward reproduces the CPU golden to 3e-6, test above), and the engine is held to
# them with the rules of tests/test_fullwidth_gpu.py: f32 mode <= 1e-3 of the f32 reference after every step, 16-bit mode no further from
# the f32 reference than SAME_DEVICE_FACTOR x the reference's own autocast run.
class _Stop(Exception):
    pass


def _engine_trajectory(eng, inp, num_steps, run_steps, mask=None, ip=None):
    from followyourclick_amd.engine import DDIMConfig
    from followyourclick_amd.engine.sampler import DDIMSampler
    got = {}

    def cb(i, t, l):
        got[i] = l.detach().float().cpu()
        if i + 1 >= run_steps:
            raise _Stop
    try:
        DDIMSampler(eng, DDIMConfig()).sample(inp["latents"], inp["text"], num_steps, 8.0, inp["first_image_latents"],
                                              inp["first_images_mask"] if mask is None else mask, fps=[2], flow=[4], ip_tokens=ip, callback=cb)
    except _Stop:
        pass
    torch.cuda.synchronize()
    return got


def _hold_to_device_reference(tag, what, ecfg, device_reference, fullwidth):
    from oracle import gpu_reference as G          # (module import only: the reference packages are imported in the subprocess)
    frames, lat, num_steps, run_steps, seed, ocfg, mask, use_ip = G.TRAJECTORIES[what]()
    ref = device_reference(what)
    inp = W.seeded_inputs(ocfg, 1, frames, lat, lat, seed=seed)
    ip = inp["ip_tokens"] if use_ip else None
    eng_f32 = {}
    for dtype, mode in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        got = _engine_trajectory(fullwidth.engine(ocfg, ecfg, dtype, seed=0), inp, num_steps, run_steps, mask=mask, ip=ip)
        if mode == "f32":
            eng_f32 = got
        for i in range(run_steps):
            assert torch.isfinite(got[i]).all(), (tag, mode, i)
            # the f32 side of step i: the reference's own f32 run where it was computed (oracle.gpu_reference.F32_STEPS: configs[3] runs ONE
            # f32 step of the reference, ~50 s of eager kernels at 32f@768^2); beyond that the engine's f32 run, which the steps before
            # have just held to the reference's to ~1e-7 and which tests/test_fullwidth_gpu.py pins at this shape against a stored golden
            have_ref32 = i in ref["f32"] if isinstance(ref["f32"], dict) else i < len(ref["f32"])
            f32_side = ref["f32"][i] if have_ref32 else eng_f32[i]
            side = "the reference's f32 run on this chip" if have_ref32 else "the engine's f32 run (held to the reference's at the steps before)"
            r32 = rel(got[i], f32_side)
            drift = rel(ref["bf16"][i], f32_side)
            if mode == "f32":
                if have_ref32:
                    report(f"{tag} step {i} engine f32 vs {side}: {r32:.3e}")
                    assert r32 < 1e-3, (tag, i, r32)
            else:
                report(f"{tag} step {i} engine bf16 vs {side} {r32:.3e} = {r32 / drift:.2f} x the reference's own bf16-autocast drift ({drift:.3e}); "
                       f"vs ref-bf16-autocast-on-cuda {rel(got[i], ref['bf16'][i]):.3e}")
                assert r32 < SAME_DEVICE_FACTOR * drift, (tag, i, r32, drift)


def test_cfg3_full_shape_trajectory_vs_device_reference(device_reference, fullwidth):
    """(in the default `-m gpu` run since round 6: the reference side comes from the session's background process, tests/conftest.py)
    BASELINE configs[3]: 32 frames at 768x768 (96x96 latent, 9 216-token spatial attention, 32x32 temporal scores, 32-row positional
    table), the first 2 steps of the 50-step schedule (motion_module.py:286-304, 371-464; diffusers/models/attention.py:649-678)"""
    _hold_to_device_reference("cfg3 full shape (32f@768^2)", "cfg3", UNet3DConfig(temporal_position_encoding_max_len=32), device_reference, fullwidth)


def test_cfg4_full_shape_ip_trajectory_vs_device_reference(device_reference, fullwidth):
    """BASELINE configs[4]: 16 frames at 512x512 with 16 IP-Adapter image tokens (scale 0.7), the rectangle region mask and the first-frame
    concat, the first 2 steps of the 25-step schedule.  On the chip the reference takes its DEPLOYED attention branch (the memory-efficient
    one, animatediff/models/attention.py:92-93, 109-110), which does not carry the CPU path's attn2-temperature quirk: the engine is compared
    with the real reference directly here, not through the no-quirk oracle."""
    _hold_to_device_reference("cfg4 full shape (16f@512^2 + 16 IP tokens + region mask)", "cfg4ip",
                              UNet3DConfig(use_ip_cross_attention=True, ip_num_tokens=16, ip_scale=0.7), device_reference, fullwidth)
