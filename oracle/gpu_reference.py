"""The UNMODIFIED reference UNet3DConditionModel on the MI355X itself (eager PyTorch-ROCm), as the reference deploys it:
`torch.autocast("cuda")` around the denoising loop (/root/reference/scripts/inference.py:294), memory-efficient attention asserted
(:157-158).  TEST INFRASTRUCTURE - nothing under followyourclick_amd/ imports this file.

Two uses:
  * tests/test_reference_gpu.py: the same-device, same-precision yardstick.  The bf16 / f16 bounds of tests/test_fullwidth_gpu.py
    are relative to a CPU EMULATION of autocast (oracle/autocast_emul.py); here the real autocast runs on the same chip as the
    engine, which (i) validates that emulation and (ii) gives the engine a measured same-device drift to be held to.
  * bench.py's `gpu_reference` leg: frames/s of the reference's own GPU path beside the CPU number (run as a subprocess of bench.py:
    `python -m oracle.gpu_reference --json ...`), on random-init weights of the same architecture, synthetic inputs of the benchmark
    shape, 1 untimed + N timed CFG-pair forwards (= DDIM steps), extrapolated to the clip.

The model code is imported from the git-ignored byte copies under oracle/_ref/ (oracle/stage_ref_scripts.py; /root/reference does not
exist on the GPU box) through oracle/refshim.py.  What is fabricated is the environment, never the reference:
  * xformers is not installed in this image.  `attention="sdpa"` registers a stand-in `xformers` module whose
    `ops.memory_efficient_attention(q, k, v, attn_bias)` is torch's fused scaled-dot-product attention - the reference then takes its
    deployed `_memory_efficient_attention_xformers` branch (animatediff/models/attention.py:92-93, diffusers/models/attention.py:
    649-678); `attention="eager"` leaves xformers absent and the reference materialises the scores (baddbmm + softmax).
"""
import argparse
import importlib.machinery as _M
import json
import sys
import time
import types

import torch


SCORE_BYTES_LIMIT = 8 << 30      # scores (+ their softmax copy) one SDPA call of the stand-in may materialise


def install_sdpa_xformers():
    """a stand-in `xformers` package: memory_efficient_attention -> torch SDPA (same math: softmax(q k^T / sqrt(d) + bias) v)"""
    if "xformers" in sys.modules and getattr(sys.modules["xformers"], "_fyc_stub", False):
        return
    import torch.nn.functional as F
    xf = types.ModuleType("xformers")
    xf.__spec__ = _M.ModuleSpec("xformers", None, is_package=True)
    xf.__path__ = []
    xf.__version__ = "0.0.0+sdpa-stand-in"
    xf._fyc_stub = True
    ops = types.ModuleType("xformers.ops")
    ops.__spec__ = _M.ModuleSpec("xformers.ops", None)

    def memory_efficient_attention(query, key, value, attn_bias=None, p=0.0, scale=None, op=None):
        # SDPA may take its materialising "math" path: cut the (batch x heads) axis so that one slice's scores stay below ~8 GiB
        # (F = 32 at 96x96 would otherwise ask for 174 GB in f32) - the same per-(batch, head) arithmetic
        # (16-bit inputs as well: on this ROCm build SDPA takes the materialising path for d = 40 at 9 216 tokens too - 162 GiB asked for)
        n = query.shape[0]
        per = query.shape[-2] * key.shape[-2] * query.element_size() * 2
        step = max(1, min(n, SCORE_BYTES_LIMIT // max(per, 1)))
        if step >= n:
            return F.scaled_dot_product_attention(query, key, value, attn_mask=attn_bias, dropout_p=p, scale=scale)
        out = [F.scaled_dot_product_attention(query[i:i + step], key[i:i + step], value[i:i + step],
                                              attn_mask=None if attn_bias is None else attn_bias[i:i + step], dropout_p=p, scale=scale)
               for i in range(0, n, step)]
        return torch.cat(out)
    ops.memory_efficient_attention = memory_efficient_attention
    xf.ops = ops
    sys.modules["xformers"] = xf
    sys.modules["xformers.ops"] = ops


def build_reference_unet(device, max_len=24, attention="sdpa", seed=0, ocfg=None):
    """the real animatediff.models.unet.UNet3DConditionModel at SD-1.5 widths with the seeded weights every golden uses"""
    from . import functional as Fn
    from . import refshim
    from . import weights as W
    if attention == "sdpa":
        install_sdpa_xformers()
    refshim.install()
    if attention == "sdpa":
        import diffusers.utils.import_utils as iu      # the reference's own module: package metadata of the stand-in does not exist
        iu._xformers_available = True
    from .make_golden import ref_unet
    cfg = ocfg or (Fn.UNetConfig(temporal_position_encoding_max_len=max_len) if max_len != 24 else Fn.UNetConfig())
    unet = ref_unet(cfg).eval()
    unet.load_state_dict(W.make_weights(W.unet_state_shapes(cfg), seed=seed), strict=True)
    if cfg.use_ip_cross_attention:          # the goldens feed projected image tokens directly (oracle/make_golden_full.py::ip)
        class Proj(torch.nn.Module):
            def forward(self, feat):
                return feat
        unet.image_proj_model = Proj()
    unet = unet.to(device)
    if attention == "sdpa":
        unet.enable_xformers_memory_efficient_attention()
    return cfg, unet


def forward(unet, x9, t, text, fps, flow, autocast_dtype=None, **kw):
    """one UNet3DConditionModel.forward as scripts/inference.py runs it: no_grad, torch.autocast("cuda", dtype) unless dtype is None (f32)"""
    dev = next(unet.parameters()).device
    args = (x9.to(dev), torch.as_tensor(t).to(dev), text.to(dev))
    kws = dict(use_fps_condition=True, fps_tensor=fps.to(dev), flow_control=flow.to(dev), **kw)
    with torch.no_grad():
        if autocast_dtype is None:
            return unet(*args, **kws).sample.float()
        with torch.autocast("cuda", dtype=autocast_dtype):
            return unet(*args, **kws).sample.float()


SCHED_KW = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1,
                clip_sample=False, prediction_type="v_prediction", rescale_betas_zero_snr=True)


def reference_trajectory(unet, inp, num_steps, run_steps, autocast_dtype=None, guidance_scale=8.0, mask=None, ip_tokens=None, fps=2, flow=4):
    """The first `run_steps` of a `num_steps`-step sampling run with the REAL UNet3DConditionModel and the REAL DDIMScheduler on the
    device: the loop body of AnimationPipeline.__call__ (pipeline_animation.py:686-773: first-frame / mask concat, CFG duplication,
    guidance, scheduler.step) around them, under `torch.autocast("cuda", dtype)` as the reference runs it (:686) or in f32.  The loop
    body itself is pinned by the full-pipeline goldens (cfg1 / cfg2 / cfg5_trajectory.npz); here it lets the trajectories the CPU cannot
    afford - BASELINE configs[3] and [4] at full shape - come from the reference's own modules on the chip under test.
    Returns {step index: latents (cpu, f32)}."""
    import contextlib
    from diffusers.schedulers.scheduling_ddim import DDIMScheduler      # the reference's (oracle/refshim.py)
    from . import functional as Fn
    dev = next(unet.parameters()).device
    sched = DDIMScheduler(**SCHED_KW)
    sched.set_timesteps(num_steps)
    lat = inp["latents"].clone().to(dev)
    first = inp["first_image_latents"].to(dev)
    m = (inp["first_images_mask"] if mask is None else mask).to(dev)
    text = inp["text"].to(dev)
    fps_t, flow_t = torch.tensor([fps, fps], device=dev), torch.tensor([flow, flow], device=dev)
    kw = dict(use_fps_condition=True, fps_tensor=fps_t, flow_control=flow_t)
    if ip_tokens is not None:
        kw.update(use_ip_cross_attention=True, reference_images_clip_feat=ip_tokens.to(dev))
    out = {}
    ctx = torch.autocast("cuda", dtype=autocast_dtype) if autocast_dtype is not None else contextlib.nullcontext()
    with torch.no_grad(), ctx:
        for i, t in enumerate(sched.timesteps[:run_steps]):
            x9 = torch.cat([Fn.build_model_input(lat, first, m)] * 2)
            pred = unet(x9, t.to(dev), text, **kw).sample.to(lat.dtype)
            u, c = pred.chunk(2)
            pred = u + guidance_scale * (c - u)
            lat = sched.step(pred, t, lat).prev_sample
            out[i] = lat.detach().float().cpu()
    return out


def time_reference(frames=16, size=512, ddim_steps=25, dtype="bf16", attention="sdpa", timed=2):
    dev = torch.device("cuda", 0)
    t0 = time.time()
    cfg, unet = build_reference_unet(dev, max_len=max(24, frames), attention=attention)
    t_build = time.time() - t0
    h = w = size // 8
    g = torch.Generator().manual_seed(1)
    text = torch.randn(2, 77, cfg.cross_attention_dim, generator=g)
    fps, flow = torch.tensor([2, 2]), torch.tensor([4, 4])
    ac = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": None}[dtype]
    times = []
    for i in range(1 + timed):
        x9 = torch.randn(2, cfg.conv_in_channels, frames, h, w, generator=g).to(dev)
        torch.cuda.synchronize()
        t0 = time.time()
        out = forward(unet, x9, 961 - 40 * i, text, fps, flow, ac)
        torch.cuda.synchronize()
        times.append(time.time() - t0)
        assert torch.isfinite(out).all()
    dt = sum(times[1:]) / timed
    return dict(value=round(frames / (ddim_steps * dt), 4), unit="frames/s", kind="reference",
                s_per_ddim_step=round(dt, 4), first_step_s=round(times[0], 2), build_s=round(t_build, 1), dtype=dtype, attention=attention,
                peak_mem_gib=round(torch.cuda.max_memory_allocated() / 2**30, 1), torch=torch.__version__,
                sample=f"the unmodified reference UNet3DConditionModel (oracle/_ref byte copies), eager PyTorch-ROCm under torch.autocast('cuda', {dtype}), "
                       f"attention = {'torch SDPA behind a stand-in xformers module (the deployed branch)' if attention == 'sdpa' else 'materialised baddbmm + softmax (no xformers in this image)'}; "
                       f"{timed} timed CFG-pair forwards (= DDIM steps) of {frames}f@{size}^2 after one untimed, {dt:.3f}s per step; frames/s = {frames} / ({ddim_steps} x {dt:.3f}s); "
                       f"the guidance / scheduler arithmetic between forwards (< 0.1 % of a step) is not included")


def time_reference_cpu(frames=16, size=512, ddim_steps=25, timed=1, cores=None):
    """bench.py's `cpu_baseline` leg with kind "reference": the UNMODIFIED reference UNet3DConditionModel on the HOST cores, fp32, as its CPU
    path runs it (no xformers: baddbmm + softmax; the 16 spatial attn1 take the reference's own sliced-attention valve, `_slice_size = 8`, which
    is bit-identical math and bounds the score tensor - the same setting the goldens were generated with, oracle/make_golden_full.py).  Threads
    are bound to `cores` (bench.py passes the physical cores of NUMA node 0).  One untimed CFG-pair forward at the shape, then `timed` timed ones."""
    import os
    if cores:
        os.sched_setaffinity(0, set(cores))
        torch.set_num_threads(len(cores))
    t0 = time.time()
    cfg, unet = build_reference_unet("cpu", max_len=max(24, frames), attention="eager")
    for m in unet.modules():
        if m.__class__.__name__ == "BasicTransformerBlock" and hasattr(m, "attn1"):
            m.attn1._slice_size = 8
    t_build = time.time() - t0
    h = w = size // 8
    g = torch.Generator().manual_seed(1)
    text = torch.randn(2, 77, cfg.cross_attention_dim, generator=g)
    fps, flow = torch.tensor([2, 2]), torch.tensor([4, 4])
    times = []
    for i in range(1 + timed):
        x9 = torch.randn(2, cfg.conv_in_channels, frames, h, w, generator=g)
        t0 = time.time()
        out = forward(unet, x9, 961 - 40 * i, text, fps, flow, None)
        times.append(time.time() - t0)
        assert torch.isfinite(out).all()
    dt = sum(times[1:]) / timed
    n = torch.get_num_threads()
    return dict(value=frames / (ddim_steps * dt), unit="frames/s", cores=n, kind="reference", s_per_frame_per_ddim_step=round(dt / frames, 2),
                build_s=round(t_build, 1), first_step_s=round(times[0], 1),
                sample=f"the unmodified reference UNet3DConditionModel (oracle/_ref byte copies) on the host cores, fp32, its CPU attention path: {timed} timed "
                       f"DDIM step(s) (CFG-pair forward at {size}x{size}) on ALL {frames} frames of the clip, {dt:.1f}s per step = {dt / frames:.2f} s/frame, after one "
                       f"untimed step at the same shape ({times[0]:.1f}s); {n} threads bound to physical cores of NUMA node 0; frames/s = {frames} / ({ddim_steps} x {dt:.1f}s)")


class GpuTurn:
    """The reference subprocess and the test session that started it take TURNS on the chip.  Round 6 first let the two run side by side:
    the reference's eager f32 forwards at 32f@768^2 made every test that ran beside them 3-8x slower (the suite 1 163 s on one box against
    the driver's 1 200-s limit).  Now only the HOST side of the reference (building the 1.28 B-parameter model, ~40 s per configuration)
    overlaps the tests.  Protocol, files in the session's out-dir: the subprocess creates `ref_wants_gpu`, takes flock(`gpu.lock`) - the test
    that is running finishes first -, computes, releases, removes the flag; tests/conftest.py holds the lock for the length of each GPU test
    and does not start one while the flag exists.  Without an out-dir (single dumps, bench.py) the context does nothing."""

    def __init__(self, out_dir):
        import os
        self.dir = out_dir if (out_dir and os.environ.get("FYC_REF_NO_TURNS") != "1") else None
        self.fd = None

    def __enter__(self):
        if self.dir is None:
            return self
        import fcntl
        import os
        open(os.path.join(self.dir, "ref_wants_gpu"), "w").close()
        self.fd = os.open(os.path.join(self.dir, "gpu.lock"), os.O_CREAT | os.O_RDWR)
        fcntl.flock(self.fd, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        if self.dir is None:
            return False
        import fcntl
        import os
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        fcntl.flock(self.fd, fcntl.LOCK_UN)
        os.close(self.fd)
        self.fd = None
        try:
            os.remove(os.path.join(self.dir, "ref_wants_gpu"))
        except FileNotFoundError:
            pass
        return False


def dump(what, out=None, out_dir=None):
    """tests/test_reference_gpu.py runs the reference in a SUBPROCESS (`python -m oracle.gpu_reference --dump a,b --out-dir D`, one per
    test session, started by tests/conftest.py while the kernel tests run): the reference's `animatediff` / `diffusers` packages and the
    product's drop-in packages of the same names cannot live in one interpreter.  Inputs are re-derived from the goldens' seeds on both
    sides; only the reference's outputs cross the files (`D/<what>.pt`, written under a temporary name and renamed when complete).
    One model is built per distinct configuration: `small` runs on the max_len-32 model of `cfg3` when both are asked for (the
    positional table is sliced to the clip length, motion_module.py:303 - the first 24 rows are the same numbers)."""
    import os
    import numpy as np
    from . import functional as Fn
    from . import weights as W
    dev = torch.device("cuda", 0)
    golden = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    whats = what.split(",")
    models = {}
    turn = GpuTurn(out_dir)

    def model(ocfg):
        key = repr(ocfg)
        if key not in models:
            models.clear()                       # one reference model on the device at a time
            torch.cuda.empty_cache()
            models[key] = build_reference_unet("cpu", attention="sdpa", ocfg=ocfg)[1]      # on the HOST: this is the part that overlaps the test session
        return models[key]

    for w in whats:
        t0 = time.time()
        if w == "small":          # the unet_full_small_fwd inputs in f32 and under the real bf16 / f16 autocast
            g = np.load(os.path.join(golden, "unet_full_small_fwd.npz"))
            unet = model(TRAJECTORIES["cfg3"]()[5] if "cfg3" in whats else Fn.UNetConfig())
            inp = W.seeded_inputs(Fn.UNetConfig(), 1, int(g["frames"]), int(g["h"]), int(g["w"]), seed=int(g["input_seed"]))
            x9 = torch.cat([Fn.build_model_input(inp["latents"], inp["first_image_latents"], inp["first_images_mask"])] * 2)
            t_host = time.time() - t0
            with turn:
                t1 = time.time()
                unet.to(dev)
                res = {name: forward(unet, x9, int(g["timestep"]), inp["text"], torch.from_numpy(g["fps"]), torch.from_numpy(g["flow"]), ac).cpu()
                       for name, ac in (("f32", None), ("bf16", torch.bfloat16), ("f16", torch.float16))}
                t_dev = time.time() - t1
        else:
            frames, lat, num_steps, run_steps, seed, ocfg, mask, use_ip = TRAJECTORIES[w]()
            unet = model(ocfg)
            inp = W.seeded_inputs(ocfg, 1, frames, lat, lat, seed=seed)
            # (f32_steps < run_steps: the reference's f32 forward at 32f@768^2 takes ~50 s on the chip - eager kernels - and the f32 mode of the
            # engine is pinned at that shape by the stored golden of tests/test_fullwidth_gpu.py as well: one f32 step, two under autocast)
            t_host = time.time() - t0
            with turn:
                t1 = time.time()
                unet.to(dev)
                res = {}
                for name, ac, n in (("bf16", torch.bfloat16, run_steps), ("f32", None, F32_STEPS.get(w, run_steps))):
                    t2 = time.time()
                    res[name] = reference_trajectory(unet, inp, num_steps, n, ac, mask=mask, ip_tokens=inp["ip_tokens"] if use_ip else None)
                    torch.cuda.synchronize()
                    print(f"[gpu_reference] {w} {name}: {n} step(s) in {time.time() - t2:.1f} s", flush=True)
                t_dev = time.time() - t1
        res["seconds"] = time.time() - t0
        path = out if (out and len(whats) == 1) else os.path.join(out_dir, w + ".pt")
        torch.save(res, path + ".tmp")
        os.replace(path + ".tmp", path)
        res["seconds_host"], res["seconds_device"] = t_host, t_dev
        print(f"[gpu_reference] {w}: {res['seconds']:.1f} s (host side {t_host:.1f} s, its turn on the chip {t_dev:.1f} s, the rest waiting for the turn)", flush=True)


def rectangle_mask(lat):
    mask = torch.zeros(1, 1, 1, lat, lat)
    mask[..., lat // 4: 3 * lat // 4, lat // 4: 3 * lat // 4] = 1.0          # the centred rectangle of SURVEY.md 8d (a SAM box stand-in)
    return mask


def _traj_cfg3():
    from . import functional as Fn      # BASELINE configs[3]: 32 frames at 768x768, the first 2 steps of the 50-step schedule
    return 32, 96, 50, 2, 64, Fn.UNetConfig(temporal_position_encoding_max_len=32), None, False


def _traj_cfg4ip():
    from . import functional as Fn      # BASELINE configs[4]: 16 frames at 512x512 + 16 IP tokens + region mask, the first 2 of 25 steps
    return 16, 64, 25, 2, 65, Fn.UNetConfig(use_ip_cross_attention=True, ip_num_tokens=16, ip_scale=0.7), rectangle_mask(64), True


TRAJECTORIES = {"cfg3": _traj_cfg3, "cfg4ip": _traj_cfg4ip}
F32_STEPS = {"cfg3": 1}          # f32 steps of the reference where fewer than the trajectory's run_steps are computed


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dump", default=None, help="comma-separated subset of small, cfg3, cfg4ip")
    ap.add_argument("--out", default=None)
    ap.add_argument("--out-dir", default=None)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--ddim-steps", type=int, default=25)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--attention", default="sdpa", choices=["sdpa", "eager"])
    ap.add_argument("--timed", type=int, default=2)
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--cpu-baseline", action="store_true", help="time the reference on the HOST cores instead (bench.py's cpu_baseline leg, kind 'reference')")
    ap.add_argument("--cores", default="", help="comma-separated logical cpu ids to bind to")
    a = ap.parse_args()
    if a.cpu_baseline:
        r = time_reference_cpu(a.frames, a.size, a.ddim_steps, a.timed, [int(c) for c in a.cores.split(",") if c] or None)
        print("CPU_REFERENCE " + json.dumps(r))
        sys.exit(0)
    if a.dump:
        dump(a.dump, a.out, a.out_dir)
        sys.exit(0)
    r = time_reference(a.frames, a.size, a.ddim_steps, a.dtype, a.attention, a.timed)
    print("GPU_REFERENCE " + json.dumps(r) if a.json else r)
