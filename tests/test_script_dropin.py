"""The reference's OWN entry script on top of the drop-in packages (SURVEY.md 8b1 / BASELINE.json north_star: "scripts/inference*.py
drop in unchanged").

`/root/reference/scripts/inference.py` is executed UNMODIFIED through runpy after `followyourclick_amd.install_dropin()`:
model construction from a `pretrained_model_path` (:152-155), the `is_xformers_available` assert (:157-158), the motion-module
checkpoint with its `module.` prefix (:170-181), `DistributedSampler` under --ddp (:260), first image + region mask from files,
`vae.encode` of the first frame (:356-358), mask plumbing (:361-365), `AnimationPipeline.__call__` (:374-395) and
`save_videos_grid` (:398-403, :424).  What is fabricated / stubbed is the ENVIRONMENT, never the script: tests/script_env.py.

The latents the script's pipeline call hands to the VAE decoder are then compared with the oracle's denoising loop on the same
weights, noise, text states and conditioning (captured at the pipeline's own method boundaries).

Runs from the reference tree (build container) or, on the GPU box, from the byte copies `oracle/stage_ref_scripts.py` puts under the
git-ignored oracle/_ref/ (`__graft_entry__.build()` stages them; they travel with the repo snapshot): the `gpu` variants execute
the scripts on the HIP kernels.  Round 4 adds `scripts/inference_org.py` (plain text-to-video: prompt file, `PromptDataset`,
motion-module checkpoint without the `state_dict` wrapper, `video_scale`) and `scripts/inference_w_image_cond.py` (first image from
the 2-D `StableDiffusionPipeline` at its hard-coded 448x768 / 50 steps, `.ckpt` spatial weights, CLIP vision model + image
processor construction) - unmodified, through the same scaffolding.
"""
import os
import runpy
import sys

import pytest
import torch

import script_env as E

pytestmark = pytest.mark.skipif(not os.path.exists(E.REF_SCRIPT), reason="the reference tree (/root/reference) is not on this box")

PROMPT = "a corgi waving its tail"


def _run_script(tmp_path, monkeypatch, device_is_gpu: bool, port: int):
    import followyourclick_amd
    from followyourclick_amd import ops as ops_mod
    monkeypatch.setattr(sys, "path", list(sys.path))
    saved_modules = {k: v for k, v in sys.modules.items() if k.split(".")[0] in ("animatediff", "diffusers", "ip_adapter")}
    if not device_is_gpu:
        from emu_ops import EmuOps
        monkeypatch.setattr(ops_mod, "impl", EmuOps())          # CPU box: host orchestration on the op emulator (tests only)
        E.alias_cuda_to_cpu(monkeypatch)
    followyourclick_amd.install_dropin(force=True)
    E.install_absent_packages(monkeypatch)
    root = str(tmp_path)
    fab = E.fabricate_model_dir(root)
    steps, size, frames = 2, 64, 2
    cfg_path, sheet, img_path, mask_path = E.write_run_files(root, fab["motion_ckpt"], steps, size)
    E.patch_authors_environment(monkeypatch, img_path, mask_path, PROMPT)
    for k, v in dict(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)).items():
        monkeypatch.setenv(k, v)

    # capture points: the pipeline's own method boundaries (class-level wrappers, installed before the script imports the class)
    from animatediff.pipelines.pipeline_animation import AnimationPipeline
    seen = {}

    def spy(name, keep):
        real = getattr(AnimationPipeline, name)

        def wrapper(self, *a, **k):
            out = real(self, *a, **k)
            keep(self, a, k, out)
            return out
        monkeypatch.setattr(AnimationPipeline, name, wrapper)
    spy("_encode_prompt", lambda s, a, k, out: seen.__setitem__("text", out.detach().float().cpu().clone()))
    spy("prepare_latents", lambda s, a, k, out: seen.__setitem__("noise", out.detach().float().cpu().clone()))
    real_decode = AnimationPipeline.decode_latents

    def decode(self, latents):
        seen["final"] = latents.detach().float().cpu().clone()
        seen["unet_sd"] = {k: v.detach().float().cpu().clone() for k, v in self.unet.state_dict().items()}
        return real_decode(self, latents)
    monkeypatch.setattr(AnimationPipeline, "decode_latents", decode)
    real_call = AnimationPipeline.__call__

    def call(self, *a, **k):
        seen["call"] = {n: (v.detach().float().cpu().clone() if torch.is_tensor(v) else v) for n, v in k.items()}
        return real_call(self, *a, **k)
    monkeypatch.setattr(AnimationPipeline, "__call__", call)
    if not device_is_gpu:                                       # the pipelines' own .to("cuda") (not an nn.Module.to) on the GPU-less box
        from diffusers import StableDiffusionPipeline
        for cls in (AnimationPipeline, StableDiffusionPipeline):
            real_to = cls.to
            monkeypatch.setattr(cls, "to", lambda self, device, _r=real_to: _r(self, "cpu"))

    out_dir = os.path.join(root, "out")
    argv = [E.REF_SCRIPT, "--config", cfg_path, "--file", sheet, "--pretrained_model_path", root, "--inference_config", E.REF_INFERENCE_CFG,
            "--output_path", out_dir, "--manually_input_image", "--use_fps_condition", "--fps", "2", "--flw_ctrl", "4",
            "--L", str(frames), "--W", str(size), "--H", str(size), "--seed", "1", "--ddp"]
    monkeypatch.setattr(sys, "argv", argv)
    monkeypatch.chdir(root)
    try:
        runpy.run_path(E.REF_SCRIPT, run_name="__main__")
    finally:
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        for k in [k for k in sys.modules if k.split(".")[0] in ("animatediff", "diffusers", "ip_adapter")]:
            del sys.modules[k]
        sys.modules.update(saved_modules)
    return seen, out_dir, dict(steps=steps, size=size, frames=frames)


def _check(seen, out_dir, run, tol):
    from oracle import functional as Fn
    # the script wrote its GIFs and its config (scripts/inference.py:398-403, 424, 429)
    runs = [d for d in os.listdir(out_dir)]
    assert len(runs) == 1, runs
    savedir = os.path.join(out_dir, runs[0])
    gifs = os.listdir(os.path.join(savedir, "sample"))
    assert gifs == [f"0_{PROMPT}.gif"], gifs
    assert os.path.getsize(os.path.join(savedir, "sample", gifs[0])) > 1000 and os.path.getsize(os.path.join(savedir, "sample.gif")) > 1000
    assert os.path.exists(os.path.join(savedir, "config.yaml"))
    # the pipeline call the script made
    c = seen["call"]
    lat = run["size"] // 8
    assert c["video_length"] == run["frames"] and c["num_inference_steps"] == run["steps"] and c["use_first_frame_mask_condition_concat"] is True
    # (the script's mask is the RGB image's 3 channels on the "frame" axis, scripts/inference.py:361-365; the pipeline takes [:, :, 0:1])
    assert tuple(c["first_image_latents"].shape) == (1, 4, lat, lat) and tuple(c["first_images_mask"].shape) == (1, 1, 3, lat, lat)
    mask = c["first_images_mask"][:, :, 0:1]
    assert set(mask.unique().tolist()) <= {0.0, 1.0} and 0 < mask.sum() < mask.numel()
    # ... and its result against the oracle's denoising loop on the same weights / noise / text states / conditioning
    cfg = Fn.tiny_unet_config()
    with torch.no_grad():
        ref = Fn.denoise(seen["unet_sd"], cfg, Fn.DDIMConfig(), seen["noise"], seen["text"], run["steps"], float(c["guidance_scale"]),
                         c["first_image_latents"], mask, torch.tensor([2]), torch.tensor([4]))
    r = ((seen["final"] - ref).norm() / ref.norm()).item()
    assert r < tol, r


def test_reference_inference_script_runs_unmodified_on_emulator(tmp_path, monkeypatch):
    """CPU box: the whole script on the op emulator (bf16 storage, f32 accumulation - the drop-in's default precision)"""
    seen, out_dir, run = _run_script(tmp_path, monkeypatch, device_is_gpu=False, port=29761)
    _check(seen, out_dir, run, tol=6e-2)


@pytest.mark.gpu
def test_reference_inference_script_runs_unmodified_on_mi355x(tmp_path, monkeypatch):
    """the same on the HIP kernels (only where the reference tree and a GPU are on one box)"""
    seen, out_dir, run = _run_script(tmp_path, monkeypatch, device_is_gpu=True, port=29762)
    _check(seen, out_dir, run, tol=6e-2)


def test_reference_inference_script_in_f16_mode_on_emulator(tmp_path, monkeypatch):
    """FYC_COMPUTE_DTYPE=f16: the unmodified script (which cannot pass constructor arguments) runs every drop-in model in IEEE half - the
    precision class of its own `torch.autocast("cuda")` - and lands ~8x closer to the f32 oracle than the bf16 default's 6e-2 bound"""
    monkeypatch.setenv("FYC_COMPUTE_DTYPE", "f16")
    seen, out_dir, run = _run_script(tmp_path, monkeypatch, device_is_gpu=False, port=29771)
    _check(seen, out_dir, run, tol=1e-2)


@pytest.mark.gpu
def test_reference_inference_script_in_f16_mode_on_mi355x(tmp_path, monkeypatch):
    monkeypatch.setenv("FYC_COMPUTE_DTYPE", "f16")
    seen, out_dir, run = _run_script(tmp_path, monkeypatch, device_is_gpu=True, port=29772)
    _check(seen, out_dir, run, tol=1e-2)


# ---- scripts/inference_org.py and scripts/inference_w_image_cond.py ----------------------------------------------------------------
def _run_t2v_script(script, tmp_path, monkeypatch, device_is_gpu: bool, port: int, extra_args, first_image: bool, cfg_extra=None,
                    own_argv=None):
    """the text-to-video scripts: pipeline call without first-frame conditioning (a 4-channel UNet3D); `first_image`: the script
    also builds a 2-D StableDiffusionPipeline and synthesises a first image with it (inference_w_image_cond.py); `cfg_extra(root, fab)`:
    more keys for the model entry of the prompt config; `own_argv(cfg_path, icfg_path, root, frames, size)`: the whole command line
    for scripts with a different argument set (animate.py)"""
    import followyourclick_amd
    import yaml
    from followyourclick_amd import ops as ops_mod
    monkeypatch.setattr(sys, "path", list(sys.path))
    saved_modules = {k: v for k, v in sys.modules.items() if k.split(".")[0] in ("animatediff", "diffusers", "ip_adapter")}
    if not device_is_gpu:
        from emu_ops import EmuOps
        monkeypatch.setattr(ops_mod, "impl", EmuOps())
        E.alias_cuda_to_cpu(monkeypatch)
    followyourclick_amd.install_dropin(force=True)
    E.install_absent_packages(monkeypatch)
    root = str(tmp_path)
    # an inference YAML for plain text-to-video: the shipped one minus the mask / first-frame concat (4 input channels)
    with open(E.REF_INFERENCE_CFG) as f:
        icfg = yaml.safe_load(f)
    icfg["unet_additional_kwargs"]["use_first_frame_mask_condition_concat"] = False
    icfg_path = os.path.join(root, "inference_t2v.yaml")
    with open(icfg_path, "w") as f:
        yaml.safe_dump(icfg, f)
    fab = E.fabricate_model_dir(root, inference_cfg=icfg_path, wrap_state_dict=False)
    steps, size, frames = 2, 64, 2
    cfg = {"TinyModel": dict(base="", path=fab["unet2d_ckpt"] if first_image else "", motion_module=[fab["motion_ckpt"]], seed=[1], steps=steps,
                             guidance_scale=8.0, lora_alpha=0.8)}
    if cfg_extra is not None:
        cfg["TinyModel"].update(cfg_extra(root, fab))
    cfg_path = os.path.join(root, "prompts.yaml")
    with open(cfg_path, "w") as f:
        yaml.safe_dump(cfg, f)
    prompt_file = os.path.join(root, "prompts.txt")
    with open(prompt_file, "w") as f:
        f.write(PROMPT + "\n")
    for k, v in dict(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)).items():
        monkeypatch.setenv(k, v)

    from animatediff.pipelines.pipeline_animation import AnimationPipeline
    from diffusers import StableDiffusionPipeline
    seen = {}
    real_encode, real_prep, real_decode, real_call = (AnimationPipeline._encode_prompt, AnimationPipeline.prepare_latents,
                                                       AnimationPipeline.decode_latents, AnimationPipeline.__call__)
    monkeypatch.setattr(AnimationPipeline, "_encode_prompt", lambda self, *a, **k: _keep(seen, "text", real_encode(self, *a, **k)))
    monkeypatch.setattr(AnimationPipeline, "prepare_latents", lambda self, *a, **k: _keep(seen, "noise", real_prep(self, *a, **k)))

    def decode(self, latents):
        seen["final"] = latents.detach().float().cpu().clone()
        seen["unet_sd"] = {k: v.detach().float().cpu().clone() for k, v in self.unet.state_dict().items()}
        return real_decode(self, latents)
    monkeypatch.setattr(AnimationPipeline, "decode_latents", decode)

    def call(self, *a, **k):
        seen["call"] = dict(k)
        return real_call(self, *a, **k)
    monkeypatch.setattr(AnimationPipeline, "__call__", call)
    if first_image:
        real_sd_call = StableDiffusionPipeline.__call__

        def sd_call(self, *a, **k):
            seen["sd_call"] = dict(k)
            out = real_sd_call(self, *a, **k)
            seen["first_image_size"] = out.images[0].size
            return out
        monkeypatch.setattr(StableDiffusionPipeline, "__call__", sd_call)
    if not device_is_gpu:
        for cls in (AnimationPipeline, StableDiffusionPipeline):
            real_to = cls.to
            monkeypatch.setattr(cls, "to", lambda self, device, _r=real_to: _r(self, "cpu"))
    out_dir = os.path.join(root, "out")
    argv = [script, "--config", cfg_path, "--prompt", prompt_file, "--pretrained_model_path", root, "--inference_config", icfg_path,
            "--L", str(frames), "--W", str(size), "--H", str(size), "--seed", "1", "--ddp"] + (extra_args(root, out_dir, fab) if extra_args is not None else [])
    if own_argv is not None:
        argv = [script] + own_argv(cfg_path, icfg_path, root, frames, size)
    monkeypatch.setattr(sys, "argv", argv)
    monkeypatch.chdir(root)
    try:
        runpy.run_path(script, run_name="__main__")
    finally:
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        for k in [k for k in sys.modules if k.split(".")[0] in ("animatediff", "diffusers", "ip_adapter")]:
            del sys.modules[k]
        sys.modules.update(saved_modules)
    return seen, root, out_dir, dict(steps=steps, size=size, frames=frames)


def _keep(seen, name, value):
    seen[name] = value.detach().float().cpu().clone()
    return value


def _check_t2v(seen, run, tol, video_scale=0.0):
    from oracle import functional as Fn
    c = seen["call"]
    assert c["video_length"] == run["frames"] and c["num_inference_steps"] == run["steps"] and c["width"] == run["size"]
    assert float(c.get("video_scale", 0.0)) == video_scale
    cfg = Fn.tiny_unet_config(use_first_frame_mask_condition_concat=False)
    with torch.no_grad():
        ref = Fn.denoise(seen["unet_sd"], cfg, Fn.DDIMConfig(), seen["noise"], seen["text"], run["steps"], float(c["guidance_scale"]), video_scale=video_scale)
    r = ((seen["final"] - ref).norm() / ref.norm()).item()
    assert r < tol, r


def _org_args(root, out_dir, fab):
    return ["--output_path", out_dir, "--video_scale", "0.5"]


@pytest.mark.skipif(not os.path.exists(os.path.join(E.REF_ROOT, "scripts", "inference_org.py")), reason="scripts/inference_org.py is not on this box")
def test_reference_inference_org_script_runs_unmodified_on_emulator(tmp_path, monkeypatch):
    """scripts/inference_org.py (the text-to-video driver): prompt file -> PromptDataset + DistributedSampler, plain pipeline call with
    `video_scale` (the per-frame unconditional pass and three-way guidance), GIFs + config written - on the op emulator"""
    seen, root, out_dir, run = _run_t2v_script(os.path.join(E.REF_ROOT, "scripts", "inference_org.py"), tmp_path, monkeypatch, False, 29763, _org_args, False)
    _check_t2v(seen, run, tol=6e-2, video_scale=0.5)
    runs = os.listdir(out_dir)
    assert len(runs) == 1 and os.path.getsize(os.path.join(out_dir, runs[0], "sample", f"0_{PROMPT}.gif")) > 1000
    assert os.path.exists(os.path.join(out_dir, runs[0], "config.yaml")) and os.path.exists(os.path.join(out_dir, runs[0], "prompts.txt"))


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(E.REF_ROOT, "scripts", "inference_org.py")), reason="scripts/inference_org.py is not on this box")
def test_reference_inference_org_script_runs_unmodified_on_mi355x(tmp_path, monkeypatch):
    seen, root, out_dir, run = _run_t2v_script(os.path.join(E.REF_ROOT, "scripts", "inference_org.py"), tmp_path, monkeypatch, True, 29764, _org_args, False)
    _check_t2v(seen, run, tol=6e-2, video_scale=0.5)


def _image_cond_args(root, out_dir, fab):
    return ["--use_local_image", "--image_pretrained_model_path", fab["clip_vision_dir"]]


def _check_image_cond(seen, root, run, tol):
    _check_t2v(seen, run, tol=tol)
    assert seen["sd_call"]["height"] == 448 and seen["sd_call"]["width"] == 768 and seen["sd_call"]["num_inference_steps"] == 50
    assert seen["first_image_size"] == (768, 448)
    assert seen["call"]["use_ip_cross_attention"] is False and tuple(seen["call"]["condition_images"].shape) == (1, 3, 224, 224)
    runs = os.listdir(os.path.join(root, "samples"))
    assert len(runs) == 1 and os.path.getsize(os.path.join(root, "samples", runs[0], "sample", f"0_{PROMPT}.gif")) > 1000


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(E.REF_ROOT, "scripts", "inference_w_image_cond.py")), reason="scripts/inference_w_image_cond.py is not on this box")
def test_reference_inference_w_image_cond_script_runs_unmodified_on_mi355x(tmp_path, monkeypatch):
    script = os.path.join(E.REF_ROOT, "scripts", "inference_w_image_cond.py")
    seen, root, out_dir, run = _run_t2v_script(script, tmp_path, monkeypatch, True, 29766, _image_cond_args, True)
    _check_image_cond(seen, root, run, tol=6e-2)


@pytest.mark.skipif(os.environ.get("FYC_SLOW_TESTS") != "1", reason="6 minutes on the op emulator (the script hard-codes a 50-step 768x448 first image): FYC_SLOW_TESTS=1; "
                                                                      "the `gpu` variant above runs it on every GPU box")
@pytest.mark.skipif(not os.path.exists(os.path.join(E.REF_ROOT, "scripts", "inference_w_image_cond.py")), reason="scripts/inference_w_image_cond.py is not on this box")
def test_reference_inference_w_image_cond_script_runs_unmodified_on_emulator(tmp_path, monkeypatch):
    """scripts/inference_w_image_cond.py without --use_ip: the 2-D UNet gets its weights from a `.ckpt` (with the `module.` prefix
    the script strips), StableDiffusionPipeline synthesises the first image at the script's hard-coded 768x448 / 50 steps,
    CLIPImageProcessor + CLIPVisionModelWithProjection are constructed and the image is pre-processed, then the video pipeline runs
    (its `condition_images` are ignored without `use_ip_cross_attention`, as in the reference).  `samples/` is the script's own
    hard-coded output directory (relative to the working directory)."""
    script = os.path.join(E.REF_ROOT, "scripts", "inference_w_image_cond.py")
    seen, root, out_dir, run = _run_t2v_script(script, tmp_path, monkeypatch, False, 29765, _image_cond_args, True)
    _check_image_cond(seen, root, run, tol=6e-2)


# ---- scripts/animate.py and scripts/inference_w_camera_lora.py -------------------------------------------------------------------
ANIMATE = os.path.join(E.REF_ROOT, "scripts", "animate.py")
CAMERA_LORA = os.path.join(E.REF_ROOT, "scripts", "inference_w_camera_lora.py")


def _animate_cfg(root, fab):
    # the upstream AnimateDiff prompt-config layout animate.py reads: prompts, negative prompts and seeds live in the YAML (:112-119)
    return dict(prompt=[PROMPT], n_prompt=["blurry"], seed=[7])


def _animate_argv(cfg_path, icfg_path, root, frames, size):
    return ["--config", cfg_path, "--pretrained_model_path", root, "--inference_config", icfg_path, "--L", str(frames), "--W", str(size), "--H", str(size)]


def _check_animate(seen, root, run, tol):
    _check_t2v(seen, run, tol=tol)
    assert seen["call"]["negative_prompt"] == "blurry" and float(seen["call"]["guidance_scale"]) == 8.0
    runs = os.listdir(os.path.join(root, "samples"))
    assert len(runs) == 1, runs
    savedir = os.path.join(root, "samples", runs[0])
    name = "0-" + "-".join(PROMPT.split(" ")[:10]) + ".gif"                                      # scripts/animate.py:144-145
    assert os.listdir(os.path.join(savedir, "sample")) == [name]
    assert os.path.getsize(os.path.join(savedir, "sample", name)) > 1000 and os.path.getsize(os.path.join(savedir, "sample.gif")) > 1000
    import yaml
    with open(os.path.join(savedir, "config.yaml")) as f:
        assert yaml.safe_load(f)["TinyModel"]["random_seed"] == [7]                              # :125-128: manual_seed(7) -> initial_seed()


@pytest.mark.skipif(not os.path.exists(ANIMATE), reason="scripts/animate.py is not on this box")
def test_reference_animate_script_runs_unmodified_on_emulator(tmp_path, monkeypatch):
    """scripts/animate.py (the upstream AnimateDiff driver the repository keeps): no DDP, prompts / negative prompts / seeds from the
    YAML, bare motion-module checkpoint through strict=False (:66-76), per-prompt `torch.manual_seed`, pipeline call under
    `torch.autocast("cuda")`, one GIF per prompt + the grid + config.yaml with the seeds - on the op emulator"""
    seen, root, out_dir, run = _run_t2v_script(ANIMATE, tmp_path, monkeypatch, False, 29767, None, False, cfg_extra=_animate_cfg, own_argv=_animate_argv)
    _check_animate(seen, root, run, tol=6e-2)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(ANIMATE), reason="scripts/animate.py is not on this box")
def test_reference_animate_script_runs_unmodified_on_mi355x(tmp_path, monkeypatch):
    seen, root, out_dir, run = _run_t2v_script(ANIMATE, tmp_path, monkeypatch, True, 29768, None, False, cfg_extra=_animate_cfg, own_argv=_animate_argv)
    _check_animate(seen, root, run, tol=6e-2)


def _fabricate_motion_lora(root, fab, rank=4, seed=3):
    """a motion-LoRA checkpoint in the layout `convert_motion_lora_ckpt_to_diffusers` consumes (reference
    animatediff/utils/convert_lora_safetensor_to_diffusers.py:26-51): `<module path>.processor.<to_q|to_k|to_v|to_out>_lora.<down|up>.weight`
    for the attention projections of every temporal transformer block, saved from a DDP run (`module.` prefix, `state_dict` wrapper)"""
    sd = torch.load(fab["motion_ckpt"], map_location="cpu")
    g = torch.Generator().manual_seed(seed)
    lora = {}
    for k, v in sd.items():
        for proj in ("to_q", "to_k", "to_v", "to_out.0"):
            if "motion_modules" in k and "attention_blocks" in k and k.endswith(proj + ".weight"):
                stem = k[:-len(proj + ".weight")]
                name = proj.split(".")[0]
                lora[f"module.{stem}processor.{name}_lora.down.weight"] = torch.randn(rank, v.shape[1], generator=g) / v.shape[1] ** 0.5
                lora[f"module.{stem}processor.{name}_lora.up.weight"] = torch.randn(v.shape[0], rank, generator=g) * 0.3 / rank ** 0.5
    assert lora, "no temporal attention projections found in the fabricated motion module"
    path = os.path.join(root, "motion_lora.ckpt")
    torch.save({"state_dict": lora}, path)
    return path, lora, sd


def _camera_cfg(root, fab):
    path, lora, base = _fabricate_motion_lora(root, fab)
    fab["motion_lora"], fab["motion_base"] = lora, base
    return dict(motion_module_lora_path=path)


def _camera_args(root, out_dir, fab):
    return ["--video_scale", "0.5"]


def _check_camera_lora(seen, root, run, fab, tol):
    # the merged temporal projections the pipeline ran with: W + 1.0 * up @ down (alpha is hard-coded, scripts/inference_w_camera_lora.py:216)
    lora, base = fab["motion_lora"], fab["motion_base"]
    n = 0
    for k, down in lora.items():
        if ".down." not in k:
            continue
        up = lora[k.replace(".down.", ".up.")]
        key = k[len("module."):].replace("processor.", "").replace("_lora", "").replace("down.", "").replace("to_out.", "to_out.0.")
        want = base[key].float() + up @ down
        assert torch.allclose(seen["unet_sd"][key], want, atol=1e-5), key
        assert (seen["unet_sd"][key] - base[key].float()).abs().max() > 1e-3, key
        n += 1
    assert n >= 8, n
    # the negative prompt is the FIRST CHARACTER of NEG_PROMPT: `list(NEG_PROMPT) * len(...)` (:249), kept as the script has it
    assert seen["call"]["negative_prompt"] == "l"
    _check_t2v(seen, run, tol=tol, video_scale=0.5)
    runs = os.listdir(os.path.join(root, "samples"))
    assert len(runs) == 1 and "_vs_0.5_" in runs[0], runs
    assert os.path.getsize(os.path.join(root, "samples", runs[0], "sample", f"0_{PROMPT}.gif")) > 1000
    assert os.path.exists(os.path.join(root, "samples", runs[0], "config.yaml")) and os.path.exists(os.path.join(root, "samples", runs[0], "prompts.txt"))


def _run_camera_lora(tmp_path, monkeypatch, device_is_gpu, port):
    import pdb
    monkeypatch.setattr(pdb, "set_trace", lambda *a, **k: None)     # the script stops in a debugger at :221; an unattended run continues
    keep = {}

    def cfg_extra(root, fab):
        keep["fab"] = fab
        return _camera_cfg(root, fab)
    seen, root, out_dir, run = _run_t2v_script(CAMERA_LORA, tmp_path, monkeypatch, device_is_gpu, port, _camera_args, False, cfg_extra=cfg_extra)
    return seen, root, run, keep["fab"]


@pytest.mark.skipif(not os.path.exists(CAMERA_LORA), reason="scripts/inference_w_camera_lora.py is not on this box")
def test_reference_inference_w_camera_lora_script_runs_unmodified_on_emulator(tmp_path, monkeypatch):
    """scripts/inference_w_camera_lora.py: bare motion-module checkpoint (strict=False, :155-163), a motion LoRA merged into the temporal
    attention projections by `convert_motion_lora_ckpt_to_diffusers` (:215-219, the engine repacks its weights afterwards),
    `PromptDataset` tuples through DistributedSampler + DataLoader, pipeline call with `video_scale` - on the op emulator.  Imports
    `load_weights` (:23), which therefore has to exist in the drop-in."""
    seen, root, run, fab = _run_camera_lora(tmp_path, monkeypatch, False, 29769)
    _check_camera_lora(seen, root, run, fab, tol=6e-2)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(CAMERA_LORA), reason="scripts/inference_w_camera_lora.py is not on this box")
def test_reference_inference_w_camera_lora_script_runs_unmodified_on_mi355x(tmp_path, monkeypatch):
    seen, root, run, fab = _run_camera_lora(tmp_path, monkeypatch, True, 29770)
    _check_camera_lora(seen, root, run, fab, tol=6e-2)
