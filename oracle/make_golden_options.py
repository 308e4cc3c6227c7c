"""Golden vectors for the reference options OUTSIDE the shipped YAMLs that round 5 implements (round-4 review, "missing" item 5), by
executing the REAL reference at the tiny widths (container only; TEST INFRASTRUCTURE):

    python -m oracle.make_golden_options          # ~1 min -> tests/golden/options_tiny.npz

  * DDIMScheduler.step with eta > 0 (explicit `variance_noise`), with `use_clipped_model_output` + clip_sample, for v_prediction and
    epsilon (diffusers/schedulers/scheduling_ddim.py:336-365);
  * AnimationPipeline.__call__ with eta = 0.6 and a seeded generator, 3 DDIM steps (pipeline_animation.py:672, 763-767);
  * UNet3DConditionModel.forward with use_camera_motion_condition (animatediff/models/unet.py:134-137, 498-508, 538-544);
  * UNet3DConditionModel.forward with use_first_frame_condition_concat + reference_images_latent (:580-590).
Weights / inputs are re-derived from seeds (oracle/weights.py); only outputs are stored.
"""
import os

import numpy as np
import torch

from . import functional as Fn
from . import refshim
from . import weights as W
from .make_golden import OUT, ref_unet
from .make_golden_full import SKW, _pipeline, log


def main():
    refshim.install()
    from diffusers.schedulers.scheduling_ddim import DDIMScheduler
    d = {}
    # ---- scheduler ------------------------------------------------------------------------------------------------------------
    g = torch.Generator().manual_seed(11)
    x, v, noise = (torch.randn(1, 4, 3, 8, 8, generator=g) for _ in range(3))
    d["sched_sample"], d["sched_model_output"], d["sched_noise"] = x.numpy(), v.numpy(), noise.numpy()
    for tag, kw, step_kw in (("v_eta", dict(), dict(eta=0.7)),
                             ("v_eta_clipped", dict(clip_sample=True), dict(eta=0.7, use_clipped_model_output=True)),
                             ("eps_eta", dict(prediction_type="epsilon"), dict(eta=0.35)),
                             ("v_clipped_eta0", dict(clip_sample=True), dict(use_clipped_model_output=True))):
        sch = DDIMScheduler(**dict(SKW, **kw))
        sch.set_timesteps(25)
        for t in (961, 481, 41):
            out = sch.step(v, t, x, variance_noise=noise if step_kw.get("eta", 0) > 0 else None, **step_kw).prev_sample
            d[f"sched_{tag}_{t}"] = out.numpy()
            log("sched", tag, t, float(out.std()))
    # ---- pipeline with eta > 0 --------------------------------------------------------------------------------------------------
    cfg = Fn.tiny_unet_config()
    unet = ref_unet(cfg).eval()
    unet.load_state_dict(W.make_weights(W.unet_state_shapes(cfg), seed=0), strict=True)
    pipe = _pipeline(unet, cfg)
    frames, lat, steps = 3, 8, 3
    pipe.decode_latents = lambda latents: np.zeros((1, 3, frames, 8, 8), dtype=np.float32)
    inp = W.seeded_inputs(cfg, 1, frames, lat, lat, seed=71)
    with torch.no_grad():
        text_emb = pipe._encode_prompt(["a corgi waving its tail"], "cpu", 1, True, ["blurry"])
    traj = {}
    pipe("a corgi waving its tail", video_length=frames, height=lat * 8, width=lat * 8, num_inference_steps=steps, guidance_scale=8.0,
         negative_prompt="blurry", latents=inp["latents"].clone(), first_image_latents=inp["first_image_latents"],
         first_images_mask=inp["first_images_mask"], use_first_frame_mask_condition_concat=True, use_fps_condition=True,
         fps_tensor=torch.tensor([2]), flow_control=torch.tensor([4]), eta=0.6, generator=torch.Generator().manual_seed(123),
         callback=lambda i, t, l: traj.__setitem__(i, l.clone().float()), callback_steps=1, output_type="latent")
    d["pipe_eta_text_embeddings"] = text_emb.numpy()
    for i in range(steps):
        d[f"pipe_eta_step{i}"] = traj[i].numpy()
        log("pipeline eta=0.6 step", i, float(traj[i].std()))
    d.update(pipe_eta=np.float64(0.6), pipe_eta_generator_seed=np.int64(123), pipe_eta_input_seed=np.int64(71), pipe_eta_frames=np.int64(frames),
             pipe_eta_lat=np.int64(lat), pipe_eta_steps=np.int64(steps))
    # ---- UNet forward options -----------------------------------------------------------------------------------------------------
    fps, flow = torch.tensor([2, 2]), torch.tensor([4, 4])
    # (a) camera-motion embedding
    ccfg = Fn.tiny_unet_config(use_camera_motion_condition=True)
    cu = ref_unet(ccfg).eval()
    cu.load_state_dict(W.make_weights(W.unet_state_shapes(ccfg), seed=0), strict=True)
    inp = W.seeded_inputs(ccfg, 1, 2, 8, 8, seed=72)
    x9 = torch.cat([Fn.build_model_input(inp["latents"], inp["first_image_latents"], inp["first_images_mask"])] * 2)
    cam = torch.tensor([3, 3])
    with torch.no_grad():
        y = cu(x9, torch.tensor(500), inp["text"], use_fps_condition=True, fps_tensor=fps, flow_control=flow,
               use_camera_motion_condition=True, camera_movement_type_tensor=cam).sample
    d.update(camera_out=y.numpy(), camera_type=cam.numpy(), camera_input_seed=np.int64(72), camera_timestep=np.int64(500))
    log("camera forward", float(y.std()))
    # (b) first-frame concat (8 input channels, conv_in output halved)
    kcfg = Fn.tiny_unet_config(use_first_frame_condition_concat=True, use_first_frame_mask_condition_concat=False)
    ku = ref_unet(kcfg).eval()
    ku.load_state_dict(W.make_weights(W.unet_state_shapes(kcfg), seed=0), strict=True)
    inp = W.seeded_inputs(kcfg, 1, 2, 8, 8, seed=73)
    lat2 = torch.cat([inp["latents"]] * 2)
    with torch.no_grad():
        y = ku(lat2, torch.tensor(500), inp["text"], use_fps_condition=True, fps_tensor=fps, flow_control=flow,
               use_first_frame_condition_concat=True, reference_images_latent=torch.cat([inp["first_image_latents"]] * 2)).sample
    d.update(concat_out=y.numpy(), concat_input_seed=np.int64(73), concat_timestep=np.int64(500))
    log("first-frame-concat forward", float(y.std()))
    # schema of the two option models (state-dict keys and shapes, in order)
    import json
    with torch.device("meta"):
        for name, c in (("schema_unet_tiny_camera.json", ccfg), ("schema_unet_tiny_concat.json", kcfg)):
            with open(os.path.join(OUT, name), "w") as f:
                json.dump({k: list(v.shape) for k, v in ref_unet(c).state_dict().items()}, f, indent=0)
    np.savez_compressed(os.path.join(OUT, "options_tiny.npz"), **d)
    print("options_tiny.npz", os.path.getsize(os.path.join(OUT, "options_tiny.npz")))


if __name__ == "__main__":
    main()
