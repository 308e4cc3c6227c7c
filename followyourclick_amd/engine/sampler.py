"""The DDIM sampling loop on the engine: latents in -> latents out.

Restates the hot loop of AnimationPipeline.__call__ (reference
animatediff/pipelines/pipeline_animation.py:686-773) for the mask + first-frame concat conditioning
with classifier-free guidance, and - for a UNet built without the concat conditioning - the loop of
StableDiffusionPipeline.__call__ (reference diffusers/pipelines/stable_diffusion/pipeline_stable_diffusion.py:522-541).  Per step: build the 9-channel channels-last input (one kernel instead
of 3 zeros_like + 2 cat), UNet forward on the CFG pair, guidance + DDIM update (one kernel, no host
sync).  Everything that is constant over the loop (text/IP K/V, all time embeddings, DDIM
coefficients) is prepared once per clip.
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Sequence

import torch

from .config import DDIMConfig
from .scheduler import DDIMTables
from .unet3d import UNet3DEngine
from .weights import pad_channels

Tensor = torch.Tensor


class DDIMSampler:
    def __init__(self, unet: UNet3DEngine, ddim: DDIMConfig):
        self.unet = unet
        self.tables = DDIMTables(ddim)
        self.clip_sample = bool(ddim.clip_sample)

    @torch.no_grad()
    def prepare(self, text_embeddings: Tensor, num_steps: int, batch: int, guidance_scale: float,
                fps: Optional[Sequence[float]] = None, flow: Optional[Sequence[float]] = None,
                ip_tokens: Optional[Tensor] = None, video_scale: float = 0.0, frames: int = 0,
                first_frame_condition: bool = False, eta: float = 0.0, camera: Optional[Sequence[float]] = None) -> dict:
        """text_embeddings: (cfg*batch, 77, D) with the unconditional half first (reference :397)."""
        u = self.unet
        cfg_on = guidance_scale > 1.0
        beff = batch * (2 if cfg_on else 1)
        if text_embeddings.shape[0] != beff:
            raise ValueError(f"text_embeddings batch {text_embeddings.shape[0]} != {beff}")
        ts = self.tables.timesteps(num_steps)

        def dup(v):
            if v is None:
                return None
            v = [float(x) for x in v]
            if len(v) == 1:
                v = v * batch
            return v * 2 if cfg_on else v

        extra = {}
        if video_scale > 0 and cfg_on:
            # `video_scale` (reference :738-760): every frame is also denoised on its own, as a one-frame clip.  The reference
            # builds that call's text batch as cat([text_embeddings] * f).chunk(2)[0], i.e. single-frame clip j gets
            # text_embeddings[j mod 2B] - for B = 1 the frames alternate between the negative and the positive prompt - and
            # passes no fps / flow conditioning; both are reproduced as they are.
            nf = batch * frames
            idx = torch.arange(nf) % (2 * batch)
            u.prepare_context(text_embeddings[idx], None)
            _, temb_s = u.prepare_time_embeddings(ts.tolist(), None, None, nf)
            extra = dict(ctx_single=u.ctx_cache, temb_single=temb_s.reshape(num_steps, nf, -1), video_scale=float(video_scale))
        if first_frame_condition:
            # `use_first_frame_condition` (reference :691-692; unet.py:523-524): frame 0 of the latents is pinned to the clean
            # first-frame latents and every ResNet gives that frame the embedding of timestep 0
            if fps is not None:
                raise ValueError("use_first_frame_condition cannot be combined with fps conditioning (the reference's embeddings "
                                 "have mismatching batch sizes there)")
            _, temb0 = u.prepare_time_embeddings([0], None, None, 1)
            extra["temb_first"] = temb0
        u.prepare_context(text_embeddings, ip_tokens)
        emb, temb = u.prepare_time_embeddings(ts.tolist(), dup(fps), dup(flow), beff, camera=dup(camera))
        coef_host = self.tables.coefficient_table(num_steps, eta)
        coef = coef_host.to(u.device)
        extra["sigma"] = coef_host[:, 4].tolist()           # eta > 0: per-step std of the added noise (0 for eta = 0)
        return dict(timesteps=ts, temb=temb.reshape(num_steps, beff, -1), coef=coef, cfg=cfg_on, beff=beff,
                    guidance=float(guidance_scale), steps=num_steps, batch=batch, ctx_main=u.ctx_cache, **extra)

    @torch.no_grad()
    def step(self, st: dict, i: int, latents: Tensor, first_image_latents: Optional[Tensor], mask: Optional[Tensor],
             variance_noise: Optional[Tensor] = None, use_clipped_model_output: bool = False) -> None:
        """one DDIM step, latents (B,4,F,h,w) f32 updated in place.  variance_noise (latents' shape, f32): the noise of a stochastic
        step (eta > 0, reference scheduling_ddim.py:346-363), scaled by the step's sigma from `prepare(eta=...)`."""
        u, o = self.unet, self.unet.ops
        B, CL, F, H, W = latents.shape
        cp = pad_channels(u.cfg.conv_in_channels)
        dupn = 2 if st["cfg"] else 1
        x = u.new(dupn * B * F * H * W, cp)
        if "temb_first" in st:
            latents[:, :, 0] = first_image_latents.reshape(B, CL, H, W)
        # (same precedence as UNet3DConfig.conv_in_channels / pack_unet and the reference's constructor, unet.py:114-126: the 8-channel
        # concat wins when both flags are set - the mask flag defaults to True)
        if u.cfg.use_first_frame_condition_concat:
            # the reference's UNet concatenates `reference_images_latent` (the clean first-frame latents) beside EVERY frame's latents
            # (unet.py:580-586; pipeline_animation.py:705-706 passes the plain latents); conv_in's `/ 2` lives in its packed weights
            o.unet_input(latents, None, first_image_latents, x, B=B, F=F, HW=H * W, c_latent=CL, c_pad=cp, cfg_dup=dupn, mode=1)
        elif u.cfg.use_first_frame_mask_condition_concat:
            o.unet_input(latents, mask, first_image_latents, x, B=B, F=F, HW=H * W, c_latent=CL, c_pad=cp, cfg_dup=dupn,
                         mask_frames=1)
        else:
            # plain latents (the 2-D Stable Diffusion first-image path, reference pipeline_stable_diffusion.py:527-528)
            n = B * F * H * W
            frames = latents.permute(0, 2, 1, 3, 4).reshape(B * F, CL, H * W).contiguous()
            for d in range(dupn):
                o.nchw_to_nhwc(frames, x[d * n:(d + 1) * n], N=B * F, C_=CL, HW=H * W, c_pad=cp, scale=1.0)
        pred = u.forward(x, st["temb"][i], dupn * B, F, H, W, temb_first=st.get("temb_first"))
        single = None
        if "ctx_single" in st:      # per-frame unconditional pass: the first (unconditional) half of x as B*F one-frame clips
            u.ctx_cache = st["ctx_single"]
            single = u.forward(x[: B * F * H * W], st["temb_single"][i], B * F, 1, H, W)
            u.ctx_cache = st["ctx_main"]
        o.cfg_ddim_step(pred, latents, st["coef"][i], B=B, F=F, HW=H * W, c_latent=CL, ld=pred.shape[1], cfg=st["cfg"],
                        guidance=st["guidance"], pred_type=self.tables.pred_type, clip_sample=self.clip_sample,
                        pred_single=single, video_scale=st.get("video_scale", 0.0),
                        variance_noise=variance_noise, sigma=st["sigma"][i] if variance_noise is not None else 0.0,
                        clipped_model_output=use_clipped_model_output)

    @torch.no_grad()
    def sample(self, latents: Tensor, text_embeddings: Tensor, num_steps: int, guidance_scale: float,
               first_image_latents: Optional[Tensor] = None, first_images_mask: Optional[Tensor] = None,
               fps: Optional[Sequence[float]] = None, flow: Optional[Sequence[float]] = None,
               ip_tokens: Optional[Tensor] = None, callback: Optional[Callable] = None, callback_steps: int = 1,
               use_graph: Optional[bool] = None, video_scale: float = 0.0, first_frame_condition: bool = False,
               eta: float = 0.0, generator=None, use_clipped_model_output: bool = False,
               camera: Optional[Sequence[float]] = None) -> Tensor:
        """use_graph: replay steps 1..n-1 from one captured hipGraph (None = the FYC_HIPGRAPH environment switch, default off).
        Measured on MI355X it buys nothing: at cfg2 the loop is GPU-bound (56 ms of kernels per step) and even the 2-D
        one-frame case (13.7 ms / step, ~700 small kernels) is bound by the kernels' own execution, not by launch overhead
        (profiles/r01_front_end_rows.json).  Kept as an option for hosts with slow Python."""
        u = self.unet
        latents = latents.to(device=u.device, dtype=torch.float32).contiguous().clone()
        B, CL, F, H, W = latents.shape
        first = None if first_image_latents is None else first_image_latents.to(u.device, torch.float32).reshape(B, CL, H * W).contiguous()
        mask = None
        if first_images_mask is not None:
            # mask for ALL frames = clamp(first_images_mask[:, :, 0:1]) (reference :632-635)
            mask = first_images_mask.to(u.device, torch.float32)[:, :, 0].reshape(B, 1, H * W).contiguous()
        st = self.prepare(text_embeddings, num_steps, B, guidance_scale, fps, flow, ip_tokens, video_scale=video_scale, frames=F,
                          first_frame_condition=first_frame_condition, eta=eta, camera=camera)
        ts = st["timesteps"].tolist()
        if use_graph is None:
            use_graph = os.environ.get("FYC_HIPGRAPH", "0") == "1"
        if eta > 0:
            use_graph = False             # the noise of every step comes from the generator: nothing to replay
        if not (use_graph and latents.is_cuda and num_steps >= 3):
            for i, t in enumerate(ts):
                noise = None
                if eta > 0:
                    # drawn exactly as the reference's scheduler draws it (scheduling_ddim.py:355-361): on the latents' device, in
                    # their dtype, from the caller's generator (or the device's global one)
                    noise = torch.randn(latents.shape, generator=generator, device=latents.device, dtype=latents.dtype)
                self.step(st, i, latents, first, mask, variance_noise=noise, use_clipped_model_output=use_clipped_model_output)
                if callback is not None and i % callback_steps == 0:
                    callback(i, t, latents)
            return latents
        # Step 0 runs eagerly (it is also the warm-up: lazily set kernel attributes, allocator pools); step 1 is captured
        # reading its time-embedding row and DDIM coefficients from two fixed buffers, which are refreshed before every replay.
        self.step(st, 0, latents, first, mask, use_clipped_model_output=use_clipped_model_output)
        if callback is not None:
            callback(0, ts[0], latents)
        temb_cur, coef_cur = torch.empty_like(st["temb"][0]), torch.empty_like(st["coef"][0])
        gst = dict(st, temb=temb_cur[None], coef=coef_cur[None])
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=u.device)
        side.wait_stream(torch.cuda.current_stream(u.device))
        with torch.cuda.graph(graph, stream=side):
            self.step(gst, 0, latents, first, mask, use_clipped_model_output=use_clipped_model_output)
        for i in range(1, num_steps):
            temb_cur.copy_(st["temb"][i])
            coef_cur.copy_(st["coef"][i])
            graph.replay()
            if callback is not None and i % callback_steps == 0:
                callback(i, ts[i], latents)
        return latents
