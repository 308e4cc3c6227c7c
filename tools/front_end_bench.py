"""Timings of the rows around the sampling loop at full size on one MI355X, random weights:
CLIP ViT-L/14 text encoder, CLIP ViT-H/14 vision tower, ImageProjModel / Resampler, VAE encode of the first frame,
2-D Stable Diffusion first-image synthesis (50 DDIM steps @512x512 + decode).  Prints one JSON object."""
import json
import time

import torch

import followyourclick_amd

followyourclick_amd.install_dropin(force=True)
from diffusers import AutoencoderKL, DDIMScheduler, StableDiffusionPipeline, UNet2DConditionModel  # noqa: E402
from followyourclick_amd.encoders import ClipTextHip, ClipVisionHip  # noqa: E402
from ip_adapter import ImageProjModel, Resampler  # noqa: E402

dev = "cuda"
res = {}


def timed(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


from transformers import CLIPTextConfig, CLIPTextModel, CLIPVisionConfig, CLIPVisionModelWithProjection  # noqa: E402

txt = ClipTextHip.from_transformers(CLIPTextModel(CLIPTextConfig())).to(dev)           # ViT-L/14 text tower: 12 x 768
ids = torch.randint(0, 49408, (2, 77), device=dev)
res["clip_text_ms"] = timed(lambda: txt(ids)[0])
vcfg = CLIPVisionConfig(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=16, image_size=224,
                        patch_size=14, projection_dim=1024, hidden_act="gelu")
vis = ClipVisionHip.from_transformers(CLIPVisionModelWithProjection(vcfg)).to(dev)       # ViT-H/14
px = torch.randn(2, 3, 224, 224, device=dev)
res["clip_vision_plus_ms"] = timed(lambda: vis(px, output_hidden_states=True).hidden_states[-2])
res["clip_vision_embeds_ms"] = timed(lambda: vis(px[:1]).image_embeds)
feats = vis(px, output_hidden_states=True).hidden_states[-2]
rs = Resampler(dim=768, depth=4, dim_head=64, heads=12, num_queries=16, embedding_dim=1280, output_dim=768, ff_mult=4).to(dev)
res["resampler_ms"] = timed(lambda: rs(feats))
pj = ImageProjModel(cross_attention_dim=768, clip_embeddings_dim=1024, clip_extra_context_tokens=4).to(dev)
emb = torch.randn(2, 1024, device=dev)
res["image_proj_ms"] = timed(lambda: pj(emb))

vae = AutoencoderKL(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4).to(dev)   # SD-1.5 VAE
img = torch.rand(1, 3, 512, 512, device=dev) * 2 - 1
res["vae_encode_512_ms"] = timed(lambda: vae.encode(img).latent_dist.mean)

unet = UNet2DConditionModel(sample_size=64, cross_attention_dim=768).to(dev)


class Tok:
    model_max_length = 77

    def __call__(self, texts, **kw):
        n = len(texts) if isinstance(texts, list) else 1
        return type("O", (), {"input_ids": torch.randint(0, 49408, (n, 77)), "attention_mask": torch.ones(n, 77)})()


sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False, steps_offset=1)
pipe = StableDiffusionPipeline.from_pretrained("-", unet=unet, vae=vae, tokenizer=Tok(), text_encoder=txt, scheduler=sched, safety_checker=None).to(dev)
pipe.progress_bar = lambda iterable=None, total=None: None
res["sd2d_50steps_512_ms"] = timed(lambda: pipe("a photo", height=512, width=512, num_inference_steps=50, guidance_scale=8.0, output_type="np"), n=2, warm=1)
res["sd2d_ms_per_step"] = res["sd2d_50steps_512_ms"] / 50
print(json.dumps({k: round(v, 3) for k, v in res.items()}))
