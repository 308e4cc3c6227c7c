"""Generate tests/golden/enc_*.npz: the conditioning encoders executed by their REAL implementations (container only):
transformers 5.15.0 `CLIPTextModel` / `CLIPVisionModelWithProjection` (third-party, what the reference calls) and the
reference's own `ip_adapter.resampler.Resampler` / `ip_adapter.my_ip_adapter.ImageProjModel`.

Run:  python -m oracle.make_golden_encoders
Weights are re-derived from seeds (oracle/encoders.py::make_encoder_weights), inputs and outputs are stored.
"""
import os

import numpy as np
import torch

from . import encoders as E
from . import refshim
from .make_golden import OUT


def main():
    import transformers
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPVisionConfig, CLIPVisionModelWithProjection
    tc, vc, rc = E.TINY_TEXT, E.TINY_VISION, E.TINY_RESAMPLER
    txt = CLIPTextModel(CLIPTextConfig(vocab_size=tc.vocab_size, hidden_size=tc.hidden_size, intermediate_size=tc.intermediate_size,
                                       num_hidden_layers=tc.num_hidden_layers, num_attention_heads=tc.num_attention_heads,
                                       max_position_embeddings=tc.max_position_embeddings, hidden_act=tc.hidden_act,
                                       bos_token_id=0, eos_token_id=2, pad_token_id=1)).eval()
    sd_t = E.make_encoder_weights(E.clip_text_shapes(tc), seed=51)
    txt.load_state_dict(sd_t, strict=True)
    ids = torch.randint(3, tc.vocab_size, (2, 77), generator=torch.Generator().manual_seed(52))
    with torch.no_grad():
        t_out = txt(ids, attention_mask=None)[0]
    np.savez_compressed(os.path.join(OUT, "enc_clip_text.npz"), input_ids=ids.numpy(), last_hidden_state=t_out.numpy(),
                        weight_seed=np.int64(51), transformers_version=np.array(transformers.__version__))

    vis = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=vc.hidden_size, intermediate_size=vc.intermediate_size,
                                                         num_hidden_layers=vc.num_hidden_layers, num_attention_heads=vc.num_attention_heads,
                                                         image_size=vc.image_size, patch_size=vc.patch_size,
                                                         projection_dim=vc.projection_dim, hidden_act=vc.hidden_act)).eval()
    sd_v = E.make_encoder_weights(E.clip_vision_shapes(vc), seed=53)
    vis.load_state_dict(sd_v, strict=True)
    img = torch.randn(2, 3, vc.image_size, vc.image_size, generator=torch.Generator().manual_seed(54))
    with torch.no_grad():
        v_out = vis(img, output_hidden_states=True)
    np.savez_compressed(os.path.join(OUT, "enc_clip_vision.npz"), pixel_values=img.numpy(), image_embeds=v_out.image_embeds.numpy(),
                        penultimate=v_out.hidden_states[-2].numpy(), last=v_out.hidden_states[-1].numpy(),
                        n_hidden_states=np.int64(len(v_out.hidden_states)), weight_seed=np.int64(53))

    refshim.install()
    from ip_adapter.my_ip_adapter import ImageProjModel
    from ip_adapter.resampler import Resampler
    proj = ImageProjModel(cross_attention_dim=64, clip_embeddings_dim=vc.projection_dim, clip_extra_context_tokens=4).eval()
    sd_p = E.make_encoder_weights(E.image_proj_shapes(vc.projection_dim, 64, 4), seed=55)
    proj.load_state_dict(sd_p, strict=True)
    res = Resampler(dim=rc.dim, depth=rc.depth, dim_head=rc.dim_head, heads=rc.heads, num_queries=rc.num_queries,
                    embedding_dim=rc.embedding_dim, output_dim=rc.output_dim, ff_mult=rc.ff_mult).eval()
    sd_r = E.make_encoder_weights(E.resampler_shapes(rc), seed=56)
    res.load_state_dict(sd_r, strict=True)
    with torch.no_grad():
        p_out = proj(v_out.image_embeds)
        p_zero = proj(torch.zeros_like(v_out.image_embeds))
        r_out = res(v_out.hidden_states[-2])
    np.savez_compressed(os.path.join(OUT, "enc_ip_adapter.npz"), image_embeds=v_out.image_embeds.numpy(), proj_tokens=p_out.numpy(),
                        proj_tokens_uncond=p_zero.numpy(), clip_hidden=v_out.hidden_states[-2].numpy(), resampler_tokens=r_out.numpy(),
                        proj_seed=np.int64(55), resampler_seed=np.int64(56))
    for f in ("enc_clip_text.npz", "enc_clip_vision.npz", "enc_ip_adapter.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
