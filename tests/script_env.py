"""Scaffolding for running the reference's OWN entry script (scripts/inference.py, unmodified, through runpy) on top of the
drop-in packages.  TEST INFRASTRUCTURE only.  Three kinds of stand-ins, none of which touches the script or the product:

  * a fabricated `pretrained_model_path` (tiny CLIP tokenizer / text encoder written by transformers itself, tiny UNet / VAE
    `config.json` + `diffusion_pytorch_model.bin` with the reference's state-dict keys, a scheduler config), a motion-module
    checkpoint with the `module.` prefix the script strips (scripts/inference.py:170-181), a prompt config, one image + mask;
  * stubs for third-party packages the script imports that are not in this image (omegaconf, torchvision.transforms) and for the
    parts of the authors' environment it hard-codes (pandas.read_excel of their sheet, the mask directory under /teg_amai);
  * on a box without a GPU: "cuda" is aliased to the CPU and the NCCL group the script opens becomes gloo, so that the host
    orchestration runs on the op emulator.
"""
import json
import os
import sys
import types

import numpy as np
import torch
import yaml

# the reference tree itself (build container), or the byte copies of its entry scripts + YAMLs that oracle/stage_ref_scripts.py
# puts under the git-ignored oracle/_ref/ (they travel to the GPU box with the repo snapshot; /root/reference does not)
_STAGED = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
REF_ROOT = "/root/reference" if os.path.exists("/root/reference/scripts/inference.py") else _STAGED
REF_SCRIPT = os.path.join(REF_ROOT, "scripts", "inference.py")
REF_INFERENCE_CFG = os.path.join(REF_ROOT, "configs", "inference", "inference_img_embed_mask_condition_zero_snr_.yaml")

WIDTHS = (64, 128, 256, 256)
CTX = 64


# ---- stand-ins for packages that are absent from the image -----------------------------------------------------------------
class _Cfg(dict):
    """just enough of omegaconf.DictConfig: attribute + item access, nested, mutable"""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return _Cfg({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _plain(x):
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    return x


def install_absent_packages(monkeypatch):
    # transformers decides at import time whether torchvision exists: resolve every name the script imports from it BEFORE the stub
    from transformers import CLIPImageProcessor, CLIPTextModel, CLIPTokenizer, CLIPVisionModelWithProjection  # noqa: F401
    if "omegaconf" not in sys.modules:
        try:
            import omegaconf  # noqa: F401
        except ImportError:
            om = types.ModuleType("omegaconf")

            class OmegaConf:
                @staticmethod
                def load(path):
                    with open(path) as f:
                        return _wrap(yaml.safe_load(f))

                @staticmethod
                def to_container(cfg, **kw):
                    return _plain(cfg)

                @staticmethod
                def save(cfg, path):
                    with open(path, "w") as f:
                        yaml.safe_dump(_plain(cfg), f)
            om.OmegaConf = OmegaConf
            import importlib.machinery as _M
            om.__spec__ = _M.ModuleSpec("omegaconf", None)
            monkeypatch.setitem(sys.modules, "omegaconf", om)
    try:
        import pytorch_lightning  # noqa: F401
    except ImportError:          # scripts/inference_w_image_cond.py:36 imports seed_everything from it
        import importlib.machinery as _M
        import random
        pl = types.ModuleType("pytorch_lightning")

        def seed_everything(seed, *a, **k):
            random.seed(seed); np.random.seed(seed % (2 ** 32)); torch.manual_seed(seed)
            return seed
        pl.seed_everything = seed_everything
        pl.__spec__ = _M.ModuleSpec("pytorch_lightning", None)
        monkeypatch.setitem(sys.modules, "pytorch_lightning", pl)
    try:
        import torchvision.transforms  # noqa: F401
    except ImportError:
        tv, tr = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")

        class Compose:
            def __init__(self, ts):
                self.ts = ts

            def __call__(self, img):
                for t in self.ts:
                    img = t(img)
                return img

        class Resize:            # int: the smaller edge, PIL image in / out (what the script uses, scripts/inference.py:320-323)
            def __init__(self, size):
                self.size = size

            def __call__(self, img):
                w, h = img.size
                s = self.size / min(w, h)
                return img.resize((max(1, round(w * s)), max(1, round(h * s))))

        class CenterCrop:
            def __init__(self, size):
                self.h, self.w = size

            def __call__(self, img):
                w, h = img.size
                l, t = (w - self.w) // 2, (h - self.h) // 2
                return img.crop((l, t, l + self.w, t + self.h))
        import importlib.machinery as _M
        tr.Compose, tr.Resize, tr.CenterCrop = Compose, Resize, CenterCrop
        tv.transforms = tr
        tv.__spec__, tr.__spec__ = _M.ModuleSpec("torchvision", None, is_package=True), _M.ModuleSpec("torchvision.transforms", None)
        tv.__path__ = []
        monkeypatch.setitem(sys.modules, "torchvision", tv)
        monkeypatch.setitem(sys.modules, "torchvision.transforms", tr)


def alias_cuda_to_cpu(monkeypatch):
    """a box without a GPU: every `.to("cuda")`, `.to(local_rank)`, `torch.Generator(device="cuda")`, `torch.cuda.set_device` of the
    script lands on the CPU; its NCCL process group becomes gloo"""
    import torch.distributed as dist

    def fix(a):
        if isinstance(a, int) and not isinstance(a, bool):
            return torch.device("cpu")
        if isinstance(a, str) and a.startswith("cuda"):
            return "cpu"
        if isinstance(a, torch.device) and a.type == "cuda":
            return torch.device("cpu")
        return a
    m_to, t_to, gen = torch.nn.Module.to, torch.Tensor.to, torch.Generator
    monkeypatch.setattr(torch.nn.Module, "to", lambda self, *a, **k: m_to(self, *[fix(x) for x in a], **{n: fix(v) for n, v in k.items()}))
    monkeypatch.setattr(torch.Tensor, "to", lambda self, *a, **k: t_to(self, *[fix(x) for x in a], **{n: fix(v) for n, v in k.items()}))
    class _Generator(gen):                                       # a type, not a function: torch annotates with `torch.Generator | None`
        def __new__(cls, device="cpu"):
            return super().__new__(cls, device=fix(device))
    monkeypatch.setattr(torch, "Generator", _Generator)
    monkeypatch.setattr(torch.cuda, "set_device", lambda *_: None)
    real_init = dist.init_process_group
    monkeypatch.setattr(dist, "init_process_group", lambda backend=None, **kw: real_init("gloo", **kw))


# ---- the fabricated checkpoint tree ----------------------------------------------------------------------------------------------
def _bytes_to_unicode():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs, n = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return [chr(c) for c in cs]


def fabricate_model_dir(root: str, seed: int = 0, inference_cfg: str = None, wrap_state_dict: bool = True) -> dict:
    """stable-diffusion-v1-5-shaped directory at tiny widths; returns the paths of the checkpoints that go with it.
    inference_cfg: the YAML whose `unet_additional_kwargs` shape the 3-D UNet (default: the reference's shipped one);
    wrap_state_dict: the motion-module checkpoint as `{"state_dict": {"module.<key>": ...}}` (what scripts/inference.py strips) or as
    a bare state dict (scripts/inference_org.py / inference_w_image_cond.py load the wrapped form with strict=True, i.e. expect a
    FULL model there; the bare form goes through strict=False)"""
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTokenizer
    from animatediff.models.unet import UNet3DConditionModel
    from diffusers import AutoencoderKL, UNet2DConditionModel
    torch.manual_seed(seed)
    # tokenizer: byte-level vocabulary without merges (every character a token)
    chars = sorted(set(_bytes_to_unicode()))
    vocab = {}
    for c in chars:
        vocab[c] = len(vocab)
    for c in chars:
        vocab[c + "</w>"] = len(vocab)
    vocab["<|startoftext|>"], vocab["<|endoftext|>"] = len(vocab), len(vocab) + 1
    CLIPTokenizer(vocab=vocab, merges=[], model_max_length=77).save_pretrained(os.path.join(root, "tokenizer"))
    tcfg = CLIPTextConfig(vocab_size=len(vocab), hidden_size=CTX, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2,
                          max_position_embeddings=77, projection_dim=CTX, bos_token_id=vocab["<|startoftext|>"],
                          eos_token_id=vocab["<|endoftext|>"], pad_token_id=vocab["<|endoftext|>"])
    CLIPTextModel(tcfg).save_pretrained(os.path.join(root, "text_encoder"))
    # 2-D UNet (the script builds the 3-D one from it with from_pretrained_2d, and a 2-D one for the first image)
    ucfg = dict(sample_size=8, in_channels=4, out_channels=4, block_out_channels=list(WIDTHS), layers_per_block=2, cross_attention_dim=CTX,
                attention_head_dim=8, down_block_types=["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"],
                up_block_types=["UpBlock2D"] + ["CrossAttnUpBlock2D"] * 3)
    unet2d = UNet2DConditionModel(**ucfg)
    sd2d = {k: torch.randn_like(v) * (0.3 if v.dim() > 1 else 0.05) + (1.0 if k.endswith("norm.weight") or "norm1.weight" in k or "norm2.weight" in k or "norm3.weight" in k or "norm_out.weight" in k else 0.0)
            for k, v in unet2d.state_dict().items()}
    for k, v in sd2d.items():                                  # fan-in scaling keeps activations O(1)
        if v.dim() > 1:
            sd2d[k] = v / (v[0].numel() ** 0.5) / 0.3
    os.makedirs(os.path.join(root, "unet"), exist_ok=True)
    with open(os.path.join(root, "unet", "config.json"), "w") as f:
        json.dump(ucfg, f)
    torch.save(sd2d, os.path.join(root, "unet", "diffusion_pytorch_model.bin"))
    # VAE
    vcfg = dict(in_channels=3, out_channels=3, down_block_types=["DownEncoderBlock2D"] * 4, up_block_types=["UpDecoderBlock2D"] * 4,
                block_out_channels=[64, 128, 128, 128], layers_per_block=2, latent_channels=4, norm_num_groups=32, sample_size=64)
    vae = AutoencoderKL(**vcfg)
    sdv = {}
    for k, v in vae.state_dict().items():
        if v.dim() > 1:
            sdv[k] = torch.randn_like(v) / (v[0].numel() ** 0.5)
        else:
            sdv[k] = torch.randn_like(v) * 0.05 + (1.0 if "norm" in k and k.endswith("weight") else 0.0)
    os.makedirs(os.path.join(root, "vae"), exist_ok=True)
    with open(os.path.join(root, "vae", "config.json"), "w") as f:
        json.dump(vcfg, f)
    torch.save(sdv, os.path.join(root, "vae", "diffusion_pytorch_model.bin"))
    os.makedirs(os.path.join(root, "scheduler"), exist_ok=True)
    with open(os.path.join(root, "scheduler", "scheduler_config.json"), "w") as f:
        json.dump(dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", steps_offset=1,
                       clip_sample=False, set_alpha_to_one=False), f)
    # motion module checkpoint: the temporal / fps / flow tensors of the 3-D model, saved from a DDP-wrapped training run ("module.")
    with open(inference_cfg or REF_INFERENCE_CFG) as f:
        extra = yaml.safe_load(f)["unet_additional_kwargs"]
    unet3d = UNet3DConditionModel.from_pretrained_2d(root, subfolder="unet", unet_additional_kwargs=extra)
    mm = {}
    for k, v in unet3d.state_dict().items():
        if "motion_modules" in k or k.startswith(("fps_embedding", "motion_embedding")) or k == "conv_in.weight":
            if k.endswith("pos_encoder.pe"):
                mm["module." + k] = v.clone()
            elif v.dim() > 1:
                mm["module." + k] = torch.randn_like(v) / (v[0].numel() ** 0.5)
            else:
                mm["module." + k] = torch.randn_like(v) * 0.05 + (1.0 if "norm" in k and k.endswith("weight") else 0.0)
    ckpt = os.path.join(root, "motion_module.ckpt")
    torch.save({"state_dict": mm} if wrap_state_dict else {k[len("module."):]: v for k, v in mm.items()}, ckpt)
    # spatial weights of the 2-D UNet as a training checkpoint (inference_w_image_cond.py: `.ckpt`, "state_dict", `module.` prefix)
    sd2d_b = {k: torch.randn_like(v) * 0.02 + v for k, v in sd2d.items()}
    unet2d_ckpt = os.path.join(root, "unet2d_finetuned.ckpt")
    torch.save({"state_dict": {"module." + k: v for k, v in sd2d_b.items()}}, unet2d_ckpt)
    # a tiny CLIP vision model with projection (image_pretrained_model_path of the image-conditioned scripts)
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    vdir = os.path.join(root, "clip_vision")
    CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2, image_size=224,
                                                   patch_size=32, projection_dim=CTX)).save_pretrained(vdir)
    return dict(motion_ckpt=ckpt, unet_extra=extra, unet2d_ckpt=unet2d_ckpt, clip_vision_dir=vdir)


def write_run_files(root: str, motion_ckpt: str, steps: int, size: int):
    """prompt config (the shipped configs/prompts layout), one first image + region mask, the sheet stand-in"""
    from PIL import Image
    cfg = {"TinyModel": dict(base="", path="", motion_module=[motion_ckpt], seed=[1], steps=steps, guidance_scale=8.0,
                             prompt=["a corgi waving its tail"], n_prompt=["blurry"])}
    cfg_path = os.path.join(root, "prompts.yaml")
    with open(cfg_path, "w") as f:
        yaml.safe_dump(cfg, f)
    rng = np.random.default_rng(3)
    img_dir = os.path.join(root, "images")
    os.makedirs(img_dir, exist_ok=True)
    img_path, mask_path = os.path.join(img_dir, "corgi.png"), os.path.join(root, "corgi_mask.png")
    Image.fromarray(rng.integers(0, 256, (size, size, 3), dtype=np.uint8)).save(img_path)
    m = np.zeros((size, size, 3), dtype=np.uint8)
    m[size // 4: 3 * size // 4, size // 4: 3 * size // 4] = 255          # the clicked region (a SAM mask stand-in)
    Image.fromarray(m).save(mask_path)
    sheet = os.path.join(root, "prompts.xlsx")
    open(sheet, "w").write("stand-in: pandas.read_excel is patched by the test\n")
    return cfg_path, sheet, img_path, mask_path


def patch_authors_environment(monkeypatch, img_path: str, mask_path: str, prompt: str):
    """pandas.read_excel of the authors' sheet, and their mask directory (hard-coded at scripts/inference.py:82)"""
    import pandas as pd
    from PIL import Image
    monkeypatch.setattr(pd, "read_excel", lambda *_a, **_k: pd.DataFrame({"prompt": [prompt], "image": [img_path]}))
    real_open = Image.open

    def open_(fp, *a, **k):
        if isinstance(fp, str) and fp.startswith("/teg_amai"):
            return real_open(mask_path, *a, **k)
        return real_open(fp, *a, **k)
    monkeypatch.setattr(Image, "open", open_)
