"""followyourclick_amd - MI355X-native denoising engine for FollowYourClick's AnimateDiff-style sampling loop.

    import followyourclick_amd
    followyourclick_amd.install_dropin()       # `animatediff`, `diffusers`, `ip_adapter` now resolve to this engine
    from animatediff.pipelines.pipeline_animation import AnimationPipeline

The compute library (libfyc_hip.so, hand-written gfx950 HIP kernels behind the C ABI of include/fyc.h) is
required: there is no CPU or eager-PyTorch fallback.
"""
import os
import sys

__version__ = "0.1.0"
DROPIN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropin")


def default_compute_dtype():
    """The element type the drop-in models compute in when their constructor is not given `compute_dtype` - which is always the case
    when the reference's unmodified scripts build them.  FYC_COMPUTE_DTYPE = bf16 (default: BASELINE configs[1]) | f16 (IEEE half, the
    precision class of the reference's own `torch.autocast("cuda")` deployment, scripts/inference.py:294) | f32 (parity mode)."""
    import torch
    name = os.environ.get("FYC_COMPUTE_DTYPE", "bf16").lower()
    table = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "f16": torch.float16, "fp16": torch.float16, "float16": torch.float16,
             "half": torch.float16, "f32": torch.float32, "fp32": torch.float32, "float32": torch.float32}
    if name not in table:
        raise ValueError(f"FYC_COMPUTE_DTYPE={name!r}: expected one of bf16, f16, f32")
    return table[name]


def install_dropin(force: bool = False) -> str:
    """Put the drop-in `animatediff` / `diffusers` / `ip_adapter` packages first on sys.path so that the
    reference's scripts import this engine instead of the reference's torch modules."""
    def foreign(mod) -> bool:     # the reference's `animatediff` has no __init__.py: a namespace package, __file__ is None
        paths = [getattr(mod, "__file__", None) or ""] + [str(p) for p in (getattr(mod, "__path__", None) or [])]
        return not any(p.startswith(DROPIN_DIR) for p in paths if p)

    loaded = [m for m in ("animatediff", "diffusers", "ip_adapter") if m in sys.modules and foreign(sys.modules[m])]
    if loaded and not force:
        raise RuntimeError(f"{loaded} already imported from elsewhere; call install_dropin() before importing them "
                           "(or pass force=True to evict them)")
    for name in [k for k in sys.modules if k.split(".")[0] in ("animatediff", "diffusers", "ip_adapter")]:
        if force or name.split(".")[0] in loaded:
            del sys.modules[name]
    if DROPIN_DIR in sys.path:
        sys.path.remove(DROPIN_DIR)
    sys.path.insert(0, DROPIN_DIR)
    return DROPIN_DIR
