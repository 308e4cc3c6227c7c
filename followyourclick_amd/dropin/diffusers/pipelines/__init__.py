from .stable_diffusion import StableDiffusionPipeline  # noqa: F401
