from .pipeline_stable_diffusion import StableDiffusionPipeline, StableDiffusionPipelineOutput  # noqa: F401
