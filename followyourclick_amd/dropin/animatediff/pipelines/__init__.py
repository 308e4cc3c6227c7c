from .pipeline_animation import AnimationPipeline, AnimationPipelineOutput  # noqa: F401
