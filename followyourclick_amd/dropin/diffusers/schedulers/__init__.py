from .scheduling_ddim import DDIMScheduler  # noqa: F401
