"""Drop-in for the names the reference imports from its vendored diffusers 0.11.1 on the hot path."""
__version__ = "0.11.1+fyc.mi355x"
from .models.vae import AutoencoderKL  # noqa: F401,E402
from .schedulers.scheduling_ddim import DDIMScheduler  # noqa: F401,E402
from .models.unet_2d_condition import UNet2DConditionModel  # noqa: F401,E402
from .pipelines.stable_diffusion import StableDiffusionPipeline  # noqa: F401,E402
