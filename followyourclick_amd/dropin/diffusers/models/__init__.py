from .vae import AutoencoderKL  # noqa: F401
from .unet_2d_condition import UNet2DConditionModel  # noqa: F401
