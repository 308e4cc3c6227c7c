"""Host-side engine: weight packing, UNet3D / VAE-decoder op schedules, DDIM tables, sampling loop."""
from .config import DDIMConfig, UNet3DConfig, VAEDecoderConfig  # noqa: F401
