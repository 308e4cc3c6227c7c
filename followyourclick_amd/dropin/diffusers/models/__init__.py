from .vae import AutoencoderKL  # noqa: F401
