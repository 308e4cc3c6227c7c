"""Drop-in `ip_adapter` package on the MI355X engine: the attention-processor API, the image-projection models and the
MyIPAdapter / MyIPAdapterPlus front-ends (CLIP vision tower + projection as HIP op schedules)."""
from .attention_processor import (AttnProcessor, AttnProcessor2_0, CNAttnProcessor, CNAttnProcessor2_0,  # noqa: F401
                                  IPAttnProcessor, IPAttnProcessor2_0)
from .my_ip_adapter import ImageProjModel, MyIPAdapter, MyIPAdapterPlus  # noqa: F401
from .resampler import Resampler  # noqa: F401
