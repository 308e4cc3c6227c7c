"""Drop-in `ip_adapter` package: the attention-processor API surface on the MI355X attention kernels.
(MyIPAdapter / MyIPAdapterPlus, i.e. the CLIP-vision front-end, are the conditioning "next" row of
SURVEY.md 8f and are not part of this package yet.)"""
from .attention_processor import (AttnProcessor, AttnProcessor2_0, CNAttnProcessor, CNAttnProcessor2_0,  # noqa: F401
                                  IPAttnProcessor, IPAttnProcessor2_0)
