"""CUDA-autocast cast policy, emulated on the CPU (TEST INFRASTRUCTURE - container only).

The reference's deployed precision is fp32 weights under ``torch.autocast("cuda")``
(scripts/inference.py:294, animatediff/pipelines/pipeline_animation.py:686): matmul-class ops
(conv / linear / matmul / bmm / baddbmm / einsum) run with operands cast to the low-precision
dtype and produce that dtype; group_norm / layer_norm / softmax / exp / pow / sum run in fp32;
everything else follows type promotion.  ``torch.autocast("cuda")`` disables itself without a
GPU and ``torch.autocast("cpu")`` applies a *different* op list (no fp32 list for the norms), so
to obtain "the same model at the production precision" from the real reference in this container
this TorchFunctionMode re-creates the CUDA policy (torch docs, "CUDA Ops that can autocast to
float16 / float32") around the reference's own Python.  bf16 on the CPU accumulates in f32 and
rounds the result once (oneDNN/AMX), like the MFMA path on the GPU.
"""
import torch
from torch.overrides import TorchFunctionMode

LOWP = {"conv1d", "conv2d", "conv3d", "conv_transpose2d", "linear", "matmul", "__matmul__", "__rmatmul__", "bmm", "baddbmm",
        "mm", "addmm", "addbmm", "mv", "einsum", "chain_matmul", "multi_dot", "prelu", "scaled_dot_product_attention"}
FP32 = {"group_norm", "layer_norm", "softmax", "log_softmax", "exp", "expm1", "log", "log2", "log10", "log1p", "pow", "__pow__",
        "__rpow__", "rsqrt", "reciprocal", "__rtruediv__", "__rdiv__", "sum", "prod", "cumsum", "cumprod", "norm", "normalize",
        "softplus", "mse_loss", "erfinv", "cosh", "sinh", "tan", "acos", "asin"}


def _cast(x, dt):
    if isinstance(x, torch.Tensor) and x.is_floating_point() and x.dtype != dt and x.dtype != torch.float64:
        return x.to(dt)
    if isinstance(x, (list, tuple)):
        return type(x)(_cast(v, dt) for v in x)
    return x


class CudaAutocastOnCpu(TorchFunctionMode):
    def __init__(self, dtype=torch.bfloat16):
        super().__init__()
        self.dtype = dtype

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", "")
        if name in LOWP:
            args = tuple(_cast(a, self.dtype) for a in args)
            kwargs = {k: _cast(v, self.dtype) for k, v in kwargs.items()}
        elif name in FP32:
            args = tuple(_cast(a, torch.float32) for a in args)
            kwargs = {k: _cast(v, torch.float32) for k, v in kwargs.items()}
        return func(*args, **kwargs)
