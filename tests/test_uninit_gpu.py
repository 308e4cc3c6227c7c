"""No result may depend on what an allocated buffer held before its producer ran.

Round 6: `test_full_width_forward_vs_reference` once gave 2.36e-2 inside a full-suite run and 1.34e-2 alone.  (That library turned out to
be a build of a half-edited source; the checks written to find the cause stay.)  A fresh process hides reads of unwritten memory - new
device memory is zero - while a long session hands out buffers full of old activations.  These tests run the SAME forward twice, every engine buffer and
the split-K / statistics scratch pre-filled with 0x00 and then with 0xFF bytes (NaN in every float format), and demand identical
bits.  On a mismatch the forward is re-run with every op wrapped, and the failure message names the first op whose output picked
the fill up.
"""
import pytest
import torch

from followyourclick_amd import ops as ops_mod
from followyourclick_amd.engine import UNet3DConfig, base
from followyourclick_amd.engine.schema import random_state_dict, unet_schema
from followyourclick_amd.engine.unet3d import UNet3DEngine
from followyourclick_amd.engine.weights import pack_unet

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(B, F, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.zeros(B * F * H * W, 64)
    x[:, :9] = torch.randn(B * F * H * W, 9, generator=g)
    return x, torch.randn(B, 77, 768, generator=g)


def _forward(eng, x, text, B, F, H, W, fill):
    o = ops_mod.get()
    old = base.ALLOC_FILL
    base.ALLOC_FILL = fill
    try:
        if o._ws is not None:
            o._ws.fill_(fill)
        eng.prepare_context(text.to(DEV))
        _, temb = eng.prepare_time_embeddings([481], [2.0] * B, [4.0] * B, B)
        out = eng.forward(x.to(DEV, eng.dtype), temb, B, F, H, W)
        torch.cuda.synchronize()
        return out[:, :4].contiguous()
    finally:
        base.ALLOC_FILL = old


def _first_poisoned_op(eng, x, text, B, F, H, W):
    """re-run under the NaN fill with every op wrapped: the first ops (in call order) that leave NaN in one of their tensors"""
    o = ops_mod.get()
    found, wrapped = [], {}

    def nan_frac(t):
        return float(torch.isnan(t.float() if t.dtype != torch.float64 else t).float().mean()) if t.is_floating_point() and t.numel() else 0.0

    def wrap(name, fn):
        def run(*a, **k):
            tens = [(f"arg{i}", t) for i, t in enumerate(a) if isinstance(t, torch.Tensor)] + [(n, t) for n, t in k.items() if isinstance(t, torch.Tensor)]
            before = {n: nan_frac(t) for n, t in tens}
            r = fn(*a, **k)
            torch.cuda.synchronize()
            after = {n: nan_frac(t) for n, t in tens}
            # a fresh output is all fill (1.0) before the call; whatever is still NaN afterwards was either not written or computed from NaN
            left = {n: (round(before[n], 4), round(after[n], 4)) for n, _ in tens if after[n] > 0.0}
            if left and len(found) < 12:
                dims = {n: v for n, v in k.items() if isinstance(v, (int, float)) and not isinstance(v, bool)}
                found.append(f"{name}: NaN fraction (before, after) {left} {dims}")
            return r
        return run

    skip = {"ensure_init", "set_tuning", "device_caps"}
    for name in dir(o):
        fn = getattr(o, name)
        if name.startswith("_") or name in skip or not callable(fn) or name.endswith(("_supported", "_bytes", "_layout", "_parts")):
            continue
        wrapped[name] = fn
        setattr(o, name, wrap(name, fn))
    try:
        _forward(eng, x, text, B, F, H, W, 0xFF)
    finally:
        for name in wrapped:
            delattr(o, name)          # the instance attributes shadow the class methods: remove them again
    return found


@pytest.fixture(scope="module")
def engines():
    """one engine per (configuration, precision) for the module; the state dict of a configuration is drawn once"""
    sds, engs = {}, {}

    def get(kw, dtype):
        key = repr(sorted(kw.items()))
        if (key, dtype) not in engs:
            cfg = UNet3DConfig(**kw)
            if key not in sds:
                sds[key] = random_state_dict(unet_schema(cfg), seed=3)
            engs[(key, dtype)] = UNet3DEngine(pack_unet(sds[key], cfg, dtype, DEV))
        return engs[(key, dtype)]
    yield get
    engs.clear()
    sds.clear()
    torch.cuda.empty_cache()


CASES = [  # (name, config kwargs, B, F, H, W)
    ("tiny 8x8", dict(block_out_channels=(64, 128, 256, 256), cross_attention_dim=64, sample_size=8), 2, 4, 8, 8),
    ("tiny odd 10x12", dict(block_out_channels=(64, 128, 256, 256), cross_attention_dim=64, sample_size=8), 2, 3, 10, 12),
    ("full width 16x16", dict(), 2, 4, 16, 16),        # (split-K convolutions + their statistics finish, M = 32 .. 2048)
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32], ids=["bf16", "f16", "f32"])
@pytest.mark.parametrize("case", CASES, ids=[c[0].replace(" ", "_") for c in CASES])
def test_forward_does_not_read_unwritten_memory(case, dtype, engines, fullwidth):
    name, kw, B, F, H, W = case
    if dtype != torch.bfloat16 and not kw:
        pytest.skip("full width runs in the benchmarked precision only (f16 shares the kernels' templates, the f32 parity mode every epilogue / statistics "
                    "path with the tiny cases): suite time")
    if kw:
        kw = dict(kw)
        text_dim = kw["cross_attention_dim"]
    else:
        text_dim = 768
    if kw:
        eng = engines(kw, dtype)
    else:       # the session's full-width engine (tests/conftest.py::fullwidth: packed once for test_fullwidth_gpu.py / test_reference_gpu.py)
        from oracle import functional as Fn
        eng = fullwidth.engine(Fn.UNetConfig(), UNet3DConfig(), dtype, seed=0)
    x, text = _inputs(B, F, H, W, 11)
    text = text[:, :, :text_dim].contiguous()
    a = _forward(eng, x, text, B, F, H, W, 0x00)
    b = _forward(eng, x, text, B, F, H, W, 0xFF)
    assert torch.isfinite(a).all()
    same = torch.equal(a.view(torch.uint8 if a.element_size() == 1 else torch.int16 if a.element_size() == 2 else torch.int32),
                       b.view(torch.uint8 if b.element_size() == 1 else torch.int16 if b.element_size() == 2 else torch.int32))
    if not same:
        ops = _first_poisoned_op(eng, x, text, B, F, H, W)
        nan_frac = float(torch.isnan(b.float()).float().mean())
        pytest.fail(f"{name} {dtype}: the output depends on the initial content of a buffer (NaN fraction under the 0xFF fill {nan_frac:.3f});\n  "
                    + "\n  ".join(ops or ["no op turned clean inputs into NaN: the dependence is on finite stale data (statistics partials?)"]))
