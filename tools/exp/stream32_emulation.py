"""CPU experiment (round 6, review item 6): how much of the 16-bit modes' distance to the f32 reference is the ROUNDING OF THE
RESIDUAL STREAM?  Runs the full-width small forward (tests/golden/unet_full_small_fwd.npz: F = 4, 16x16 latent, SD-1.5 widths) on the op
emulator (tests/emu_ops.py) three ways per 16-bit type:
  plain      every tensor stored in the 16-bit type (what FYC_F16 / bf16 do today);
  stream32   the tensors that carry `+ residual` between blocks (outputs of every GEMM / conv with a residual) stay f32; every MMA
             operand, every norm output, q / k / v, the attention output are rounded to the 16-bit type as today.
Test infrastructure only (imports tests/ and oracle/).  Usage: python tools/exp/stream32_emulation.py [f16|bf16]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("FYC_FUSE_FF", "0")
os.environ.setdefault("FYC_FUSE_TEMPORAL", "0")
os.environ.setdefault("FYC_FUSE_PANEL", "0")
import numpy as np
import torch

from emu_ops import EmuOps
from followyourclick_amd.engine import UNet3DConfig
from followyourclick_amd.engine.unet3d import UNet3DEngine
from followyourclick_amd.engine.weights import pack_unet
from oracle import functional as Fn
from oracle import weights as W


class Stream32(EmuOps):
    """storage is f32 everywhere (engine.dtype is switched to f32 after packing); this wrapper re-creates the 16-bit roundings of the
    real kernels except on the residual stream"""
    name = "emu"

    def __init__(self, T):
        super().__init__()
        self.T = T

    def r(self, t):
        return None if t is None else (t if t.dtype != torch.float32 else t.to(self.T))     # a 16-bit tensor: the conv slab size follows the element size

    def rnd_(self, t):
        if t is not None and t.dtype == torch.float32:
            t.copy_(t.to(self.T).float())

    def gemm(self, a, w, out, **kw):
        if w.dtype == torch.float32 and a.dtype == torch.float32 and kw.get("mode", 0) == 0 and kw.get("heads") is None and w.numel() < 4e6 and kw["M"] <= 64:
            return super().gemm(a, w, out, **kw)            # the f32 time-embedding MLPs
        if kw.get("a2") is not None:
            kw["a2"] = self.r(kw["a2"])
        super().gemm(self.r(a), w, out, **kw)
        if kw.get("heads") is not None:
            for t in kw["heads"]["outs"]:
                self.rnd_(t)
        elif kw.get("residual") is None:
            self.rnd_(out)                                   # not a stream tensor: stored in 16 bits as today

    def attention(self, q, k, vt, o, **kw):
        super().attention(q.to(self.T), k.to(self.T), vt.to(self.T), o, **kw)
        self.rnd_(o)

    def temporal_attention(self, qkv, o, **kw):
        super().temporal_attention(qkv.to(self.T), o, **kw)
        self.rnd_(o)

    def gn_apply(self, x, stats, gamma, beta, y, **kw):
        super().gn_apply(x, stats, gamma, beta, y, **kw)
        self.rnd_(y)

    def gn_apply_cs(self, x1, cs1, gamma, beta, y, **kw):
        super().gn_apply_cs(x1, cs1, gamma, beta, y, **kw)
        self.rnd_(y)

    def layernorm(self, x, gamma, beta, y, **kw):
        super().layernorm(x, gamma, beta, y, **kw)
        self.rnd_(y)

    def row_stats(self, x, stats, **kw):
        super().row_stats(self.r(x), stats, **kw)            # statistics of the 16-bit copy the consuming GEMM reads

    def concat_channels(self, a_, b_, y, **kw):
        super().concat_channels(a_, b_, y, **kw)
        self.rnd_(y)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def run(sd, g, inp, x9, T, stream32):
    cfg = UNet3DConfig()
    ops = Stream32(T) if stream32 else EmuOps()
    eng = UNet3DEngine(pack_unet(sd, cfg, T, "cpu"), ops=ops)
    if stream32:
        eng.dtype = torch.float32                            # buffers are f32; the wrapper rounds what the kernels would store in 16 bits
    eng.prepare_context(inp["text"])
    _, temb = eng.prepare_time_embeddings([int(g["timestep"])], g["fps"].tolist(), g["flow"].tolist(), 2)
    F, H, Wd = int(g["frames"]), int(g["h"]), int(g["w"])
    x = x9.permute(0, 2, 3, 4, 1).reshape(-1, 9)
    xin = torch.zeros(x.shape[0], 64, dtype=eng.dtype)
    xin[:, :9] = x.to(T).to(eng.dtype) if stream32 else x.to(T)
    out = eng.forward(xin, temb, 2, F, H, Wd).float().reshape(2, F, H, Wd, 4).permute(0, 4, 1, 2, 3)
    return out


def main():
    T = {"f16": torch.float16, "bf16": torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else "f16"]
    g = {k: torch.from_numpy(v) if v.shape else v for k, v in np.load(os.path.join(ROOT, "tests/golden/unet_full_small_fwd.npz")).items()}
    cfg = Fn.UNetConfig()
    sd = W.make_weights(W.unet_state_shapes(cfg), seed=0)
    F, H, Wd = int(g["frames"]), int(g["h"]), int(g["w"])
    inp = W.seeded_inputs(cfg, 1, F, H, Wd, seed=int(g["input_seed"]))
    x9 = torch.cat([Fn.build_model_input(inp["latents"], inp["first_image_latents"], inp["first_images_mask"])] * 2)
    for s32 in (False, True):
        t0 = time.time()
        with torch.no_grad():
            out = run(sd, g, inp, x9, T, s32)
        print(f"{T} {'stream32' if s32 else 'plain   '}: rel-L2 vs ref-f32 {rel(out, g['out_f32']):.3e}   ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
