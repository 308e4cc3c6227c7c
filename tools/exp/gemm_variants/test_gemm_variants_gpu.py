"""GPU tests of the round-4 GEMM main-loop experiments (tools/exp/gemm_variants/: ping-pong wave groups = tile configs 21 / 22 / 23,
epilogue under the next tile's K loop = config 31).  They are NOT part of the product library or of `pytest tests/`: build a library
with them and point the tests at it -

    FYC_GEMM_VARIANTS=1 FYC_BUILD_LIB=tools/exp/libfyc_variants.so python -m followyourclick_amd._build
    FYC_LIB_PATH=tools/exp/libfyc_variants.so python -m pytest tools/exp/gemm_variants/test_gemm_variants_gpu.py -q      (GPU box)

Moved out of tests/test_kernels_gpu.py in round 5 (68 items for kernels no default path selects, measured slower on every shape:
profiles/r04_gemm_pingpong_and_prefetch_sweep.txt, r04_gemm_phase_trace.txt section C)."""
import math
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_kernels_gpu import DT, RTOL, close, emu, hip, rnd  # noqa: E402,F401  (fixtures + helpers of the product suite)
import test_kernels_gpu as K_  # noqa: E402

pytestmark = pytest.mark.gpu
if not os.environ.get("FYC_LIB_PATH"):
    pytest.skip("needs a library built with FYC_GEMM_VARIANTS=1 (FYC_LIB_PATH)", allow_module_level=True)


# ---- ping-pong main loop (csrc/gemm_pp_kernel.h: tile configs 21 / 22 / 23) -------------------------------------------------------
PP_TILES = [21, 22, 23, 31]       # 31: the overlapped-epilogue kernel (csrc/gemm_ov_kernel.h) takes the same streams


@pytest.mark.parametrize("tile", PP_TILES)
@pytest.mark.parametrize("M,N,K,feat", [
    (70000, 960, 320, "rowbias"),            # more tiles than CUs: the persistent stream crosses tile boundaries; ragged last row tile
    (33000, 640, 640, "res"),                # two column tiles, residual
    (3000, 1280, 1280, "res"),               # 20 K tiles
    (5000, 320, 1600, "dual"),               # dual-source K (merged FF2 | proj_out): the switch to a2 inside the K loop
    (777, 200, 128, "res"),                  # two K tiles (the shortest stream the loop is built for), ragged N and M
    (40000, 2560, 320, "geglu"),             # GEGLU epilogue + folded LayerNorm, wide layer (column-strip tile order)
    (9000, 1920, 640, "heads"),              # head-split epilogue
])
def test_gemm_pingpong_plain(hip, emu, tile, M, N, K, feat):
    """the ping-pong K loop against the specification on streams of many tiles per workgroup, every epilogue family"""
    T = torch.bfloat16
    if feat == "geglu" and tile in (22, 31):
        pytest.skip("tiles 22 / 31 give a wave an odd number of column blocks: GEGLU runs on the 256x320 tile")
    a, w = rnd((M, K), T, 1), rnd((N, K), T, 2, 1 / math.sqrt(K))
    bias = rnd((N,), torch.float32, 3)
    kw = dict(M=M, N=N, K=K, lda=K, ldw=K, ldo=N, ldr=N)
    ex = {}
    if feat == "rowbias":
        rpb = 1000
        ex = dict(rowbias=rnd(((M + rpb - 1) // rpb, N), torch.float32, 5), rows_per_batch=rpb, out_scale=0.75)
    elif feat == "res":
        ex = dict(residual=rnd((M, N), T, 4))
    elif feat == "dual":
        k2 = 320
        a = rnd((M, K - k2), T, 1)
        ex = dict(a2=rnd((M, k2), T, 7), k_split=K - k2, lda2=k2, residual=rnd((M, N), T, 4))
        kw["lda"] = K - k2
    elif feat == "geglu":
        st = torch.stack([rnd((M,), torch.float32, 8) * 0.1, rnd((M,), torch.float32, 9).abs() + 0.5], dim=1).contiguous()
        ex = dict(epilogue=1, ln_stats=st, ln_colsum=rnd((N,), torch.float32, 10))
        kw["ldo"] = N // 2
    dev = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in ex.items()}
    if feat == "heads":
        H, tokens = 8, 1000
        d = N // 3 // H
        shapes = [(M // tokens, H, tokens, d), (M // tokens, H, tokens, d), (M // tokens, H, d, tokens)]
        outs_h = [torch.full(sh, float("nan"), dtype=T, device="cuda") for sh in shapes]
        outs_e = [torch.zeros(sh, dtype=T) for sh in shapes]
        hd = dict(seg_cols=N // 3, heads=H, tokens=tokens, transposed=[0, 0, 1], ld=[0, 0, tokens])
        kwh = {k: v for k, v in kw.items() if k not in ("ldo", "ldr")}
        hip.gemm(a.cuda(), w.cuda(), None, bias=bias.cuda(), epilogue=2, heads=dict(hd, outs=outs_h), tile=tile, **kwh)
        torch.cuda.synchronize()
        emu.gemm(a, w, None, bias=bias, epilogue=2, heads=dict(hd, outs=outs_e), **kwh)
        for i, (h_, e_) in enumerate(zip(outs_h, outs_e)):
            close(h_, e_, f"pp heads seg {i} tile {tile}", RTOL["bf16"])
        return
    o_h = torch.full((M, kw["ldo"]), float("nan"), dtype=T, device="cuda")
    hip.gemm(a.cuda(), w.cuda(), o_h, bias=bias.cuda(), tile=tile, **dev, **kw)
    torch.cuda.synchronize()
    o_e = torch.zeros(M, kw["ldo"], dtype=T)
    emu.gemm(a, w, o_e, bias=bias, **ex, **kw)
    close(o_h, o_e, f"pp gemm {feat} {M}x{N}x{K} tile {tile}", RTOL["bf16"])


@pytest.mark.parametrize("tile", PP_TILES)
@pytest.mark.parametrize("mode,stride,frames,H,W,Cin,Cout", [
    (1, 1, 9, 64, 64, 64, 320),      # 36 864 rows: several tiles per workgroup, 9 K tiles (one slab)
    (1, 1, 3, 40, 24, 192, 640),     # rows per frame not a multiple of the tile: frame borders inside a DMA piece
    (1, 2, 4, 32, 32, 128, 320),     # stride-2 downsample
    (2, 1, 3, 16, 16, 128, 640),     # nearest-2x upsample folded into the gather
    (3, 1, 2, 13, 9, 64, 320),       # upsample to a forwarded odd size
])
def test_gemm_pingpong_conv(hip, emu, tile, mode, stride, frames, H, W, Cin, Cout):
    T = torch.bfloat16
    if mode == 3:
        mode, Ho, Wo = 2, 2 * H - 1, 2 * W - 1
    else:
        Ho, Wo = (2 * H, 2 * W) if mode == 2 else ((H - 1) // stride + 1, (W - 1) // stride + 1)
    M, K = frames * Ho * Wo, 9 * Cin
    x, w = rnd((frames * H * W, Cin), T, 1), rnd((Cout, K), T, 2, 1 / math.sqrt(K))
    bias, rowb, res = rnd((Cout,), torch.float32, 3), rnd((frames, Cout), torch.float32, 5), rnd((M, Cout), T, 6)
    conv = dict(Hout=Ho, Wout=Wo, Hin=H, Win=W, Cin=Cin, stride=stride)
    kw = dict(M=M, N=Cout, K=K, lda=Cin, ldw=K, ldo=Cout, ldr=Cout, mode=mode, conv=conv, rows_per_batch=Ho * Wo)
    o_h = torch.full((M, Cout), float("nan"), dtype=T, device="cuda")
    hip.gemm(x.cuda(), w.cuda(), o_h, bias=bias.cuda(), rowbias=rowb.cuda(), residual=res.cuda(), tile=tile, **kw)
    torch.cuda.synchronize()
    o_e = torch.zeros(M, Cout, dtype=T)
    emu.gemm(x, w, o_e, bias=bias, rowbias=rowb, residual=res, **kw)
    close(o_h, o_e, f"pp conv mode={mode} s={stride} {frames}x{H}x{W} {Cin}->{Cout} tile {tile}", RTOL["bf16"])


@pytest.mark.parametrize("tile", [31, 6, 5])
@pytest.mark.parametrize("kind,M,N,K,feat", [
    ("gemm", 40064, 640, 640, "res+stats"),            # 313 row tiles x 2 column tiles: several tiles per workgroup, statistics published under the next tile
    ("gemm", 33000, 1920, 640, "ln+rb256"),            # LayerNorm fold + a per-frame row bias that is uniform per tile
    ("gemm", 8192, 3840, 1280, "ln+rb64"),             # ... and one that changes inside a tile (two rows per 128-row tile)
    ("gemm", 20096, 320, 1600, "dual+res+stats"),      # merged FF2 | proj_out: dual-source K
    ("conv", 36864, 320, 576, "res+rb+stats"),         # 3x3 conv, time-embedding row + residual + statistics (9 K tiles)
    ("gemm", 2050, 320, 320, "res+stats"),             # exactly 5 K tiles (the shortest stream the kernel takes), ragged last row tile
])
def test_gemm_overlapped_epilogue(hip, emu, tile, kind, M, N, K, feat):
    """csrc/gemm_ov_kernel.h (tile config 31): the previous tile's epilogue - pack, four store steps, statistics reduction and
    publication - runs under the K loop of the next tile.  Same inputs through the one-phase kernels (tiles 6 / 5) as a cross-check
    of the test itself; outputs against the specification, statistics against sums of the stored values."""
    T = torch.bfloat16
    w, bias = rnd((N, K), T, 2, 1 / math.sqrt(K)), rnd((N,), torch.float32, 3)
    kw = dict(M=M, N=N, K=K, lda=K, ldw=K, ldo=N, ldr=N)
    ex = {}
    cs_rows = 0
    if kind == "conv":
        Cin, side = K // 9, 64
        a = rnd((M, Cin), T, 1)
        kw.update(lda=Cin, mode=1, conv=dict(Hout=side, Wout=side, Hin=side, Win=side, Cin=Cin, stride=1))
    elif "dual" in feat:
        k2 = 320
        a = rnd((M, K - k2), T, 1)
        ex.update(a2=rnd((M, k2), T, 7), k_split=K - k2, lda2=k2)
        kw["lda"] = K - k2
    else:
        a = rnd((M, K), T, 1)
    if "res" in feat:
        ex["residual"] = rnd((M, N), T, 4)
    if "ln" in feat:
        ex["ln_stats"] = torch.stack([rnd((M,), torch.float32, 8) * 0.1, rnd((M,), torch.float32, 9).abs() + 0.5], dim=1).contiguous()
        ex["ln_colsum"] = rnd((N,), torch.float32, 10)
    for key, rpb in (("rb256", 256), ("rb64", 64), ("rb+", 4096)):
        if key in feat:
            ex.update(rowbias=rnd(((M + rpb - 1) // rpb, N), torch.float32, 5), rows_per_batch=rpb)
    dev = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in ex.items()}
    parts = None
    if "stats" in feat:
        cs_rows = 128 if M % 128 == 0 else 0
        if cs_rows == 0:                      # ragged M: statistics samples must tile the rows; use the largest divisor that is a multiple of 16
            cs_rows = next(c for c in (2048, 1024, 512, 256, 128, 64, 32, 16) if M % c == 0) if any(M % c == 0 for c in (2048, 1024, 512, 256, 128, 64, 32, 16)) else 0
    o_h = torch.full((M, N), float("nan"), dtype=T, device="cuda")
    if cs_rows:
        nt, tile_rows, slots = hip.gemm_stat_layout(T, M=M, N=N, K=K, cs_rows=cs_rows, mode=kw.get("mode", 0), tile=tile)
        if not 1 <= slots <= 4:
            cs_rows = 0
        else:
            parts = torch.full((nt * slots * N * 2,), float("nan"), device="cuda")
    hip.gemm(a.cuda(), w.cuda(), o_h, bias=bias.cuda(), tile=tile, chan_parts=parts, cs_rows=cs_rows, **dev, **kw)
    torch.cuda.synchronize()
    o_e = torch.zeros(M, N, dtype=T)
    emu.gemm(a, w, o_e, bias=bias, **ex, **kw)
    close(o_h, o_e, f"ov {kind} {feat} {M}x{N}x{K} tile {tile}", RTOL["bf16"])
    if cs_rows:
        cs = torch.zeros(M // cs_rows, N, 2, dtype=torch.float64, device="cuda")
        hip.chan_stats_reduce(parts, cs, rows=M, N=N, cs_rows=cs_rows, tile_rows=tile_rows, slots=slots)
        v = o_h.double().reshape(M // cs_rows, cs_rows, N)
        close(cs, torch.stack([v.sum(dim=1), (v * v).sum(dim=1)], dim=-1).cpu(), f"ov chan stats {kind} {feat} tile {tile}", 2e-6)


def test_gemm_pingpong_is_repeatable(hip):
    """the ping-pong loop hands LDS stages between wave groups by barrier counting: a misplaced wait shows as launch-to-launch
    differences under cold / warm caches, not as a tolerance failure - 12 launches per shape must be bitwise equal"""
    T = torch.bfloat16
    for tile in PP_TILES:
        for (M, N, K) in ((131072, 320, 320), (8192, 1280, 1280)):
            a, w, r = rnd((M, K), T, 1).cuda(), rnd((N, K), T, 2, 1 / math.sqrt(K)).cuda(), rnd((M, N), T, 4).cuda()
            junk = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
            ref = None
            for it in range(12):
                if it % 3 == 0:
                    junk.fill_(float(it))              # evict L2 / Infinity Cache
                o = torch.empty(M, N, dtype=T, device="cuda")
                hip.gemm(a, w, o, M=M, N=N, K=K, lda=K, ldw=K, ldo=N, ldr=N, residual=r, tile=tile)
                torch.cuda.synchronize()
                if ref is None:
                    ref = o
                else:
                    assert torch.equal(o.view(torch.int16), ref.view(torch.int16)), f"tile {tile} {M}x{N}x{K}: launch {it} differs from launch 0"


def test_gemm_pingpong_split_k(hip, emu):
    """split-K work items through the ping-pong loop (fyc_set_tuning key 9 = 2 routes the library's own choices to it)"""
    T = torch.bfloat16
    M, N, K = 2048, 1280, 6400
    a, w, bias, r = rnd((M, K), T, 1), rnd((N, K), T, 2, 1 / math.sqrt(K)), rnd((N,), torch.float32, 3), rnd((M, N), T, 4)
    hip.set_tuning(9, 2)
    try:
        o_h = torch.full((M, N), float("nan"), dtype=T, device="cuda")
        hip.gemm(a.cuda(), w.cuda(), o_h, M=M, N=N, K=K, lda=K, ldw=K, ldo=N, ldr=N, bias=bias.cuda(), residual=r.cuda())
        torch.cuda.synchronize()
    finally:
        hip.set_tuning(9, 0)
    o_e = torch.zeros(M, N, dtype=T)
    emu.gemm(a, w, o_e, M=M, N=N, K=K, lda=K, ldw=K, ldo=N, ldr=N, bias=bias, residual=r)
    close(o_h, o_e, "pp split-K", RTOL["bf16"])
