"""Builds libfyc_hip.so (gfx950) in-tree with hipcc.  `python -m followyourclick_amd._build`.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so travels to
the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored)."""
import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# A/B builds (tools/): FYC_BUILD_EXTRA="-DFYC_ATTN_BUILTIN_DMA" FYC_BUILD_LIB=tools/exp/libfyc_hip_x.so python -m followyourclick_amd._build
EXTRA = os.environ.get("FYC_BUILD_EXTRA", "").split()
LIB = os.path.abspath(os.environ.get("FYC_BUILD_LIB") or os.path.join(HERE, "libfyc_hip.so"))
OBJ = os.path.join(HERE, "_obj") if not (EXTRA or os.environ.get("FYC_BUILD_LIB")) else LIB + ".obj"
SOURCES = ["api.hip", "gemm.hip", "gemm_bf16_plain.hip", "gemm_bf16_conv.hip", "gemm_bf16_act.hip", "gemm_f16_plain.hip", "gemm_f16_conv.hip", "gemm_f16_act.hip", "gemm_f32.hip", "attention.hip", "attention_small.hip", "attention_medium.hip", "attention_large.hip", "attention_small_f16.hip", "attention_medium_f16.hip", "attention_large_f16.hip", "temporal_attn.hip", "temporal_block.hip", "temporal_block_rr.hip", "ff_block.hip", "panel_linear.hip", "norm.hip", "elementwise.hip"]
# FYC_GEMM_VARIANTS=1: also build the round-4 main-loop experiments of tools/exp/gemm_variants/ (off by default, measured slower)
VARIANTS = os.environ.get("FYC_GEMM_VARIANTS") == "1"
VARIANT_DIR = os.path.join(HERE, "..", "tools", "exp", "gemm_variants")
VARIANT_SOURCES = ["gemm_pp_plain.hip", "gemm_pp_conv.hip", "gemm_ov.hip"] if VARIANTS else []
if VARIANTS:
    EXTRA = EXTRA + ["-DFYC_GEMM_VARIANTS"]
    if not os.environ.get("FYC_BUILD_LIB"):
        raise SystemExit("FYC_GEMM_VARIANTS=1 needs FYC_BUILD_LIB=<path>: the experiment kernels are not part of the product library")
# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs (gfx950 has a unified register file and every
# kernel here fits in 256 registers), which removes the v_accvgpr_read/write traffic around the
# softmax rescale and the epilogues (312 -> 0 such moves in the attention main loop).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-value",
         "-mllvm", "-amdgpu-mfma-vgpr-form=1"]
# ff_block.hip / panel_linear.hip / temporal_block_rr.hip run one wave per SIMD with the whole 512-register file: the accumulators must be AGPRs
AGPR_SOURCES = {"ff_block.hip", "panel_linear.hip", "temporal_block_rr.hip"}


# one wave per SIMD: a packed f32 VALU instruction beside MFMAs costs such a wave ~22 cycles more than the two scalar ones it replaces
# (MI355X_MICROARCH.md), and -O3's SLP vectoriser re-packs scalar f32 code into v_pk_*: off for these files
NO_SLP_SOURCES = set(filter(None, os.environ.get("FYC_NO_SLP", "ff_block.hip").split(",")))


# attention: no NaN ever enters the softmax (masked keys are -inf, scores are finite products of finite operands), and hipcc otherwise puts an
# sNaN-quieting `v_max_f32 x, x, x` in front of every running-maximum operand - 16 instructions per 64-key tile in a loop that is bound by
# instruction issue (profiles/r05_attention_pmc.txt)
NO_NAN_SOURCES = {s for s in SOURCES if s.startswith("attention")}


def _flags(src: str):
    fl = FLAGS
    if src in AGPR_SOURCES:
        fl = [f for f in FLAGS if f not in ("-mllvm", "-amdgpu-mfma-vgpr-form=1")]
    if src in NO_SLP_SOURCES:
        fl = fl + ["-fno-slp-vectorize"]
    if src in NO_NAN_SOURCES and os.environ.get("FYC_ATTN_HONOR_NANS") != "1":
        fl = fl + ["-fno-honor-nans"]
    return fl + EXTRA


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def _digest(path: str) -> str:
    h = hashlib.sha256()
    for dep in [path, os.path.join(CSRC, "fyc_common.h"), os.path.join(CSRC, "gemm_kernel.h"), os.path.join(CSRC, "attention_kernel.h"), os.path.join(CSRC, "attention_groups.h"), os.path.join(HERE, "..", "include", "fyc.h")]:
        with open(dep, "rb") as f:
            h.update(f.read())
    h.update(" ".join(_flags(os.path.basename(path))).encode())
    return h.hexdigest()


def source_digest() -> str:
    """sha256 over everything the kernels are compiled from (all csrc sources and headers, include/fyc.h, the compiler flags per
    source): the reproducible identity of the library.  hipcc's output is not bit-stable across output paths, so profiles taken
    on one build (tools/hbm_traffic.py) are matched to another build of the same sources through this, not through the .so"""
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    for f in files:
        h.update(f.encode())
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    with open(os.path.join(HERE, "..", "include", "fyc.h"), "rb") as fh:
        h.update(fh.read())
    for src in SOURCES:
        h.update((src + " " + " ".join(_flags(src))).encode())
    return h.hexdigest()


def family_digest(family: str = "gemm") -> str:
    """sha256 over what ONE kernel family is compiled from (its sources, the headers they include below csrc/, their flags): the
    identity a per-family profile is matched by - `roofline.traffic` is a fact about fyc_gemm_kernel and stays valid when another family's
    source (the attention kernel, round 5) changes after the PMC passes were taken.  include/fyc.h is left out on purpose: it changes with
    every ABI note; a change of the family's own argument struct shows up in the family's sources that use it."""
    if family != "gemm":
        raise ValueError(family)
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.startswith("gemm") and f.endswith((".hip", ".h"))) + ["fyc_common.h"]
    for f in files:
        h.update(f.encode())
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    for src in SOURCES:
        if src.startswith("gemm"):
            h.update((src + " " + " ".join(_flags(src))).encode())
    return h.hexdigest()


def _compile(src: str) -> str:
    path = os.path.join(VARIANT_DIR if src in VARIANT_SOURCES else CSRC, src)
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    stamp = obj + ".sha"
    dig = _digest(path)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj
    cmd = [_hipcc(), *_flags(src), "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
    with open(stamp, "w") as f:
        f.write(dig)
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with cf.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(_compile, SOURCES + VARIANT_SOURCES))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    if verbose:
        print(f"[fyc] built {LIB} ({os.path.getsize(LIB) >> 10} KiB)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
