"""Stages the reference's OWN entry scripts and inference YAMLs under the git-ignored oracle/_ref/ (TEST INFRASTRUCTURE; container
only - run by `__graft_entry__.build()` wherever /root/reference exists):

    python -m oracle.stage_ref_scripts

The GPU box has no /root/reference, but it receives oracle/_ref/ with the repo snapshot (git-ignored, not gpurun-ignored), so
tests/test_script_dropin.py can execute `scripts/inference*.py` UNMODIFIED on the HIP kernels there ("scripts/
inference*.py drop in unchanged").  Round 5 also stages the reference's model code: oracle/gpu_reference.py imports it ON THE GPU BOX
through oracle/refshim.py to run the real UNet3DConditionModel under torch.autocast("cuda") on the MI355X itself - the same-device,
same-precision yardstick of tests/test_reference_gpu.py and the `gpu_reference` / `cpu_baseline` legs of bench.py.  Round 6: only the
IMPORT CLOSURE of that yardstick is staged (the modules a subprocess ends up with after building the reference UNet3D + DDIMScheduler:
~30 files), not the whole vendored `diffusers/` tree (170 files, 61 pipelines).  Nothing under oracle/_ref/ is ever committed, imported by the product, or edited: the files
are byte copies (sha256 listed in oracle/_ref/MANIFEST.json), executed through runpy on top of followyourclick_amd.install_dropin().
"""
import hashlib
import json
import os
import shutil

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
DST = os.path.join(HERE, "_ref")
FILES = ["scripts/inference.py", "scripts/inference_org.py", "scripts/inference_w_image_cond.py", "scripts/inference_w_camera_lora.py",
         "scripts/animate.py", "configs/inference/inference_img_embed_mask_condition_zero_snr_.yaml",
         "configs/prompts/0-StableDiffusion_zero_snr_sd1.5_448x256.yaml"]
TREES = ["animatediff", "diffusers", "ip_adapter"]      # packages the same-device yardstick imports FROM (only its import closure is staged)

_CLOSURE = r"""
import json, os, sys
sys.path.insert(0, {root!r})
os.environ["FYC_REFERENCE_ROOT"] = {ref!r}
from oracle import gpu_reference, refshim
from oracle import functional as Fn
gpu_reference.install_sdpa_xformers()
refshim.install()
import diffusers.utils.import_utils as iu
iu._xformers_available = True
from oracle.make_golden import ref_unet
from diffusers.schedulers.scheduling_ddim import DDIMScheduler
for cfg in (Fn.tiny_unet_config(), Fn.tiny_unet_config(use_ip_cross_attention=True)):      # construction runs the lazy imports
    u = ref_unet(cfg)      # (enable_xformers_memory_efficient_attention needs a GPU; it lives in modules that are imported by now)
DDIMScheduler(**gpu_reference.SCHED_KW)
ref = os.path.realpath({ref!r}) + os.sep
out = sorted(os.path.relpath(os.path.realpath(m.__file__), ref) for m in list(sys.modules.values())
             if getattr(m, "__file__", None) and os.path.realpath(m.__file__).startswith(ref))
print("CLOSURE " + json.dumps(out))
"""


def import_closure():
    """reference files (relative paths) the same-device yardstick imports, found by running its imports in a subprocess"""
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, "-c", _CLOSURE.format(root=root, ref=REF)], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    for line in r.stdout.splitlines():
        if line.startswith("CLOSURE "):
            return json.loads(line[8:])
    raise RuntimeError("import closure of the reference yardstick failed:\n" + r.stdout[-2000:] + r.stderr[-4000:])


def stage() -> int:
    if not os.path.isdir(REF):
        return 0
    man = {}
    files = list(FILES)
    closure = import_closure()
    files += [f for f in closure if f.split(os.sep)[0] in TREES]
    for tree in TREES:                                   # a previous (wider) staging must not linger
        shutil.rmtree(os.path.join(DST, tree), ignore_errors=True)
    for rel in files:
        src = os.path.join(REF, rel)
        if not os.path.exists(src):
            continue
        dst = os.path.join(DST, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(src, dst)
        with open(dst, "rb") as f:
            man[rel] = hashlib.sha256(f.read()).hexdigest()
    with open(os.path.join(DST, "MANIFEST.json"), "w") as f:
        json.dump(dict(source=REF, files=man), f, indent=1)
    return len(man)


if __name__ == "__main__":
    print(f"staged {stage()} reference files under {DST}")
