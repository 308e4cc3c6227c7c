"""Stages the reference's OWN entry scripts and inference YAMLs under the git-ignored oracle/_ref/ (TEST INFRASTRUCTURE; container
only - run by `__graft_entry__.build()` wherever /root/reference exists):

    python -m oracle.stage_ref_scripts

The GPU box has no /root/reference, but it receives oracle/_ref/ with the repo snapshot (git-ignored, not gpurun-ignored), so
tests/test_script_dropin.py can execute `scripts/inference*.py` UNMODIFIED on the HIP kernels there ("scripts/
inference*.py drop in unchanged").  Round 5 also stages the reference's model packages (`animatediff/`, `diffusers/`, `ip_adapter/`,
*.py only): oracle/gpu_reference.py imports them ON THE GPU BOX through oracle/refshim.py to run the real UNet3DConditionModel under
torch.autocast("cuda") on the MI355X itself - the same-device, same-precision yardstick of tests/test_reference_gpu.py and the
`gpu_reference` leg of bench.py.  Nothing under oracle/_ref/ is ever committed, imported by the product, or edited: the files
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
TREES = ["animatediff", "diffusers", "ip_adapter"]      # model code the same-device yardstick imports (python sources only)


def stage() -> int:
    if not os.path.isdir(REF):
        return 0
    man = {}
    files = list(FILES)
    for tree in TREES:
        for d, _, names in os.walk(os.path.join(REF, tree)):
            files += [os.path.relpath(os.path.join(d, n), REF) for n in sorted(names) if n.endswith(".py")]
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
