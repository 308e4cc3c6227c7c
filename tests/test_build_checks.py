"""Static checks on the generated gfx950 code (no GPU): properties that parity tests cannot see.

The register-resident kernels issue their LDS-DMA from inline asm, invisible to hipcc's wait-count bookkeeping.  Round 3 found what
that costs when the compiler places an LDS-queue instruction of its own (`ds_bpermute_b32` from a `__shfl_xor`) between those DMAs
and waits for it with a COUNTED lgkmcnt: a launch-to-launch race inside every tolerance (profiles/r03_ff_block_race.txt).  The rule
since then - no LDS-queue instruction other than fragment reads / staging writes in these kernels - is checked here on the
assembly, so that a compiler upgrade cannot bring the race back silently.  Also: the GEMM kernels' hot loops must not
touch scratch (a reload is a vmcnt(0) in the middle of the DMA stream)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "followyourclick_amd", "csrc")


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    pytest.skip("hipcc not available")


def _device_asm(src: str, tmp_path) -> str:
    from followyourclick_amd import _build
    out = tmp_path / (src + ".s")
    cmd = [_hipcc(), *_build._flags(src), "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return out.read_text()


def _kernel_bodies(asm: str):
    """{symbol: text} per function of the assembly file"""
    bodies, cur, name = {}, [], None
    for line in asm.split("\n"):
        m = re.match(r"^(_Z\w+):\s", line)
        if m:
            if name:
                bodies[name] = "\n".join(cur)
            name, cur = m.group(1), []
        elif name:
            cur.append(line)
    if name:
        bodies[name] = "\n".join(cur)
    return bodies


@pytest.mark.parametrize("src", ["ff_block.hip", "temporal_block_rr.hip", "panel_linear.hip"])
def test_no_compiler_lds_queue_traffic_between_asm_dmas(src, tmp_path):
    asm = _device_asm(src, tmp_path)
    kernels = {k: v for k, v in _kernel_bodies(asm).items() if "global_load_lds" in v}
    assert kernels, f"{src}: no kernel with LDS-DMA found"
    for name, body in kernels.items():
        bad = re.findall(r"^\s*(ds_bpermute_b32|ds_permute_b32|ds_swizzle_b32)\b", body, flags=re.M)
        assert not bad, f"{src}::{name}: {len(bad)} x {set(bad)} - cross-lane traffic through the LDS queue next to asm-issued DMA (use v_permlane*_swap)"
        # every statement that writes M0 for a DMA restores it
        for stmt in re.findall(r";;#ASMSTART(.*?);;#ASMEND", body, flags=re.S):
            if "global_load_lds" in stmt and re.search(r"s_mov_b32 m0,", stmt):
                assert re.search(r"s_mov_b32 (s\d+|vcc_lo|vcc_hi|ttmp\d+), m0", stmt) and len(re.findall(r"s_mov_b32 m0,", stmt)) >= 2, f"{src}::{name}: a DMA statement leaves M0 modified:\n{stmt}"
