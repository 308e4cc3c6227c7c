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


def test_lane_swap_maximum_keeps_both_halves(tmp_path):
    """hipcc 7.2 simplifies `fmaxf(r[0], r[1])` of `__builtin_amdgcn_permlane32_swap(x, x, ...)` to `r[0]`: the attention kernel's running
    maximum then ignored the keys held by lanes 32..63 (round 4: infinite probabilities in the f16 form, hidden by bf16's exponent range
    before that).  fyc_common.h::swap32_max / swap16_max make the two results opaque; this checks the generated code: every
    v_permlane*_swap of the helper is followed by a v_max that reads BOTH of its registers - and documents the miscompiled plain form,
    so that a compiler which no longer needs the workaround shows up here as well."""
    src = tmp_path / "swapmax.hip"
    src.write_text('#include "%s"\n' % os.path.join(CSRC, "fyc_common.h") + '''
__global__ void fixed(const float* in, float* out) { out[threadIdx.x] = swap32_max(swap16_max(in[threadIdx.x])); }
__global__ void fsum(const float* in, float* out) { out[threadIdx.x] = swap32_sum(swap16_sum(in[threadIdx.x])); }
__global__ void plain(const float* in, float* out) {          // the attention kernel's quad_max as it was written in rounds 1-4
  float mx = in[threadIdx.x];
  mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, mx), 0x401F)));
  const unsigned u = __builtin_bit_cast(unsigned, mx);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  out[threadIdx.x] = fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
}
''')
    out = tmp_path / "swapmax.s"
    r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", str(src), "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    bodies = _kernel_bodies(out.read_text())
    fixed = next(v for k, v in bodies.items() if "fixed" in k)
    swaps = list(re.finditer(r"v_permlane(16|32)_swap_b32\w*\s+(v\d+),\s*(v\d+)", fixed))
    assert len(swaps) == 2, fixed
    for m in swaps:
        tail = fixed[m.end():]
        nxt = re.search(r"v_permlane(16|32)_swap", tail)
        tail = tail[:nxt.start()] if nxt else tail
        a, b = m.group(2), m.group(3)
        # (canonicalising v_max x, x, x first, then the max of the two)
        assert re.search(rf"v_max_f32\w*\s+v\d+,\s*({a},\s*{b}|{b},\s*{a})\b", tail), f"the maximum after {m.group(0)} does not combine both results:\n{fixed}"
    # the sums: same helpers, same requirement - the v_add after each swap reads BOTH of its registers (round-4 advice: with the operand
    # held in one `unsigned`, hipcc 7.2 emits `v_add_f32 v1, v1, v1` = twice one half)
    fsum = next(v for k, v in bodies.items() if "fsum" in k)
    swaps = list(re.finditer(r"v_permlane(16|32)_swap_b32\w*\s+(v\d+),\s*(v\d+)", fsum))
    assert len(swaps) == 2, fsum
    for m in swaps:
        tail = fsum[m.end():]
        nxt = re.search(r"v_permlane(16|32)_swap", tail)
        tail = tail[:nxt.start()] if nxt else tail
        a, b = m.group(2), m.group(3)
        assert re.search(rf"v_add_f32\w*\s+v\d+,\s*({a},\s*{b}|{b},\s*{a})\b", tail), f"the sum after {m.group(0)} does not combine both results:\n{fsum}"
    plain = next(v for k, v in bodies.items() if "plain" in k)
    m = re.search(r"v_permlane32_swap_b32\w*\s+(v\d+),\s*(v\d+)", plain)
    combined = m and re.search(rf"v_max_f32\w*\s+v\d+,\s*({m.group(1)},\s*{m.group(2)}|{m.group(2)},\s*{m.group(1)})\b", plain[m.end():])
    if combined:
        pytest.skip("this hipcc compiles the plain fmaxf form correctly: the asm barrier in swap32_max / swap16_max is no longer needed")
    # (hipcc 7.2.0: the plain form stores the swap's first result unreduced - the defect the helpers exist for)


def test_attention_issue_statement_restores_m0_and_exec(tmp_path):
    """round 5: the d <= 48 attention kernels issue the DMAs of a full K / V^T tile from ONE asm statement that rewrites EXEC (precomputed lane
    masks) and M0 (LDS destination).  Both are compiler-owned: the statement must save them first and restore them last, every masked DMA must
    sit behind its `s_cbranch_execz` (an instruction with an empty mask would not count in vmcnt and break the counted waits), and the
    sNaN-quieting `v_max_f32 x, x, x` the no-NaN build flag exists to remove must be gone from the loop."""
    asm = _device_asm("attention_small.hip", tmp_path)
    kernels = {k: v for k, v in _kernel_bodies(asm).items() if "fyc_attn_kernel" in k}
    assert kernels
    checked = 0
    for name, body in kernels.items():
        for stmt in re.findall(r";;#ASMSTART(.*?);;#ASMEND", body, flags=re.S):
            if "global_load_lds" not in stmt or "s_mov_b64 exec" not in stmt:
                continue
            lines = [l.strip() for l in stmt.strip().split("\n") if l.strip()]
            m0 = re.match(r"s_mov_b32 (s\d+), m0", lines[0])
            ex = re.match(r"s_mov_b64 (s\[\d+:\d+\]), exec", lines[1])
            assert m0 and ex, f"{name}: the statement does not start by saving M0 and EXEC:\n{stmt}"
            assert lines[-2] == f"s_mov_b64 exec, {ex.group(1)}" and lines[-1] == f"s_mov_b32 m0, {m0.group(1)}", f"{name}: M0 / EXEC not restored:\n{stmt}"
            dmas = [i for i, l in enumerate(lines) if l.startswith("global_load_lds_dwordx4")]
            assert len(dmas) == 4
            for i in dmas:
                assert lines[i - 1] == "s_nop 0" and lines[i - 2].startswith("s_mov_b32 m0,") and lines[i - 3].startswith("s_cbranch_execz") \
                    and lines[i - 4].startswith("s_mov_b64 exec,"), f"{name}: DMA not behind mask / branch / M0 / wait state:\n{stmt}"
            checked += 1
        canon = re.findall(r"v_max_f32_e32 (v\d+), (v\d+), (v\d+)", body)
        assert not [c for c in canon if c[1] == c[2]], f"{name}: sNaN-quieting v_max_f32 x, x, x is back (is -fno-honor-nans still on the attention sources?)"
        assert "s_setprio" not in body, f"{name}: s_setprio in an issue-bound loop (-DFYC_ATTN_SETPRIO builds it)"
    assert checked >= len(kernels), (checked, len(kernels))      # at least one such statement per kernel (prologue + loop)


def test_reciprocal_division_of_the_head_split_epilogue_is_exact():
    """csrc/gemm_kernel.h::fdiv_small (round 6): floor(n / d) as (int)((n + 0.5f) * (1.0f / d)) replaces the run-time integer divisions of the
    head-split epilogue's index arithmetic.  Exhaustive over the range fyc_gemm admits (N < 2^16 columns, segment width / head dim <= 2^12):
    the f32 formula, evaluated with numpy's IEEE single arithmetic as the GPU evaluates it (no contraction: -ffp-contract=off), never differs
    from the integer quotient."""
    import numpy as np
    n = np.arange(65536, dtype=np.int64)
    nf = n.astype(np.float32) + np.float32(0.5)
    for d in range(1, 4097):
        q = (nf * (np.float32(1.0) / np.float32(d))).astype(np.int64)
        assert np.array_equal(q, n // d), d
