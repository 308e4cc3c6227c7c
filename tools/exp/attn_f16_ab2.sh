#!/bin/bash
# round 4 diagnostic (historic): spike height x position sweep of the f16 attention, and the same on a build whose probabilities were
# clamped to 60000 (macro FYC_ATTN_F16_CLAMP, removed again).  Result: NaN rows exactly where a key of lane quad 2 / 3 (key % 32 >= 16)
# beats the running maximum by >= 2^16; the clamped build is finite -> overflowing probabilities, i.e. the maximum did not move ->
# quad_max's cross-half step was folded away by hipcc (fyc_common.h::swap32_max).
cd "$GRAFT_REPO_ROOT"
python tools/exp/attn_f16_diag2.py 2>&1 | grep -v amdgpu.ids
echo "== variant CLAMP"
FYC_LIB_PATH=$GRAFT_REPO_ROOT/tools/exp/libfyc_CLAMP.so python tools/exp/attn_f16_diag2.py 2>&1 | grep -v amdgpu.ids | grep "key 440\|key 200"
