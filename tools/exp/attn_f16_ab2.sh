#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python tools/exp/attn_f16_diag2.py 2>&1 | grep -v amdgpu.ids
echo "== variant CLAMP"
FYC_LIB_PATH=$GRAFT_REPO_ROOT/tools/exp/libfyc_CLAMP.so python tools/exp/attn_f16_diag2.py 2>&1 | grep -v amdgpu.ids | grep "key 440\|key 200"
