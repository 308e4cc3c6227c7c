#!/bin/bash
cd "$GRAFT_REPO_ROOT"
./tools/exp/f16_denorm_probe 2>&1 | tail -45
for v in FLUSH SELECT SCALAR_CVT; do
  echo "== variant $v"
  FYC_LIB_PATH=$GRAFT_REPO_ROOT/tools/exp/libfyc_$v.so FYC_DIAG_SHORT=1 python tools/exp/attn_f16_diag.py 2>&1 | grep -v amdgpu.ids | tail -8
done
