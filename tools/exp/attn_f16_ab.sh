#!/bin/bash
# round 4 diagnostic (historic): the f16 attention's NaN rows under three variant builds of attention_kernel.h (f16 denormals flushed through
# MODE, probabilities below 2^-14 forced to zero, scalar instead of packed conversion - macros FYC_ATTN_F16_FLUSH / _SELECT / _SCALAR_CVT,
# removed from the kernel again after this run) and the denormal probe.  Result: all three variants failed identically, the probe showed
# v_cvt_pk_f16_f32 and v_mfma_*_f16 handling denormals correctly -> not a denormal problem ("Spatial attention").
cd "$GRAFT_REPO_ROOT"
./tools/exp/f16_denorm_probe 2>&1 | tail -45
for v in FLUSH SELECT SCALAR_CVT; do
  echo "== variant $v"
  FYC_LIB_PATH=$GRAFT_REPO_ROOT/tools/exp/libfyc_$v.so FYC_DIAG_SHORT=1 python tools/exp/attn_f16_diag.py 2>&1 | grep -v amdgpu.ids | tail -8
done
