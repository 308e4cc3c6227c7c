#!/bin/bash
# round-3 GPU call D: attention with asm LDS-DMA (A/B against the builtin form, same box), GEGLU tile regression, ff_block tests
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "attention or geglu or ff_block" 2>&1 | tail -8 > $O/d_tests.txt; cat $O/d_tests.txt
for i in 1 2; do
  echo "--- asm DMA (default library)" >> $O/d_attn_ab.txt; timeout 200 python tools/attn_bench.py >> $O/d_attn_ab.txt 2>&1
  echo "--- builtin DMA (tools/exp/libfyc_hip_attn_builtin.so)" >> $O/d_attn_ab.txt; FYC_LIB_PATH=$R/tools/exp/libfyc_hip_attn_builtin.so timeout 200 python tools/attn_bench.py >> $O/d_attn_ab.txt 2>&1
done
cat $O/d_attn_ab.txt
