#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "ff_block" 2>&1 | tail -30 > $O/g_ff_only.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -q 2>&1 | grep -v "^$" | tail -60 > $O/g_kernels.txt
for i in 1 2 3; do timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "ff_block_matches" 2>&1 | tail -3; done > $O/g_repeat.txt
tail -5 $O/g_ff_only.txt; grep -n "AssertionError\|rel-L2\|passed\|failed" $O/g_kernels.txt | head; cat $O/g_repeat.txt
