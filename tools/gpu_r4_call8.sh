#!/bin/bash
# round 4, GPU call 8: 4-wave big tiles (configs 41 = 256x256, 42 = 256x320; csrc/gemm_w4.hip): parity and cold-operand sweep
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4; mkdir -p $O
PROBE_SWEEP=1 PROBE_CFGS=0,5,6,7,41,42 timeout 500 python tools/gemm_probe.py > $O/c8_probe.txt 2>&1; tail -24 $O/c8_probe.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "gemm or conv or statistics" > $O/c8_tests.txt 2>&1; tail -8 $O/c8_tests.txt
