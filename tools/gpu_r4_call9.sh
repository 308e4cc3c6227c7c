#!/bin/bash
# round 4, GPU call 9: hand-pipelined 4-wave 256x256 tile (config 41, csrc/gemm_w4_kernel.h): parity and cold-operand sweep
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "gemm and not conv" > $O/c9_tests.txt 2>&1; tail -8 $O/c9_tests.txt
PROBE_SWEEP=1 PROBE_CFGS=0,5,6,7,41 timeout 500 python tools/gemm_probe.py > $O/c9_probe.txt 2>&1; head -18 $O/c9_probe.txt
