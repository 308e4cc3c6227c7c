#!/bin/bash
# round 4, GPU call 7: is the K loop bound by the latency of the A operand?  Same shapes with every A row aliasing row 0 (lda = 0)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4; mkdir -p $O
PROBE_CFGS=5,6 FYC_LIB_PATH=tools/exp/libfyc_trace.so timeout 300 python tools/gemm_phase_probe.py > $O/c7_phase_cold.txt 2>&1
PROBE_LDA0=1 PROBE_CFGS=5,6 FYC_LIB_PATH=tools/exp/libfyc_trace.so timeout 300 python tools/gemm_phase_probe.py > $O/c7_phase_lda0.txt 2>&1
paste -d'\n' <(grep -v conv $O/c7_phase_cold.txt | cut -c1-120) <(grep -v conv $O/c7_phase_lda0.txt | cut -c1-120) | grep "loop/kt"
