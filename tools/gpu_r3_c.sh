#!/bin/bash
# round-3 GPU call C: fused FF in the pipeline - kernel parity, full-width parity, bench; diagnosis of the M=8192 GEMM
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
timeout 300 python tools/gemm_diag.py > $O/c_gemm_diag.txt 2>&1; cat $O/c_gemm_diag.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "ff_block" 2>&1 | tail -6 > $O/c_ffblock_tests.txt; cat $O/c_ffblock_tests.txt
timeout 300 python tools/ff_probe.py > $O/c_ff_probe.txt 2>&1; cat $O/c_ff_probe.txt
timeout 900 python -m pytest tests/test_fullwidth_gpu.py -x -q 2>&1 | tail -5 > $O/c_fullwidth.txt; cat $O/c_fullwidth.txt
FYC_BENCH_SHAPES=$O/c_shapes.txt timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/c_bench.json 2> $O/c_bench.err; cat $O/c_bench.json
