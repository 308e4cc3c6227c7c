#!/bin/bash
# round-3 GPU call A: fused FF kernel - parity, probe, engine parity, short bench
mkdir -p gpurun_out
export FYC_SKIP_REBUILD=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "ff_block" 2>&1 | tail -15 > gpurun_out/a_ffblock_tests.txt
cat gpurun_out/a_ffblock_tests.txt
timeout 300 python tools/ff_probe.py > gpurun_out/a_ff_probe.txt 2>&1
cat gpurun_out/a_ff_probe.txt
timeout 900 python -m pytest tests/test_fullwidth_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/a_fullwidth.txt
cat gpurun_out/a_fullwidth.txt
FYC_BENCH_SHAPES=gpurun_out/a_shapes.txt timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err
cat gpurun_out/a_bench.json
FYC_FUSE_FF=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/a_bench_nofuse.json 2>> gpurun_out/a_bench.err
cat gpurun_out/a_bench_nofuse.json
