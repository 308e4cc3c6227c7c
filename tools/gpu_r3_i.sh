#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_fullwidth_gpu.py -q -x 2>&1 | tail -5 > $O/i_fullwidth.txt; cat $O/i_fullwidth.txt
timeout 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/i_bench.json 2> $O/i_bench.err; cat $O/i_bench.json
FYC_TEMPORAL_RR=0 timeout 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $O/i_bench_nrr.json 2>> $O/i_bench.err; cat $O/i_bench_nrr.json
