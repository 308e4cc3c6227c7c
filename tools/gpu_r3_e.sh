#!/bin/bash
# round-3 GPU call E: panel linear - parity, probe, then full-width parity and bench with everything on
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "panel_linear" 2>&1 | tail -12 > $O/e_tests.txt; cat $O/e_tests.txt
timeout 300 python tools/panel_probe.py > $O/e_panel_probe.txt 2>&1; cat $O/e_panel_probe.txt
timeout 1200 python -m pytest tests/test_fullwidth_gpu.py -x -q 2>&1 | tail -6 > $O/e_fullwidth.txt; cat $O/e_fullwidth.txt
FYC_BENCH_SHAPES=$O/e_shapes.txt timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/e_bench.json 2> $O/e_bench.err; cat $O/e_bench.json
FYC_FUSE_PANEL=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/e_bench_nopanel.json 2>> $O/e_bench.err; cat $O/e_bench_nopanel.json
