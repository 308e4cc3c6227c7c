#!/bin/bash
# round-3 GPU call F: full GPU suite on the current library, eager vs hipGraph replay on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/f_gputests.txt; cat $O/f_gputests.txt
timeout 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $O/f_bench_eager.json 2> $O/f_bench.err; cat $O/f_bench_eager.json
timeout 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --graph > $O/f_bench_graph.json 2>> $O/f_bench.err; cat $O/f_bench_graph.json
timeout 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $O/f_bench_eager2.json 2>> $O/f_bench.err; cat $O/f_bench_eager2.json
