#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03c; mkdir -p $O
python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/bench_eager.json 2>/dev/null
python $R/bench.py --steps 3 --warmup 1 --graph --no-cpu-baseline --no-roofline > $O/bench_graph.json 2>/dev/null
python $R/bench.py --steps 2 --warmup 1 --frames 8 --size 256 --ddim-steps 5 --no-cpu-baseline --no-roofline > $O/bench_cfg1.json 2>/dev/null
python $R/bench.py --steps 1 --warmup 1 --frames 32 --size 768 --ddim-steps 50 --no-cpu-baseline --no-roofline > $O/bench_cfg4.json 2>/dev/null
python $R/bench.py --steps 2 --warmup 1 --ip-tokens 16 --no-cpu-baseline --no-roofline > $O/bench_cfg5.json 2>/dev/null
for f in eager graph cfg1 cfg4 cfg5; do python -c "
import json; d=json.load(open('$O/bench_$f.json')); print('$f', d['value'], d['ms_per_step'])"; done
