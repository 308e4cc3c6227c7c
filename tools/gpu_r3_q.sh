#!/bin/bash
# core of tools/collect_profiles.sh (bench line, kernel trace, PMC traffic) on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 5 --warmup 2 --vae > $O/bench_final.json 2> $O/bench_final.err
python -c "
import json; d=json.load(open('$O/bench_final.json')); print(d['value'], d['gpu_ms_per_ddim_step'], d['roofline']['frac'], d['roofline']['traffic'])"
rocprofv3 --kernel-trace --stats -d $O/ktrace -o kt -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python $R/tools/rocprof_summary.py $(find $O/ktrace -name "*.db" | head -1) $O/kernel_stats.txt
rm -rf $O/ktrace
head -3 $O/kernel_stats.txt
