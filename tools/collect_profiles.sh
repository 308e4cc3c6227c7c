#!/bin/bash
# Measurement set of a round (run on the GPU box: gpurun -- 'ROUND=r05 bash tools/collect_profiles.sh').  Writes under gpurun_out/<round>; the
# summaries are then copied to profiles/ (see profiles/README.md).  PMC passes are separate runs (no trace domains beside them).
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUND=${ROUND:-r04}
O=$R/gpurun_out/$ROUND
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# HBM bytes per launch first: bench.py reports them as roofline.traffic when the profile is of the library it runs (digest-stamped)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python $R/tools/hbm_traffic.py $O/pmc_fetch $O/pmc_write $O/hbm_traffic.json > $O/hbm_traffic.txt
cp $O/hbm_traffic.json $R/profiles/${ROUND}_hbm_traffic.json
python $R/bench.py --steps 5 --warmup 2 --vae > $O/bench_final.json 2> $O/bench_final.err
python $R/bench.py --steps 5 --warmup 2 --dtype f16 --vae --no-cpu-baseline > $O/bench_f16.json 2> $O/bench_f16.err
python $R/bench.py --steps 3 --warmup 1 --graph --no-cpu-baseline --no-roofline > $O/bench_graph.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/ktrace -o kt -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python $R/tools/rocprof_summary.py $(find $O/ktrace -name "*.db" | head -1) $O/kernel_stats.txt
rocprofv3 --kernel-trace --stats -d $O/ktrace_vae -o kv -- python $R/tools/vae_only.py 16 512 2 > $O/vae_only.txt 2>&1
python $R/tools/rocprof_summary.py $(find $O/ktrace_vae -name "*.db" | head -1) $O/vae_kernel_stats.txt
# the other BASELINE.json configurations: labels follow the arguments, every line carries its roofline
python $R/bench.py --steps 2 --warmup 1 --frames 8 --size 256 --ddim-steps 5 --no-cpu-baseline > $O/bench_cfg1.json 2>/dev/null
python $R/bench.py --steps 1 --warmup 1 --frames 32 --size 768 --ddim-steps 50 --no-cpu-baseline > $O/bench_cfg4.json 2>/dev/null
python $R/bench.py --steps 2 --warmup 1 --ip-tokens 16 --no-cpu-baseline > $O/bench_cfg5.json 2>/dev/null
# raw traces / counter CSVs stay on the box (gpurun merges at most 64 MiB back): the summaries above are what profiles/ keeps
rm -rf $O/ktrace $O/ktrace_vae $O/pmc_fetch $O/pmc_write 2>/dev/null
ls -la $O
