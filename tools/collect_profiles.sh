#!/bin/bash
# Measurement set of a round (run on the GPU box: gpurun -- 'ROUND=r05 bash tools/collect_profiles.sh').  Writes under gpurun_out/<round>; the
# summaries are then copied to profiles/ (see profiles/README.md).  PMC passes are separate runs (no trace domains beside them).
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUND=${ROUND:-r05}
O=$R/gpurun_out/$ROUND
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# (--no-vae on the counter and trace passes: the VAE decode's 1-2 GB convolution launches are `fyc_gemm_kernel` too and would enter the DDIM loop's per-launch averages)
# HBM bytes per launch first: bench.py reports them as roofline.traffic when the profile is of the library it runs (digest-stamped)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline --no-parity --no-gpu-reference --no-vae > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline --no-parity --no-gpu-reference --no-vae > /dev/null 2>&1
python $R/tools/hbm_traffic.py $O/pmc_fetch $O/pmc_write $O/hbm_traffic.json > $O/hbm_traffic.txt
cp $O/hbm_traffic.json $R/profiles/${ROUND}_hbm_traffic.json
# shader clock / power while the loop runs, bf16 vs f16 (round-4 review: the f16 mode's 3 % was "unexplained by measurement")
sample_clocks() { while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket Graphics Package Power" | tr '\n' ' '; echo; sleep 1; done; }
# the driver's own command (all legs: roofline, parity, gpu_reference, cpu_baseline on the whole clip) + the VAE decode outside the metric
sample_clocks > $O/clocks_bf16.txt & SC=$!
python $R/bench.py --steps 5 --warmup 2 --vae > $O/bench_final.json 2> $O/bench_final.err
kill $SC
sample_clocks > $O/clocks_f16.txt & SC=$!
python $R/bench.py --steps 5 --warmup 2 --dtype f16 --no-cpu-baseline > $O/bench_f16.json 2> $O/bench_f16.err
kill $SC
python $R/bench.py --steps 1 --warmup 1 --dtype f32 --no-cpu-baseline --no-gpu-reference > $O/bench_f32.json 2> $O/bench_f32.err
rocprofv3 --kernel-trace --stats -d $O/ktrace -o kt -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-parity --no-gpu-reference --no-vae > /dev/null 2>&1
python $R/tools/rocprof_summary.py $(find $O/ktrace -name "*.db" | head -1) $O/kernel_stats.txt
# the other BASELINE.json configurations: labels follow the arguments, every line carries its roofline
python $R/bench.py --steps 1 --warmup 1 --frames 32 --size 768 --ddim-steps 50 --no-cpu-baseline --no-gpu-reference > $O/bench_cfg4.json 2>/dev/null
python $R/bench.py --steps 2 --warmup 1 --ip-tokens 16 --no-cpu-baseline > $O/bench_cfg5.json 2>/dev/null
# raw traces / counter CSVs stay on the box (gpurun merges at most 64 MiB back): the summaries above are what profiles/ keeps
rm -rf $O/ktrace $O/pmc_fetch $O/pmc_write 2>/dev/null
ls -la $O
