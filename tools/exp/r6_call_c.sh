# the counter and trace passes of tools/collect_profiles.sh alone (round 6: re-taken with --no-vae)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline --no-parity --no-gpu-reference --no-vae > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline --no-parity --no-gpu-reference --no-vae > /dev/null 2>&1
python $R/tools/hbm_traffic.py $O/pmc_fetch $O/pmc_write $O/hbm_traffic.json > $O/hbm_traffic.txt
cat $O/hbm_traffic.txt
rocprofv3 --kernel-trace --stats -d $O/ktrace -o kt -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-parity --no-gpu-reference --no-vae > /dev/null 2>&1
python $R/tools/rocprof_summary.py $(find $O/ktrace -name "*.db" | head -1) $O/kernel_stats.txt
head -8 $O/kernel_stats.txt | cut -c1-160
rm -rf $O/ktrace $O/pmc_fetch $O/pmc_write 2>/dev/null
