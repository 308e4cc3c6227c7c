#!/bin/bash
# round 4, call 11 (after the swap32_max fix of the attention kernel's running maximum): every test that touches the attention /
# temporal kernels or the f16 mode, then the measurement set on the final library (PMC traffic -> bench line, f16 line, kernel traces)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04c
mkdir -p $O
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 ))s] $*"; }
timeout 200 python -m pytest tests/test_kernels_f16_gpu.py tests/test_kernels_gpu.py -q -m gpu -k "(f16 and not bf16) or attention or temporal or tattn" -p no:cacheprovider > $O/kernel_tests.txt 2>&1
stamp "kernel tests: $(tail -1 $O/kernel_tests.txt)"
grep -E "^(FAILED|ERROR)" $O/kernel_tests.txt | head -30
timeout 330 python -m pytest tests/test_fullwidth_gpu.py tests/test_engine_gpu.py -q -m gpu -p no:cacheprovider > $O/engine_tests.txt 2>&1
stamp "engine tests: $(tail -1 $O/engine_tests.txt)"
grep -E "^(FAILED|ERROR)" $O/engine_tests.txt | head -30
cp gpurun_out/parity_report.txt $O/parity_report.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python $R/tools/hbm_traffic.py $O/pmc_fetch $O/pmc_write $O/hbm_traffic.json > $O/hbm_traffic.txt 2>&1
cp $O/hbm_traffic.json $R/profiles/r04_hbm_traffic.json
python $R/bench.py --steps 5 --warmup 2 --vae > $O/bench_final.json 2> $O/bench_final.err
stamp "bench bf16: $(python -c "import json;d=json.load(open('$O/bench_final.json'));print(d['value'],d['ms_per_step'],d['roofline']['achieved'],d['roofline']['frac'],d['roofline']['traffic'],d.get('roofline_attention',{}).get('achieved'))" 2>&1)"
python $R/bench.py --steps 5 --warmup 2 --dtype f16 --vae --no-cpu-baseline > $O/bench_f16.json 2> $O/bench_f16.err
stamp "bench f16: $(python -c "import json;d=json.load(open('$O/bench_f16.json'));print(d['value'],d['ms_per_step'],d['roofline']['achieved'],d['roofline']['frac'])" 2>&1)"
rocprofv3 --kernel-trace --stats -d $O/ktrace -o kt -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python $R/tools/rocprof_summary.py $(find $O/ktrace -name "*.db" | head -1) $O/kernel_stats.txt > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/ktrace16 -o kt -- python $R/bench.py --steps 1 --warmup 0 --dtype f16 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python $R/tools/rocprof_summary.py $(find $O/ktrace16 -name "*.db" | head -1) $O/kernel_stats_f16.txt > /dev/null 2>&1
rm -rf $O/ktrace $O/ktrace16 $O/pmc_fetch $O/pmc_write 2>/dev/null
stamp "traces done"
cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout ${REST_TIMEOUT:-60} python -m pytest tests/test_dropin_gpu.py tests/test_script_dropin.py -q -m gpu -p no:cacheprovider > $O/dropin_tests.txt 2>&1
stamp "drop-in / script tests: $(tail -1 $O/dropin_tests.txt)"
stamp done
