#!/bin/bash
# round 4, call 10: everything NEW since the last full GPU suite first (FYC_F16 kernels / engine / script tests, the animate.py and
# inference_w_camera_lora.py script tests), then the measurement set of the final library (PMC traffic -> bench line; f16 bench line
# beside it; kernel trace), then the engine-level bf16 / f32 tests as a sample of the unchanged paths.  Each part writes its own file.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $O
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 ))s] $*"; }
timeout 420 python -m pytest tests/test_kernels_f16_gpu.py tests/test_kernels_gpu.py tests/test_fullwidth_gpu.py tests/test_script_dropin.py -q -m gpu \
    -k "(f16 and not bf16) or animate or camera or vae_decode_full_width" -p no:cacheprovider > $O/new_tests.txt 2>&1
stamp "new tests: $(tail -1 $O/new_tests.txt)"
grep -E "^(FAILED|ERROR)" $O/new_tests.txt | head -40
cp gpurun_out/parity_report.txt $O/parity_report_f16.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python $R/tools/hbm_traffic.py $O/pmc_fetch $O/pmc_write $O/hbm_traffic.json > $O/hbm_traffic.txt 2>&1
cp $O/hbm_traffic.json $R/profiles/r04_hbm_traffic.json
stamp "traffic: $(tail -2 $O/hbm_traffic.txt | tr '\n' ' ')"
python $R/bench.py --steps 5 --warmup 2 --vae > $O/bench_final.json 2> $O/bench_final.err
stamp "bench bf16: $(python -c "import json;d=json.load(open('$O/bench_final.json'));print(d['value'],d['ms_per_step'],d['roofline']['achieved'],d['roofline']['frac'],d['roofline']['traffic'])" 2>&1)"
python $R/bench.py --steps 5 --warmup 2 --dtype f16 --vae --no-cpu-baseline > $O/bench_f16.json 2> $O/bench_f16.err
stamp "bench f16: $(python -c "import json;d=json.load(open('$O/bench_f16.json'));print(d['value'],d['ms_per_step'],d['roofline']['achieved'],d['roofline']['frac'])" 2>&1)"
python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_bf16_again.json 2>/dev/null
stamp "bench bf16 again: $(python -c "import json;d=json.load(open('$O/bench_bf16_again.json'));print(d['value'])" 2>&1)"
rocprofv3 --kernel-trace --stats -d $O/ktrace -o kt -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python $R/tools/rocprof_summary.py $(find $O/ktrace -name "*.db" | head -1) $O/kernel_stats.txt > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/ktrace16 -o kt -- python $R/bench.py --steps 1 --warmup 0 --dtype f16 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python $R/tools/rocprof_summary.py $(find $O/ktrace16 -name "*.db" | head -1) $O/kernel_stats_f16.txt > /dev/null 2>&1
rm -rf $O/ktrace $O/ktrace16 $O/pmc_fetch $O/pmc_write 2>/dev/null
stamp "traces done"
cd $R
timeout ${SAMPLE_TIMEOUT:-240} python -m pytest tests/test_engine_gpu.py tests/test_dropin_gpu.py tests/test_distributed_gpu.py -q -m gpu -p no:cacheprovider > $O/sample_tests.txt 2>&1
stamp "sample of the unchanged paths: $(tail -1 $O/sample_tests.txt)"
grep -E "^(FAILED|ERROR)" $O/sample_tests.txt | head -20
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
stamp done
