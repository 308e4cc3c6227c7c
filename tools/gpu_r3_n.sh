#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash $R/tools/collect_profiles.sh > /dev/null 2>&1
python -c "
import json; d=json.load(open('$R/gpurun_out/r03/bench_final.json')); print(d['value'], d['ms_per_step'], d['gpu_ms_per_ddim_step'], d['roofline']['achieved'], d['roofline']['traffic']); print({k:(v['ms_per_ddim_step'], v.get('tflops')) for k,v in d['kernel_families'].items()})"
cd $R; timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee $R/gpurun_out/n_gputests.txt
