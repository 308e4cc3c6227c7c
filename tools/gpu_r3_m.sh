#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_fullwidth_gpu.py tests/test_engine_gpu.py -q -x 2>&1 | tail -4 > $O/m_parity.txt; cat $O/m_parity.txt
timeout 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/m_bench.json 2> $O/m_bench.err; python -c "
import json; d=json.load(open('$O/m_bench.json')); print(d['value'], d['ms_per_step'], d['gpu_ms_per_ddim_step'], d['roofline']['achieved']); print({k:(v['ms_per_ddim_step'], v.get('tflops')) for k,v in d['kernel_families'].items()})"
