#!/bin/bash
# round 4, GPU call 3: overlapped-epilogue kernel (tile config 31), fixed staging slice / residual loads of the packed epilogue
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > $O/c3_gemm_tests.txt 2>&1; tail -4 $O/c3_gemm_tests.txt
timeout 900 python -m pytest tests/test_fullwidth_gpu.py -x -q -m gpu -k "cfg5_ip or cfg1" -s > $O/c3_traj.txt 2>&1; grep -i "passed\|failed\|error" $O/c3_traj.txt | tail -4
PROBE_SWEEP=1 PROBE_CFGS=0,5,6,31 timeout 400 python tools/gemm_probe.py > $O/c3_probe.txt 2>&1; tail -22 $O/c3_probe.txt
PROBE_CFGS=5,6,31 FYC_LIB_PATH=tools/exp/libfyc_trace.so timeout 300 python tools/gemm_phase_probe.py > $O/c3_phase.txt 2>&1; tail -24 $O/c3_phase.txt | cut -c1-150
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/c3_bench_v1.json 2> $O/c3_bench_v1.err
FYC_TUNING=9=3 FYC_BENCH_SHAPES=$O/c3_shapes_ov.txt timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/c3_bench_ov.json 2> $O/c3_bench_ov.err
python - <<'PY'
import json
for f in ("c3_bench_v1", "c3_bench_ov"):
    try:
        d = json.load(open(f"gpurun_out/r4/{f}.json"))
        print(f, d["value"], d["gpu_ms_per_ddim_step"], {k: (v["ms_per_ddim_step"], v.get("tflops")) for k, v in d["kernel_families"].items() if k in ("gemm", "conv3x3")})
    except Exception as e:
        print(f, "failed", e, open(f"gpurun_out/r4/{f}.err").read()[-600:])
PY
