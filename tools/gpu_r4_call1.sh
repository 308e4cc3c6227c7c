#!/bin/bash
# round 4, GPU call 1: ping-pong GEMM loop - correctness, per-shape A/B against the one-phase loop, phase timing, bench A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "pingpong" > $O/c1_pp_tests.txt 2>&1
tail -5 $O/c1_pp_tests.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "test_gemm_plain or test_gemm_conv or test_gemm_output_statistics or geglu_every" > $O/c1_gemm_tests.txt 2>&1
tail -3 $O/c1_gemm_tests.txt
PROBE_SWEEP=1 PROBE_CFGS=0,5,21,6,22,7,23 timeout 600 python tools/gemm_probe.py > $O/c1_probe_pp.txt 2>&1
FYC_TUNING=8=1 PROBE_SWEEP=1 PROBE_CFGS=21,22 timeout 300 python tools/gemm_probe.py > $O/c1_probe_pp_noprio.txt 2>&1
FYC_LIB_PATH=tools/exp/libfyc_trace.so timeout 300 python tools/gemm_phase_probe.py > $O/c1_phase.txt 2>&1
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/c1_bench_v1.json 2> $O/c1_bench_v1.err
FYC_TUNING=9=2 FYC_BENCH_SHAPES=$O/c1_shapes_pp.txt timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/c1_bench_pp.json 2> $O/c1_bench_pp.err
cat $O/c1_probe_pp.txt | tail -25
cat $O/c1_phase.txt | tail -34
python - <<'PY'
import json
for f in ("c1_bench_v1", "c1_bench_pp"):
    try:
        d = json.load(open(f"gpurun_out/r4/{f}.json"))
        print(f, d["value"], d["gpu_ms_per_ddim_step"], {k: (v["ms_per_ddim_step"], v.get("tflops")) for k, v in d["kernel_families"].items() if k in ("gemm", "conv3x3")})
    except Exception as e:
        print(f, "failed", e)
PY
