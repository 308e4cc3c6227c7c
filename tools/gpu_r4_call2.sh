#!/bin/bash
# round 4, GPU call 2: packed LINEAR epilogue + residual in accumulators + fenced fragment reads + L2 prefetch (tuning key 10);
# new full-shape parity tests, reference script on the HIP kernels, RCCL single-GPU tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > $O/c2_gemm_tests.txt 2>&1; tail -4 $O/c2_gemm_tests.txt
FYC_TUNING=10=2 timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > $O/c2_gemm_tests_pf.txt 2>&1; tail -2 $O/c2_gemm_tests_pf.txt
timeout 900 python -m pytest tests/test_fullwidth_gpu.py -x -q -m gpu -k "full_shape or cfg5 or cfg1" -s > $O/c2_fullshape.txt 2>&1; grep -i "FULL shape\|cfg5 IP\|passed\|failed\|error" $O/c2_fullshape.txt | tail -24
timeout 600 python -m pytest tests/test_script_dropin.py tests/test_distributed_gpu.py tests/test_abi.py -q -m gpu > $O/c2_script_dist.txt 2>&1; tail -5 $O/c2_script_dist.txt
for pf in 0 2 3; do
  FYC_TUNING=10=$pf PROBE_SWEEP=1 PROBE_CFGS=0,5,6,1 timeout 400 python tools/gemm_probe.py > $O/c2_probe_pf$pf.txt 2>&1
done
paste <(tail -22 $O/c2_probe_pf0.txt) <(tail -22 $O/c2_probe_pf2.txt | cut -c48-) <(tail -22 $O/c2_probe_pf3.txt | cut -c48-)
for pf in 0 2; do
  FYC_TUNING=10=$pf PROBE_CFGS=5,6 FYC_LIB_PATH=tools/exp/libfyc_trace.so timeout 300 python tools/gemm_phase_probe.py > $O/c2_phase_pf$pf.txt 2>&1
  tail -17 $O/c2_phase_pf$pf.txt | cut -c1-150
done
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/c2_bench_pf0.json 2> $O/c2_bench_pf0.err
FYC_TUNING=10=2 FYC_BENCH_SHAPES=$O/c2_shapes_pf2.txt timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/c2_bench_pf2.json 2> $O/c2_bench_pf2.err
python - <<'PY'
import json
for f in ("c2_bench_pf0", "c2_bench_pf2"):
    try:
        d = json.load(open(f"gpurun_out/r4/{f}.json"))
        print(f, d["value"], d["gpu_ms_per_ddim_step"], {k: (v["ms_per_ddim_step"], v.get("tflops")) for k, v in d["kernel_families"].items() if k in ("gemm", "conv3x3")})
    except Exception as e:
        print(f, "failed", e)
PY
