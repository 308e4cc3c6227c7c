#!/bin/bash
# round 4, GPU call 4: ordered (atomic-free) statistics in fyc_gemm / fyc_gn_stats, bitwise determinism of the sampling loop,
# ff_block after the M0 save/restore, bench labels / cpu baseline / VAE roofline
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or conv or groupnorm or statistics or ff_block or panel or temporal_block or layernorm" > $O/c4_kernel_tests.txt 2>&1; tail -4 $O/c4_kernel_tests.txt
timeout 600 python -m pytest tests/test_engine_gpu.py -x -q -m gpu > $O/c4_engine_tests.txt 2>&1; tail -4 $O/c4_engine_tests.txt
timeout 900 python -m pytest tests/test_fullwidth_gpu.py -x -q -m gpu -k "bitwise or cfg1 or full_width_forward" -s > $O/c4_fullwidth.txt 2>&1; grep -i "passed\|failed\|error\|bitwise" $O/c4_fullwidth.txt | tail -6
PROBE_CFGS=5,6 FYC_LIB_PATH=tools/exp/libfyc_trace.so timeout 300 python tools/gemm_phase_probe.py > $O/c4_phase.txt 2>&1; tail -18 $O/c4_phase.txt | cut -c1-150
timeout 300 python tools/cpu_baseline_sweep.py 1 8 16 32 64 > $O/c4_cpu_sweep.txt 2>&1; cat $O/c4_cpu_sweep.txt | tail -6
FYC_BENCH_SHAPES=$O/c4_shapes.txt timeout 600 python bench.py --vae > $O/c4_bench.json 2> $O/c4_bench.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r4/c4_bench.json"))
    print(d["value"], d["gpu_ms_per_ddim_step"], d["metric"], "|", d["config"]["workload"][:40])
    print({k: (v["ms_per_ddim_step"], v.get("tflops")) for k, v in d["kernel_families"].items()})
    print(d["roofline"]["achieved"], d["roofline"]["frac"], d.get("roofline_vae"), d.get("vae_decode_ms"))
    print(d.get("cpu_baseline"))
except Exception as e:
    print("bench failed", e, open("gpurun_out/r4/c4_bench.err").read()[-1500:])
PY
