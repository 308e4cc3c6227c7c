#!/bin/bash
# round 4, GPU call 6: tile choice at M < 4096, non-temporal epilogue stores / residual loads A/B (tools/exp builds)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -x -q -m gpu -k "gemm or conv or statistics or sampling or unet or vae" > $O/c6_tests.txt 2>&1; tail -3 $O/c6_tests.txt
for v in base nts ntsl; do
  if [ $v = base ]; then unset FYC_LIB_PATH; else export FYC_LIB_PATH=tools/exp/libfyc_$v.so; fi
  PROBE_SWEEP=1 PROBE_CFGS=0 timeout 300 python tools/gemm_probe.py > $O/c6_probe_$v.txt 2>&1
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/c6_bench_$v.json 2> $O/c6_bench_$v.err
done
unset FYC_LIB_PATH
paste <(cut -c1-56 $O/c6_probe_base.txt) <(cut -c49-56 $O/c6_probe_nts.txt) <(cut -c49-56 $O/c6_probe_ntsl.txt) | tail -24
python - <<'PY'
import json
for f in ("base", "nts", "ntsl"):
    try:
        d = json.load(open(f"gpurun_out/r4/c6_bench_{f}.json"))
        print(f, d["value"], d["gpu_ms_per_ddim_step"], {k: (v["ms_per_ddim_step"], v.get("tflops")) for k, v in d["kernel_families"].items() if k in ("gemm", "conv3x3", "gn_apply", "row_stats")})
    except Exception as e:
        print(f, "failed", e, open(f"gpurun_out/r4/c6_bench_{f}.err").read()[-600:])
PY
