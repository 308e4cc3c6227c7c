#!/bin/bash
# Round-5 GPU calls, one parametrised script:  gpurun -- 'bash tools/gpu_r5.sh <section> [<section> ...]'
# Everything is written under gpurun_out/r5/; the summaries worth keeping are copied to profiles/ by hand afterwards.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r5
mkdir -p $OUT
IL=${IL_LIB:-tools/exp/libfyc_il60.so}
for sec in "$@"; do
  echo "=== $sec ($(date +%T)) ==="
  case $sec in
    il_parity)    # the interleaved-DMA main loop against the op specification
      FYC_LIB_PATH=$IL timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "test_gemm_plain or test_gemm_conv or test_gemm_output_statistics or geglu or heads" 2>&1 | tail -5 | tee $OUT/il_parity.txt ;;
    il_probe)     # cold-operand per-shape A/B: product library vs interleaved library at several start skews (tuning key 5)
      PROBE_SWEEP=1 PROBE_CFGS=0 timeout 600 python tools/gemm_probe.py > $OUT/probe_base.txt 2>&1
      for sk in 1 3 5 8; do
        FYC_LIB_PATH=$IL PROBE_TUNING=5=$sk PROBE_SWEEP=1 PROBE_CFGS=0 timeout 600 python tools/gemm_probe.py > $OUT/probe_il_skew$sk.txt 2>&1
      done
      tail -n +1 $OUT/probe_base.txt $OUT/probe_il_skew*.txt | cut -c1-110 ;;
    il_bench)     # whole-loop A/B inside one call
      for i in 1 2; do
        timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gpu-reference --no-parity > $OUT/bench_base_$i.json 2> $OUT/bench_base_$i.err
        FYC_LIB_PATH=$IL timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gpu-reference --no-parity > $OUT/bench_il_$i.json 2> $OUT/bench_il_$i.err
      done
      for f in $OUT/bench_base_*.json $OUT/bench_il_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1], d["value"], "frames/s  gemm", d["roofline"]["achieved"], {k:v["ms_per_ddim_step"] for k,v in list(d["kernel_families"].items())[:6]})
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
      done ;;
    attn_pmc)     # the counters behind DESIGN's attention paragraph: SQ busy / VALU / MFMA / waits / LDS per head dim
      for shape in "32 8 4096 4096 40" "32 8 1024 1024 80" "32 8 256 256 160" "32 8 4096 77 40"; do
        tag=$(echo $shape | tr ' ' '_')
        for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
                    "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE"; do
          d=$OUT/attn_pmc/$tag/$(echo $pass | cut -c1-20 | tr ' ' '_')
          (cd /tmp && timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OLDPWD/$d -o p -- python $OLDPWD/tools/one_attn.py $shape 3 > /dev/null 2>&1)
        done
        echo "## attention B H nq nk d = $shape" >> $OUT/attention_pmc.txt
        python tools/pmc_summary.py $OUT/attn_pmc/$tag fyc_attn >> $OUT/attention_pmc.txt 2>&1
      done
      rm -rf $OUT/attn_pmc
      cat $OUT/attention_pmc.txt | head -120 ;;
    attn_ab)      # attention kernel: parity tests on the new library, then per-shape A/B against tools/exp/libfyc_base.so (alternating)
      timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_f16_gpu.py -x -q -k "attention" 2>&1 | tail -4 | tee $OUT/attn_parity.txt
      for i in 1 2; do
        echo "-- base $i"; FYC_LIB_PATH=tools/exp/libfyc_base.so timeout 300 python tools/attn_bench.py 2>&1 | grep "B="
        echo "-- new $i"; timeout 300 python tools/attn_bench.py 2>&1 | grep "B="
      done | tee $OUT/attn_ab.txt ;;
    bench_ab)     # whole loop: base library vs the in-tree one, alternating
      for i in 1 2; do
        FYC_LIB_PATH=tools/exp/libfyc_base.so timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gpu-reference --no-parity > $OUT/bench_ab_base_$i.json 2> $OUT/bench_ab_base_$i.err
        timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gpu-reference --no-parity > $OUT/bench_ab_new_$i.json 2> $OUT/bench_ab_new_$i.err
      done
      for f in $OUT/bench_ab_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1], d["value"], "frames/s  gemm", d["roofline"]["achieved"], {k:v["ms_per_ddim_step"] for k,v in list(d["kernel_families"].items())[:8]})
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
      done ;;
    ref_gpu)      # the unmodified reference on this chip: parity yardstick + its own frames/s
      timeout 900 python -m pytest tests/test_reference_gpu.py -q 2>&1 | tail -8 | tee $OUT/ref_gpu_tests.txt
      for at in sdpa eager; do
        timeout 600 python -m oracle.gpu_reference --json --attention $at --dtype bf16 2>&1 | grep GPU_REFERENCE | tee -a $OUT/gpu_reference.txt
      done
      timeout 600 python -m oracle.gpu_reference --json --attention sdpa --dtype f16 2>&1 | grep GPU_REFERENCE | tee -a $OUT/gpu_reference.txt
      cp gpurun_out/parity_report.txt $OUT/parity_report_ref.txt 2>/dev/null ;;
    bench)        # the driver's own command
      timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 3000 $OUT/bench_default.json ;;
    tests)        # the driver's GPU tier, with the slowest items listed
      timeout 1500 python -m pytest tests -m gpu -x -q --durations=40 2>&1 | tail -60 | tee $OUT/gpu_tests.txt ;;
    *) echo "unknown section $sec" ;;
  esac
done
