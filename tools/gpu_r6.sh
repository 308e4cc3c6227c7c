#!/bin/bash
# Round-6 GPU calls, one parametrised script:  gpurun -- 'bash tools/gpu_r6.sh <section> [<section> ...]'
# Everything is written under gpurun_out/r6/; the summaries worth keeping are copied to profiles/ by hand afterwards.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r6
mkdir -p $OUT
BASE=${BASE_LIB:-tools/exp/libfyc_base.so}
benchline() {   # one line per bench JSON: value, gemm roofline, the first kernel families
  python - "$@" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        fam={k:v["ms_per_ddim_step"] for k,v in list(d.get("kernel_families",{}).items())[:9]}
        print(f, d["value"], "frames/s  ms/ddim-step", round(d["ms_per_step"]/25,2), " gemm", d.get("roofline",{}).get("achieved"), fam)
    except Exception as e: print(f, "FAILED", e)
PY
}
for sec in "$@"; do
  echo "=== $sec ($(date +%T)) ==="
  case $sec in
    tests)        # the whole -m gpu suite with durations
      timeout 1500 python -m pytest tests -x -q -m gpu --durations=30 2>&1 | tail -60 | tee $OUT/gpu_tests_durations.txt
      cp gpurun_out/parity_report.txt $OUT/parity_report.txt 2>/dev/null ;;
    tests_new)    # only the tests this round added / moved
      timeout 900 python -m pytest tests -x -q -m gpu -k "split_k or non_finite or clipped_model_output or reference_gpu" --durations=10 2>&1 | tail -30 | tee $OUT/tests_new.txt ;;
    bench)        # the driver's command
      timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -3 $OUT/bench_default.err; benchline $OUT/bench_default.json ;;
    bench_quick)
      timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gpu-reference --no-parity --no-vae > $OUT/bench_quick.json 2> $OUT/bench_quick.err; benchline $OUT/bench_quick.json ;;
    bench_ab)     # whole loop: base library vs the in-tree one, alternating
      for i in 1 2; do
        FYC_BENCH_SHAPES=$OUT/shapes_base_$i.txt FYC_LIB_PATH=$BASE timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gpu-reference --no-parity --no-vae > $OUT/bench_ab_base_$i.json 2> $OUT/bench_ab_base_$i.err
        FYC_BENCH_SHAPES=$OUT/shapes_new_$i.txt timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gpu-reference --no-parity --no-vae > $OUT/bench_ab_new_$i.json 2> $OUT/bench_ab_new_$i.err
      done
      benchline $OUT/bench_ab_*.json | tee $OUT/bench_ab.txt
      python - $OUT <<'PY' | tee $OUT/shapes_ab.txt
import sys, re, collections
out = sys.argv[1]
def load(tag):
    d = collections.defaultdict(list)
    for i in (1, 2):
        try:
            for l in open(f"{out}/shapes_{tag}_{i}.txt"):
                f = l.split()
                d[" ".join(f[6:])].append((float(f[0]), int(f[2].replace("n=", "")) if f[2].startswith("n=") and len(f[2]) > 2 else int(f[3])))
        except OSError:
            pass
    return d
b, n = load("base"), load("new")
rows = []
for k in set(b) | set(n):
    mb = sum(v[0] for v in b.get(k, [])) / max(len(b.get(k, [])), 1)
    mn = sum(v[0] for v in n.get(k, [])) / max(len(n.get(k, [])), 1)
    rows.append((mn - mb, mb, mn, k))
print("per DDIM step, in the pipeline: base ms -> new ms (delta), sorted by delta")
for d, mb, mn, k in sorted(rows):
    if abs(d) >= 0.01:
        print(f"{mb:8.3f} -> {mn:8.3f}  ({d:+.3f})  {k}")
print(f"sum of deltas {sum(r[0] for r in rows):+.3f} ms")
PY
      ;;
    probe)        # cold-operand per-shape times of the engine's GEMM calls: PROBE_* environment as tools/gemm_probe.py documents
      timeout 900 python tools/gemm_probe.py 2>&1 | tee $OUT/probe_${PROBE_TAG:-default}.txt | cut -c1-150 ;;
    gemm_parity)  # what changed in the GEMM family this round (tile config 11, split-K statistics)
      timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_f16_gpu.py -x -q -k "test_gemm_plain or test_gemm_conv or output_statistics or heads or split_k or geglu" 2>&1 | tail -6 | tee $OUT/gemm_parity.txt ;;
    gemm_ab)      # cold-operand per-shape sweep over the tile configs incl. 128x160 x 2 workgroups (11); small-M split-K policies; A-gather ablation
      PROBE_SWEEP=1 PROBE_CFGS=0,5,6,11,1 timeout 900 python tools/gemm_probe.py > $OUT/probe_tiles.txt 2>&1; cut -c1-120 $OUT/probe_tiles.txt
      PROBE_SWEEP=1 PROBE_SMALL=1 PROBE_CFGS=0,2,1,6,11 timeout 600 python tools/gemm_probe.py > $OUT/probe_small_tiles.txt 2>&1; cut -c1-120 $OUT/probe_small_tiles.txt
      for v in 4 5 7 10; do
        PROBE_TUNING=10=$v PROBE_SWEEP=1 PROBE_SMALL=1 PROBE_CFGS=0 timeout 600 python tools/gemm_probe.py > $OUT/probe_small_splitk$v.txt 2>&1; echo "-- split-K min K tiles $v"; cut -c1-120 $OUT/probe_small_splitk$v.txt
      done
      echo "-- A gather of 1 tap in 9 only (timing build, upper bound of a halo tile)"
      FYC_LIB_PATH=tools/exp/libfyc_ablate_a1.so PROBE_SWEEP=1 PROBE_CFGS=0,5,6 timeout 600 python tools/gemm_probe.py 2>&1 | grep -i "conv\|case" | tee $OUT/probe_conv_ablate_a1.txt | cut -c1-120 ;;
    phase_ab)     # every other block of an XCD starts v x 1024 cycles late: per-shape sweep
      for v in 0 8 16 24 32 48 64; do
        echo "-- phase delay $v x 1024 cycles"
        PROBE_TUNING=11=$v PROBE_SWEEP=1 PROBE_CFGS=${PHASE_CFGS:-0,5,6,11} timeout 600 python tools/gemm_probe.py > $OUT/probe_phase$v.txt 2>&1; grep -v amdgpu.ids $OUT/probe_phase$v.txt | cut -c1-110
      done ;;
    epi_ablation) # where does the packed LINEAR epilogue's time go?  timing builds -DFYC_ABL_EPI=1 (no global stores) / 2 (pass 1 only) / 3 (no epilogue)
      for v in 0 1 2 3; do
        echo "-- FYC_ABL_EPI=$v"
        lib=tools/exp/libfyc_abl_epi$v.so; [ $v = 0 ] && lib=followyourclick_amd/libfyc_hip.so
        FYC_LIB_PATH=$lib PROBE_SWEEP=1 PROBE_CFGS=0,5,6 timeout 600 python tools/gemm_probe.py 2>&1 | grep -v "amdgpu.ids\|GEGLU\|heads" | tee $OUT/probe_abl_epi$v.txt | cut -c1-100
      done ;;
    pre_ab)       # epilogue inputs pre-staged by DMA (default) vs loaded in the epilogue (tuning key 12 = 1)
      for v in 0 1 0 1; do
        echo "-- key 12 = $v"
        PROBE_TUNING=12=$v PROBE_SWEEP=1 PROBE_CFGS=0,5,6 timeout 600 python tools/gemm_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_pre_key12_$v.txt | cut -c1-100
      done ;;
    fast1_ab)     # specialised pass 1 of the packed LINEAR epilogue (default) vs the generic one (tuning key 13 = 1)
      for v in 0 1 0 1; do
        echo "-- key 13 = $v"
        PROBE_TUNING=13=$v PROBE_SWEEP=1 PROBE_CFGS=0,5,6 timeout 600 python tools/gemm_probe.py 2>&1 | grep -v "amdgpu.ids\|GEGLU\|heads" | tee $OUT/probe_fast1_key13_$v.txt | cut -c1-100
      done ;;
    epi_trace)    # s_memtime stamps inside the packed LINEAR epilogue (timing build tools/exp/libfyc_trace.so)
      FYC_LIB_PATH=tools/exp/libfyc_trace.so timeout 600 python tools/gemm_epilogue_trace.py 2>&1 | grep -v amdgpu.ids | tee $OUT/epilogue_trace.txt
      echo "-- generic pass 1 (key 13 = 1), inputs loaded in the epilogue (key 12 = 1)"
      FYC_LIB_PATH=tools/exp/libfyc_trace.so PROBE_TUNING=13=1,12=1 timeout 600 python tools/gemm_epilogue_trace.py 2>&1 | grep -v amdgpu.ids | tee $OUT/epilogue_trace_old.txt ;;
    epi2_ab)      # GEGLU / head-split epilogues: packed GEGLU + fast pass 1 (default) vs key 13 = 1 (round-5 code), and no epilogue at all (timing build)
      for v in 0 1 0 1; do
        echo "-- key 13 = $v"
        PROBE_TUNING=13=$v PROBE_SWEEP=1 PROBE_CFGS=0,5,6 timeout 600 python tools/gemm_probe.py 2>&1 | grep "GEGLU\|heads\|tQKV\|case" | tee $OUT/probe_epi2_key13_$v.txt | cut -c1-100
      done
      echo "-- FYC_ABL_EPI=3 (no epilogue)"
      FYC_LIB_PATH=tools/exp/libfyc_abl_epi3.so PROBE_SWEEP=1 PROBE_CFGS=0,5,6 timeout 600 python tools/gemm_probe.py 2>&1 | grep "GEGLU\|heads\|tQKV\|case" | tee $OUT/probe_epi2_abl3.txt | cut -c1-100 ;;
    mi32)         # 32x32x16 matrix instruction in the K loop (tile configs 12 / 13 / 14 = twins of 5 / 7 / 3): parity, cold-operand probe, whole loop
      timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_f16_gpu.py -x -q -k "test_gemm_plain or test_gemm_conv or output_statistics or heads or split_k or geglu" 2>&1 | tail -6 | tee $OUT/mi32_parity.txt
      PROBE_SWEEP=1 PROBE_CFGS=${MI32_CFGS:-0,5,12,7,13,3,14} timeout 900 python tools/gemm_probe.py > $OUT/probe_mi32.txt 2>&1; grep -v amdgpu.ids $OUT/probe_mi32.txt | cut -c1-130 ;;
    mi32_probe)   # the probe alone
      PROBE_SWEEP=1 PROBE_CFGS=${MI32_CFGS:-0,5,12} timeout 900 python tools/gemm_probe.py > $OUT/probe_mi32_v2.txt 2>&1; grep -v amdgpu.ids $OUT/probe_mi32_v2.txt | cut -c1-130 ;;
    libs_probe)   # cold-operand probe of several libraries: LIBS="a.so b.so" (first = reference), PROBE_CFGS as usual
      for lib in ${LIBS}; do
        echo "-- $lib"
        FYC_LIB_PATH=$lib PROBE_SWEEP=1 PROBE_CFGS=${PROBE_CFGS:-0,5,6,7} timeout 600 python tools/gemm_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_$(basename $lib .so).txt | cut -c1-110
      done ;;
    libs_bench)   # whole loop, per-shape, alternating over LIBS (2 passes)
      for i in 1 2; do for lib in ${LIBS}; do
        tag=$(basename $lib .so)_$i
        FYC_LIB_PATH=$lib FYC_BENCH_SHAPES=$OUT/shapes_lib_$tag.txt timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gpu-reference --no-parity --no-vae > $OUT/bench_lib_$tag.json 2> $OUT/bench_lib_$tag.err
      done; done
      benchline $OUT/bench_lib_*.json | tee $OUT/bench_libs.txt ;;
    keys_ab)      # in-pipeline per-shape times of the in-tree library under tuning keys (KEYS="12=1 13=1 12=1,13=1")
      for k in "" ${KEYS:-12=1 13=1}; do
        tag=$(echo "x$k" | tr '=,' '__')
        FYC_TUNING=$k FYC_BENCH_SHAPES=$OUT/shapes_keys_$tag.txt timeout 600 python bench.py --steps ${KEYS_STEPS:-2} --warmup 1 --no-cpu-baseline --no-gpu-reference --no-parity --no-vae > $OUT/bench_keys_$tag.json 2> $OUT/bench_keys_$tag.err
      done
      benchline $OUT/bench_keys_*.json
      python - $OUT <<'PY' | tee $OUT/shapes_keys.txt
import sys, glob, os
out = sys.argv[1]
tabs = {}
for f in sorted(glob.glob(f"{out}/shapes_keys_*.txt")):
    tag = os.path.basename(f)[len("shapes_keys_"):-4]
    tabs[tag] = {" ".join(l.split()[6:]): float(l.split()[0]) for l in open(f)}
tags = sorted(tabs)
keys = sorted(tabs[tags[0]], key=lambda k: -tabs[tags[0]][k])
print("ms per DDIM step in the pipeline |", " | ".join(tags))
for k in keys[:45]:
    print(" ".join(f"{tabs[t].get(k, 0):8.3f}" for t in tags), " ", k)
PY
      ;;
    *) echo "unknown section $sec" ;;
  esac
done
