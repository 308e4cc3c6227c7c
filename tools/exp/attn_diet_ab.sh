mkdir -p gpurun_out/r5
echo "== parity AB (in-tree)"; timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_f16_gpu.py -x -q -k "attention" 2>&1 | tail -2
echo "== parity ABC"; FYC_LIB_PATH=tools/exp/libfyc_attn_c.so timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_f16_gpu.py -x -q -k "attention" 2>&1 | tail -2
for i in 1 2; do
  echo "-- base $i"; FYC_LIB_PATH=tools/exp/libfyc_base.so timeout 200 python tools/attn_bench.py 2>&1 | grep "B="
  echo "-- AB (no setprio, no-honor-nans) $i"; timeout 200 python tools/attn_bench.py 2>&1 | grep "B="
  echo "-- ABC (+ one-statement masked DMA issue) $i"; FYC_LIB_PATH=tools/exp/libfyc_attn_c.so timeout 200 python tools/attn_bench.py 2>&1 | grep "B="
done
