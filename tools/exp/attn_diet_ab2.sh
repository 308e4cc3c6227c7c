mkdir -p gpurun_out/r5
{
echo "== parity (in-tree: ABC + steady-state loop)"; timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_f16_gpu.py -x -q -k "attention" 2>&1 | tail -2
for i in 1 2 3; do
  echo "-- ABC $i"; FYC_LIB_PATH=tools/exp/libfyc_attn_c.so timeout 200 python tools/attn_bench.py 2>&1 | grep "B="
  echo "-- ABC + steady-state loop $i"; timeout 200 python tools/attn_bench.py 2>&1 | grep "B="
done
} 2>&1 | tee gpurun_out/r5/attn_diet_ab2.txt
