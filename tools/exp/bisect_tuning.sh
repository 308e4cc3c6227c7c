set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
for k in "" "13=1" "12=1" "10=16" "13=1,12=1,10=16"; do
  echo "=== FYC_TUNING='$k'"
  FYC_TUNING=$k timeout 400 python -m pytest tests/test_fullwidth_gpu.py -x -q -m gpu -k "test_full_width_forward_vs_reference" 2>&1 | grep -E "full-width fwd|passed|failed|Error" | tail -5
done
echo "=== heads parity on the working-tree lib"
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_f16_gpu.py -x -q -k "heads" 2>&1 | tail -5
