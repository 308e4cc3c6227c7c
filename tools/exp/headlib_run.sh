set -u
export TMPDIR=/tmp
echo "=== HEAD lib, failing test alone"
FYC_LIB_PATH=tools/exp/libfyc_head.so timeout 400 python -m pytest tests/test_fullwidth_gpu.py -x -q -m gpu -k "test_full_width_forward_vs_reference" 2>&1 | grep -E "full-width fwd|passed|failed|Error" | tail -5
echo "=== HEAD lib, uninit test"
FYC_LIB_PATH=tools/exp/libfyc_head.so timeout 600 python -m pytest tests/test_uninit_gpu.py -q -m gpu 2>&1 | grep -v "^$" | tail -40
