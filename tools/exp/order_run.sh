set -u
export TMPDIR=/tmp
echo "=== fullsize then fullwidth"
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_fullwidth_gpu.py -q -m gpu -k "parity_mode or clips_are or full_width_forward_vs_reference" 2>&1 | grep -E "full-width fwd|passed|failed|Error|zero page|assert" | tail -12
echo "=== engine + dropin then fullwidth"
timeout 900 python -m pytest tests/test_dropin_gpu.py tests/test_engine_gpu.py tests/test_fullwidth_gpu.py -q -m gpu -k "not (trajectory or ip_adapter_forward or cfg4 or vae or cfg5)" 2>&1 | grep -E "full-width fwd|passed|failed|Error|zero page|assert" | tail -12
