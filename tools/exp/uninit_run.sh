set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_uninit_gpu.py -q -m gpu 2>&1 | grep -v "^$" | tail -80
