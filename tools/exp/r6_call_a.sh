set -u
export TMPDIR=/tmp
echo "=== gemm parity on the in-tree library (FRAG_ROW + XSTEP + heads_pk gating)"
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_f16_gpu.py tests/test_uninit_gpu.py -x -q 2>&1 | tail -4
echo "=== reference subprocess alone: where does its time go"
mkdir -p /tmp/refdump && (time MIOPEN_FIND_MODE=FAST OMP_NUM_THREADS=16 python -m oracle.gpu_reference --dump small,cfg4ip,cfg3 --out-dir /tmp/refdump) 2>&1 | grep -v amdgpu.ids | tail -15
