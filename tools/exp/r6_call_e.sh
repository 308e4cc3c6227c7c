set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
echo "=== parity (GEMM family) on the 16-byte residual loads"
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_f16_gpu.py -x -q -k "test_gemm_plain or test_gemm_conv or output_statistics or split_k or conv_output" 2>&1 | tail -3
echo "=== bitwise: residual-carrying GEMMs of both libraries give the same bits"
python - <<'PY'
import os, subprocess, sys, torch
code = r'''
import sys, torch
sys.path.insert(0, ".")
from followyourclick_amd import ops
h = ops.get(); h.ensure_init(torch.device("cuda:0"))
T = torch.bfloat16
g = torch.Generator().manual_seed(1)
outs = []
for (M, N, K, tile) in [(32768, 640, 640, 0), (131072, 320, 320, 0), (8192, 1280, 1280, 0), (4096, 320, 2880, 5), (1000, 960, 320, 5), (300, 328, 64, 1), (4096, 640, 640, 7)]:
    a = torch.randn(M, K, generator=g).to(T).cuda(); w = (torch.randn(N, K, generator=g) / K ** 0.5).to(T).cuda(); r = torch.randn(M, N, generator=g).to(T).cuda()
    o = torch.empty(M, N, dtype=T, device="cuda")
    h.gemm(a, w, o, M=M, N=N, K=K, lda=K, ldw=K, ldo=N, ldr=N, residual=r, bias=torch.randn(N, generator=g).cuda(), tile=tile)
    torch.cuda.synchronize()
    outs.append(o.cpu())
torch.save(outs, sys.argv[1])
'''
for tag, lib in (("new", "followyourclick_amd/libfyc_hip.so"), ("res8", "tools/exp/libfyc_res8.so")):
    subprocess.run([sys.executable, "-c", code, f"/tmp/res_{tag}.pt"], env=dict(os.environ, FYC_LIB_PATH=lib), check=True)
a, b = torch.load("/tmp/res_new.pt"), torch.load("/tmp/res_res8.pt")
print("bitwise equal per case:", [bool(torch.equal(x.view(torch.int16), y.view(torch.int16))) for x, y in zip(a, b)])
PY
