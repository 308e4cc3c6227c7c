"""The RCCL side of the multi-GPU path on the ONE GPU a test box has: `bench.py` exactly as the driver launches it for N > 1
(`python -m torch.distributed.run ... bench.py --gpus N`), with N = 1.  Under a launcher `distributed.init_from_env` opens the
"nccl" (= RCCL) process group even for a single rank, so the weight broadcast (host- and device-resident tensors), the barrier and
the max-over-ranks all-reduce run through RCCL once before an 8-GPU node sees them."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_under_torch_distributed_run_on_one_gpu():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29781",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--frames", "2", "--size", "64", "--ddim-steps", "2",
           "--no-cpu-baseline", "--no-roofline", "--no-gpu-reference", "--no-parity", "--no-vae"]      # (the legs a default run adds are covered by the driver's own bench run)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["data"] == "synthetic"
    assert d["rccl_ranks"] == 1 and d["collective_backend"] == "nccl"       # the all-reduce of ones went through RCCL
    # the packed weights went through dist.broadcast on the nccl backend (bytes moved > 0 only with a live process group)
    assert "weight broadcast 0.00 GiB" not in d["config"]["parallelism"], d["config"]["parallelism"]


def test_broadcast_of_host_resident_module_over_rccl(tmp_path):
    """the INTEGRATION.md recipe wraps `unet.load_state_dict(...)` while the reference's UNet is still on the CPU
    (scripts/inference.py:178 vs :213).  nccl moves device memory only: the buckets are staged through the GPU."""
    code = f'''
import os, sys, torch
sys.path.insert(0, {ROOT!r})
from followyourclick_amd import distributed as D
r, w, l = D.init_from_env()
assert torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl"
m = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.BatchNorm1d(64))          # on the HOST
ref = {{k: v.clone() for k, v in m.state_dict().items()}}
moved = D.broadcast_module(m)
assert moved > 0 and all(torch.equal(ref[k], v) for k, v in m.state_dict().items()) and next(m.parameters()).device.type == "cpu"
out = D.load_on_rank0(m, lambda: "loaded")
assert out == "loaded"
try:
    D.load_on_rank0(m, lambda: (_ for _ in ()).throw(FileNotFoundError("x")))
    raise SystemExit("failing loader swallowed")
except FileNotFoundError:
    pass
D.barrier()
assert D.max_over_ranks(3.5, torch.device("cuda", l)) == 3.5
torch.distributed.destroy_process_group()
print("RCCL_OK")
'''
    script = tmp_path / "w.py"
    script.write_text(code)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29782", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
