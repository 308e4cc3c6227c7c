"""Data-parallel clip sharding: one process per GPU, zero collectives inside the DDIM loop.

The reference shards prompts with DistributedSampler and makes every rank read every checkpoint from
disk (reference scripts/inference.py:44-51, 260, 430; no collective is ever issued).  Here rank 0 may load the
weights once and broadcast them over RCCL (xGMI): `broadcast_packed` for an engine's packed tree (bench.py),
`load_on_rank0` / `broadcast_module` for the drop-in nn.Modules - explicit calls, never hidden in a forward;
clips are then independent DDIM trajectories with no collective.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist

from .engine.weights import Packed


def init_from_env(backend: Optional[str] = None) -> tuple:
    """(rank, world, local_rank); initialises torch.distributed from RANK/WORLD_SIZE/MASTER_* if world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    # under a launcher (torch.distributed.run sets TORCHELASTIC_RUN_ID) the group is opened even for ONE rank: a 1-GPU box then
    # exercises the same RCCL init / broadcast / barrier / all-reduce calls an 8-GPU node will make (tests/test_distributed_gpu.py)
    launched = "TORCHELASTIC_RUN_ID" in os.environ and "RANK" in os.environ
    if (world > 1 or launched) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """clip i -> rank i mod world (the CFG pair of a clip stays on one GPU)."""
    return list(range(rank, n_items, world))


def _leaves(node, prefix=""):
    if isinstance(node, torch.Tensor):
        yield prefix, node
    elif isinstance(node, dict):
        for k in sorted(node.keys(), key=str):
            if k in ("cfg", "dtype"):
                continue
            yield from _leaves(node[k], f"{prefix}.{k}" if prefix else str(k))
    elif isinstance(node, (list, tuple)):
        for i, v in enumerate(node):
            yield from _leaves(v, f"{prefix}.{i}")


def packed_tensors(P: Packed) -> List[tuple]:
    """deterministic (name, tensor) walk of a packed parameter tree"""
    return list(_leaves(P))


def _staging_device() -> Optional[torch.device]:
    """RCCL ("nccl") moves device memory only: host-resident tensors (a model that was loaded but not yet moved - the reference
    calls `unet.load_state_dict` at scripts/inference.py:178 and `.to("cuda")` at :213) are staged through the current GPU.
    Decided by what the group can move, not by string equality with "nccl": a group opened elsewhere with the composite default
    ("cpu:gloo,cuda:nccl") moves host tensors itself and needs no staging; a pure-nccl group does."""
    backend = str(dist.get_backend())
    if "nccl" in backend and "gloo" not in backend and "mpi" not in backend:
        return torch.device("cuda", torch.cuda.current_device())
    return None


def broadcast_packed(P: Packed, src: int = 0, bucket_bytes: int = 256 << 20) -> int:
    """Broadcast every tensor of the tree from `src`.  Small tensors are coalesced into flat buckets
    (xGMI is point-to-point: few large transfers beat ~3000 tiny ones).  Tensors may live on the host or on the GPU, in any
    mix: under the nccl backend each bucket is assembled on the current GPU, broadcast, and copied back.  Returns bytes moved."""
    if not dist.is_initialized():
        return 0
    stage = _staging_device()
    total = 0
    groups = {}
    for name, t in packed_tensors(P):
        groups.setdefault(t.dtype, []).append(t)
    for dtype, ts in groups.items():
        bucket, size = [], 0
        def flush():
            nonlocal bucket, size
            if not bucket:
                return
            dev = stage if stage is not None else bucket[0].device
            if dist.get_rank() == src:
                flat = torch.cat([t.detach().reshape(-1).to(dev) for t in bucket])
            else:        # receivers do not upload what the broadcast overwrites (a full-model H2D per rank otherwise)
                flat = torch.empty(sum(t.numel() for t in bucket), dtype=bucket[0].dtype, device=dev)
            dist.broadcast(flat, src=src)
            off = 0
            for t in bucket:
                n = t.numel()
                t.copy_(flat[off:off + n].reshape(t.shape))
                off += n
            bucket, size = [], 0
        for t in ts:
            nbytes = t.numel() * t.element_size()
            total += nbytes
            if nbytes >= bucket_bytes:
                flush()
                if stage is not None and t.device != stage:
                    big = t.detach().to(stage) if dist.get_rank() == src else torch.empty(t.shape, dtype=t.dtype, device=stage)
                    dist.broadcast(big, src=src)
                    t.copy_(big)
                else:
                    dist.broadcast(t, src=src)
                continue
            if size + nbytes > bucket_bytes:
                flush()
            bucket.append(t)
            size += nbytes
        flush()
    return total


def broadcast_module(module: torch.nn.Module, src: int = 0, bucket_bytes: int = 256 << 20) -> int:
    """Overwrite every parameter and buffer of `module` with rank `src`'s, in place (bucketed like `broadcast_packed`).  An
    EXPLICIT collective: every rank must call it, at the same point of the script, for the same module.  It covers whatever
    the module holds - UNet3D / UNet2D (with `image_proj_model`), both halves of AutoencoderKL, Resampler, ImageProjModel,
    the CLIP wrappers - because it moves the state dict itself, not a packed engine tree; the engines re-pack at the next
    forward (in-place copies bump the tensor versions they key on).  Returns bytes moved (0 without a process group)."""
    if not dist.is_initialized():
        return 0
    tree = Packed({k: v for k, v in module.state_dict(keep_vars=True).items() if isinstance(v, torch.Tensor)})
    with torch.no_grad():
        return broadcast_packed(tree, src=src, bucket_bytes=bucket_bytes)


def load_on_rank0(module, loader=None, *args, **kwargs):
    """`load_on_rank0(module, loader, *args)`: run a checkpoint loader (e.g. `lambda: unet.load_state_dict(torch.load(path))`) on
    rank 0 only, then broadcast the module's parameters and buffers from rank 0 (`broadcast_module`) - one disk read instead of
    the reference's N (scripts/inference.py:44-51, 152-181).  Every rank must make the call.  Returns the loader's result on
    rank 0, None elsewhere.  Without a process group it just runs the loader.
    (Round-2 form `load_on_rank0(loader)`, which relied on a broadcast hidden in the first forward, is gone: that broadcast
    covered only part of the modules and could deadlock rank-divergent scripts.)"""
    if loader is None or not isinstance(module, torch.nn.Module):
        raise TypeError("load_on_rank0(module, loader, *args): pass the nn.Module whose weights the loader fills, then the loader")
    if not dist.is_initialized():
        return loader(*args, **kwargs)
    # rank 0's loader may raise (missing file, key mismatch): the others must not be left waiting in the weight broadcast, so a
    # status word goes first and every rank raises together
    out, err = None, None
    if dist.get_rank() == 0:
        try:
            out = loader(*args, **kwargs)
        except Exception as e:           # noqa: BLE001 - re-raised below, on every rank
            err = e
    dev = _staging_device() or torch.device("cpu")
    flag = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=dev)
    dist.broadcast(flag, src=0)
    if int(flag.item()) != 0:
        if err is not None:
            raise err
        raise RuntimeError("load_on_rank0: the loader failed on rank 0 (see its traceback); no weights were broadcast")
    broadcast_module(module, src=0)
    return out


def barrier() -> None:
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
