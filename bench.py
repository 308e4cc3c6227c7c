#!/usr/bin/env python
"""Benchmark of the FollowYourClick denoising hot path on MI355X.

metric  : denoised frames/sec, 16f x 512^2 clip @ 25 DDIM steps (BASELINE.json)
step    : ONE clip per GPU = the whole 25-step DDIM loop (latents in -> latents out: 9-channel input
          build, CFG-pair UNet3D forward, guidance, DDIM update) on synthetic (1,4,16,64,64) latents,
          random-init SD-1.5 UNet3D + motion modules (no checkpoints exist offline), bf16.
value   : whole-job frames/s = n_gpus * 16 * K / max-over-ranks(time of K clips); inputs are resident in
          HBM when the timed region starts; VAE decode is outside the metric (reported in DESIGN.md).
scaling : weak - every rank denoises its own clips; the only collective is the one-time RCCL
          broadcast of the packed weights (outside the timed region).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...

Adds to the JSON line: "roofline" (dominant kernel = the MFMA GEMM/implicit-conv family, timed per launch
with HIP events in an instrumented extra pass; on every configuration's line), with --vae "roofline_vae"
(the same family inside the VAE decode), "cpu_baseline" (whole-clip DDIM steps at the same resolution, one untimed +
--cpu-steps timed, threads bound to one NUMA node; rank 0, N=1 only: kind "reference" = the unmodified reference's own CPU path from the
byte copies under oracle/_ref/, in a subprocess; kind "port" = the CPU oracle, when the reference files are absent or its run fails) and "parity" (rel-L2 of THIS
binary and dtype against the reference's own cfg2 trajectory, tests/golden/cfg2_trajectory.npz).
`--gpus N` without a launcher re-executes itself under torch.distributed.run; a world size that differs from --gpus is an error.
`metric` and `config.workload` are derived from the arguments: only the default arguments produce BASELINE.json's
configs[1] line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from followyourclick_amd import distributed as D  # noqa: E402
from followyourclick_amd.engine import DDIMConfig, UNet3DConfig  # noqa: E402
from followyourclick_amd.engine.sampler import DDIMSampler  # noqa: E402
from followyourclick_amd.engine.schema import random_state_dict, unet_schema  # noqa: E402
from followyourclick_amd.engine.unet3d import UNet3DEngine  # noqa: E402
from followyourclick_amd.engine.weights import pack_unet  # noqa: E402
from followyourclick_amd.profiling import TimedOps  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def synthetic_inputs(cfg, frames, h, w, seed, device):
    g = torch.Generator().manual_seed(seed)
    lat = torch.randn(1, cfg.in_channels, frames, h, w, generator=g)
    first = 0.18215 * 5.0 * torch.randn(1, cfg.in_channels, h, w, generator=g)
    mask = torch.zeros(1, 1, 1, h, w)
    mask[..., h // 4: 3 * h // 4, w // 4: 3 * w // 4] = 1.0          # centred rectangle
    text = torch.randn(2, 77, cfg.cross_attention_dim, generator=torch.Generator().manual_seed(3000))
    return dict(latents=lat.to(device), first=first.to(device), mask=mask.to(device), text=text.to(device))


def numa_node_cores(node: int = 0):
    """(logical cpu ids of the physical cores of one NUMA node - one SMT sibling each -, logical cpus of the host).  The oracle's
    fp32 matmuls gain nothing from hyper-threads and LOSE from threads spread over sockets: on the round-3 box 128 threads took
    10.9 s per frame where BASELINE.md's 8 cores take 6.5 s."""
    def parse(txt):
        out = []
        for part in txt.strip().split(","):
            if "-" in part:
                lo, hi = part.split("-")
                out.extend(range(int(lo), int(hi) + 1))
            elif part:
                out.append(int(part))
        return out
    try:
        allowed = set(os.sched_getaffinity(0))
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            cpus = [c for c in parse(f.read()) if c in allowed]
        firsts = []
        for c in cpus:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as f:
                if min(parse(f.read())) == c:
                    firsts.append(c)
        if firsts:
            return firsts, os.cpu_count() or len(cpus)
    except OSError:
        pass
    n = os.cpu_count() or 2
    return list(range(max(1, n // 2))), n


CPU_THREADS_MAX = 32          # past ~32 threads the oracle's many small ops stop scaling (profiles/r04_cpu_baseline_thread_sweep.txt)
REFERENCE_CPU_S_PER_FRAME = 6.5   # the reference's own CPU path, 8 cores, seconds per frame per DDIM step at 512^2


def cpu_baseline_reference(frames, size, ddim_steps, timed_steps, timeout_s):
    """kind "reference": the UNMODIFIED reference UNet3DConditionModel (byte copies under the git-ignored oracle/_ref/) on the host cores, in a
    subprocess (`python -m oracle.gpu_reference --cpu-baseline`, test infrastructure), threads bound like the port's.  None when the reference
    files are not staged or the subprocess fails / times out - the caller then falls back to the port."""
    import subprocess
    if not (os.path.exists(os.path.join(ROOT, "oracle", "_ref", "animatediff", "models", "unet.py")) or os.path.isdir("/root/reference/animatediff")):
        return None, "reference model files not staged under oracle/_ref/"
    cores, logical = numa_node_cores(0)
    node_cores = len(cores)
    cores = cores[:CPU_THREADS_MAX]
    cmd = [sys.executable, "-m", "oracle.gpu_reference", "--cpu-baseline", "--frames", str(frames), "--size", str(size), "--ddim-steps", str(ddim_steps),
           "--timed", str(max(1, timed_steps)), "--cores", ",".join(str(c) for c in cores)]
    try:
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout_s, env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
    except subprocess.TimeoutExpired:
        return None, f"the reference's CPU run did not finish in {timeout_s}s"
    for line in r.stdout.splitlines():
        if line.startswith("CPU_REFERENCE "):
            d = json.loads(line[len("CPU_REFERENCE "):])
            d.update(cores_on_numa_node0=node_cores, logical_cpus_on_host=logical, reference_cpu_s_per_frame_per_ddim_step=REFERENCE_CPU_S_PER_FRAME)
            return d, None
    return None, (r.stderr or r.stdout)[-200:]


def cpu_baseline(sd, frames, h, w, ddim_steps, timed_steps=1):
    """The oracle (CPU port of the reference math, fp32) on the STATED workload: CFG-pair UNet3D forwards (= DDIM
    steps) of the whole clip - all `frames` frames, so the cross-frame GroupNorms and the temporal attention run on the benchmark's
    own problem - at the benchmark resolution, threads BOUND to the physical cores of NUMA node 0 (at most CPU_THREADS_MAX).  One
    untimed step at the same shape first (oneDNN primitive creation, allocator growth), then `timed_steps` timed ones;
    frames/s = frames / (ddim_steps x seconds per step).  About 1.5 min per step on a 32-core node."""
    from oracle import functional as Fn  # test infrastructure: used only as the reported CPU baseline
    cfg = Fn.UNetConfig()
    g = torch.Generator().manual_seed(1)
    text = torch.randn(2, 77, cfg.cross_attention_dim, generator=g)
    fps, flow = torch.tensor([2, 2]), torch.tensor([4, 4])
    cores, logical = numa_node_cores(0)
    node_cores = len(cores)
    cores = cores[:CPU_THREADS_MAX]
    before_threads = torch.get_num_threads()
    before_aff = os.sched_getaffinity(0)
    try:
        os.sched_setaffinity(0, set(cores))
        torch.set_num_threads(len(cores))

        def fwd(i):
            x9 = torch.randn(2, cfg.conv_in_channels, frames, h, w, generator=g)
            t0 = time.time()
            with torch.no_grad():
                Fn.unet3d_forward(sd, cfg, x9, torch.tensor(961 - 40 * i), text, fps, flow)
            return time.time() - t0
        t_warm = fwd(0)
        times = [fwd(1 + i) for i in range(max(1, timed_steps))]
    finally:
        os.sched_setaffinity(0, before_aff)
        torch.set_num_threads(before_threads)
    dt = sum(times) / len(times)
    return dict(value=frames / (ddim_steps * dt), unit="frames/s", cores=len(cores), cores_on_numa_node0=node_cores, logical_cpus_on_host=logical, kind="port",
                s_per_frame_per_ddim_step=round(dt / frames, 2), reference_cpu_s_per_frame_per_ddim_step=REFERENCE_CPU_S_PER_FRAME,
                sample=f"{len(times)} timed DDIM step(s) (CFG-pair UNet3D forward at {h * 8}x{w * 8}, fp32 oracle) on ALL {frames} frames of the clip, "
                       f"{dt:.1f}s per step = {dt / frames:.2f} s/frame (BASELINE.md: the reference's CPU path, 8 cores, {REFERENCE_CPU_S_PER_FRAME} s/frame), after one "
                       f"untimed step at the same shape ({t_warm:.1f}s); {len(cores)} threads bound to physical cores of NUMA node 0 ({node_cores} there, "
                       f"{logical} logical cpus on the host); frames/s = {frames} / ({ddim_steps} x {dt:.1f}s)")


def parity_leg(args, dtype, device):
    """rel-L2 of THIS library binary in THIS dtype against the real reference's own trajectory of the benchmarked workload
    (tests/golden/cfg2_trajectory.npz: AnimationPipeline.__call__ of /root/reference on seeded weights and inputs, f32 and under
    bf16 autocast, latents after DDIM steps 0 / 4 / 24; recipe oracle/make_golden_full.py cfg2).  One extra clip on the golden's
    weights, after the timed region.  The seeded weight / input generators live under oracle/ (test infrastructure): they are used
    here only to reproduce the golden's inputs - the checker, never the thing measured."""
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "cfg2_trajectory.npz")
    if not os.path.exists(path):
        return {"value": None, "note": "tests/golden/cfg2_trajectory.npz not present"}
    from oracle import functional as Fn
    from oracle import weights as W
    g = np.load(path)
    ocfg = Fn.UNetConfig()
    sd = W.make_weights(W.unet_state_shapes(ocfg), seed=int(g["weight_seed"]))
    eng = UNet3DEngine(pack_unet(sd, UNet3DConfig(), dtype, device))
    del sd
    F, lat, steps = int(g["frames"]), int(g["lat"]), int(g["steps"])
    inp = W.seeded_inputs(ocfg, 1, F, lat, lat, seed=int(g["input_seed"]))
    keep = [int(k) for k in g["keep"]]
    got = {}

    def cb(i, t, l):
        if i in keep:
            got[i] = l.detach().float().cpu()
    DDIMSampler(eng, DDIMConfig()).sample(inp["latents"], torch.from_numpy(g["text_embeddings"]), steps, 8.0, inp["first_image_latents"],
                                          inp["first_images_mask"], fps=[2], flow=[4], callback=cb)
    torch.cuda.synchronize()

    def rel(a, b):
        b = torch.from_numpy(b)
        return float((a - b).norm() / b.norm())
    out = {"metric": "rel-L2 of the latents vs the reference pipeline's own run on identical seeds / inputs", "dtype": args.dtype,
           "golden": "tests/golden/cfg2_trajectory.npz (oracle/make_golden_full.py cfg2: the real AnimationPipeline, 16f@512^2, 25 DDIM steps)",
           "north_star_tolerance": 1e-3}
    for i in keep:
        out[f"step{i}_vs_ref_f32"] = float(f"{rel(got[i], g[f'step{i}_f32']):.3e}")
        out[f"step{i}_ref_bf16_autocast_vs_ref_f32"] = float(f"{float(g[f'drift{i}']):.3e}")
    out["value"] = out[f"step{keep[-1]}_vs_ref_f32"]
    out["within_north_star_tolerance"] = bool(out["value"] <= 1e-3)
    del eng
    torch.cuda.empty_cache()
    return out


def gpu_reference(args):
    """The reference's OWN GPU path on this chip, beside the CPU number: the unmodified UNet3DConditionModel (byte copies under the
    git-ignored oracle/_ref/, staged by oracle/stage_ref_scripts.py) in eager PyTorch-ROCm under torch.autocast, 1 untimed + 2 timed
    CFG-pair forwards of the benchmark shape, as a SUBPROCESS (oracle/gpu_reference.py, test infrastructure): a crash or a timeout
    there costs this field, never the line.  A reported baseline, not the target."""
    import subprocess
    if not (os.path.exists(os.path.join(ROOT, "oracle", "_ref", "animatediff", "models", "unet.py")) or os.path.isdir("/root/reference/animatediff")):
        return {"value": None, "note": "reference model files not staged under oracle/_ref/ (python -m oracle.stage_ref_scripts, build container only)"}
    cmd = [sys.executable, "-m", "oracle.gpu_reference", "--json", "--frames", str(args.frames), "--size", str(args.size), "--ddim-steps", str(args.ddim_steps),
           "--dtype", args.dtype, "--attention", "sdpa", "--timed", "2"]
    try:
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=args.gpu_reference_timeout)
    except subprocess.TimeoutExpired:
        return {"value": None, "error": f"oracle.gpu_reference did not finish in {args.gpu_reference_timeout}s"}
    for line in r.stdout.splitlines():
        if line.startswith("GPU_REFERENCE "):
            return json.loads(line[len("GPU_REFERENCE "):])
    return {"value": None, "error": (r.stderr or r.stdout)[-300:]}


def workload_label(args):
    """BASELINE.json's configs entry these arguments reproduce, or 'custom'"""
    key = (args.frames, args.size, args.ddim_steps, args.ip_tokens)
    names = {(8, 256, 5, 0): "configs[0]", (16, 512, 25, 0): "configs[1]", (32, 768, 50, 0): "configs[3]", (16, 512, 25, 16): "configs[4]"}
    label = names.get(key, "custom (no BASELINE.json config)")
    dt = getattr(args, "dtype", "bf16")
    if dt != "bf16" and key in names:
        label += f" at {dt} instead of the config's bf16"
    return label


def _self_launch(n, argv):
    """re-exec this very script (sys.argv[0]: bench.py, or the tests-side wrapper around it) under torch.distributed.run"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(sys.argv[0]), *argv]
    print(f"[bench] --gpus {n} without a launcher: starting {' '.join(cmd[1:8])} ...", file=sys.stderr)
    rc = subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))
    if rc != 0:
        raise SystemExit(rc)
    return None


def rccl_ranks(device) -> int:
    """ranks that actually took part in a collective on the data-path backend: an all-reduce of ones (1 without a process group)"""
    if not torch.distributed.is_initialized():
        return 1
    t = torch.ones(1, dtype=torch.float32, device=device)
    torch.distributed.all_reduce(t)
    return int(round(float(t.item())))


def main(argv=None, emulation=None):
    """emulation: None in the product.  tests/bench_emulated.py passes {"ops": <op emulator>, "cfg": <tiny UNet3DConfig>} to drive
    this same entry point (rendezvous, barrier, max-over-ranks timing, JSON line) with gloo on CPU ranks; its line says
    "data": "emulated" and is no measurement."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed clips per GPU (one step = one 25-step DDIM clip)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--ddim-steps", type=int, default=25)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"],
                    help="storage / MFMA operand type: bf16 (BASELINE configs[1], the default), f16 (the reference's deployed autocast precision), f32 (parity mode)")
    ap.add_argument("--ip-tokens", type=int, default=0, help="configs[4]: IP-Adapter decoupled cross-attention with this many image tokens")
    ap.add_argument("--vae", dest="vae", action="store_true", default=True,
                    help="time the VAE decode of the final latents (outside the metric: vae_decode_ms, roofline_vae, end_to_end_frames_per_sec); ON by default since round 6 (SURVEY 8d)")
    ap.add_argument("--no-vae", dest="vae", action="store_false")
    ap.add_argument("--graph", action="store_true", help="replay DDIM steps 1..n-1 from one captured hipGraph (A/B vs eager launches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=2, help="timed DDIM steps of the cpu_baseline leg (after one untimed step; BASELINE.md 3 asks for 2)")
    ap.add_argument("--cpu-port", action="store_true", help="cpu_baseline from the oracle port (kind 'port') instead of the staged reference itself (kind 'reference')")
    ap.add_argument("--cpu-timeout", type=int, default=600, help="seconds the reference's CPU run may take before the port is used instead")
    ap.add_argument("--no-parity", action="store_true", help="skip the `parity` field (one extra clip against tests/golden/cfg2_trajectory.npz)")
    ap.add_argument("--no-gpu-reference", action="store_true",
                    help="skip the `gpu_reference` leg (the UNMODIFIED reference UNet3D, eager PyTorch-ROCm under torch.autocast, from the byte copies under oracle/_ref/)")
    ap.add_argument("--gpu-reference-timeout", type=int, default=420, help="seconds the gpu_reference subprocess may take")
    args = ap.parse_args(argv)
    emulate = emulation is not None

    # `python bench.py --gpus N` with no launcher around it: become the launcher (one rank per GPU over RCCL, rendezvous on
    # 127.0.0.1) instead of silently running one rank - the reference shards inside the script it is handed
    # (/root/reference/scripts/inference.py:44-51, 260)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return _self_launch(args.gpus, argv if argv is not None else sys.argv[1:])
    rank, world, local = D.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to print a line for a different job size")
    if not emulate and torch.cuda.is_available() and torch.cuda.device_count() < (int(os.environ.get("LOCAL_WORLD_SIZE", world))):
        raise SystemExit(f"[bench] {torch.cuda.device_count()} HIP device(s) visible for {world} ranks on this node: one process drives one GPU")
    if emulate:
        from followyourclick_amd import ops as ops_mod
        ops_mod.impl = emulation["ops"]
        device = torch.device("cpu")
        sync = lambda: None      # noqa: E731
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device (no CPU fallback in the product path)")
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
        sync = torch.cuda.synchronize
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    is16 = args.dtype in ("bf16", "f16")                       # v_mfma_*_bf16 and v_mfma_*_f16 have the same dense peak on gfx950

    cfg = UNet3DConfig(use_ip_cross_attention=args.ip_tokens > 0, ip_num_tokens=max(args.ip_tokens, 4))
    if emulate:
        cfg = emulation["cfg"]
    if args.frames > cfg.temporal_position_encoding_max_len:
        cfg.temporal_position_encoding_max_len = args.frames
    schema = unet_schema(cfg)
    sd = random_state_dict(schema, seed=0, materialize=(rank == 0))
    packed = pack_unet(sd, cfg, dtype, device)
    t_b = time.time()
    moved = D.broadcast_packed(packed, src=0)          # one-time weight broadcast over RCCL/xGMI
    sync()
    t_b = time.time() - t_b
    if rank != 0:
        sd = None
    eng = UNet3DEngine(packed)
    sampler = DDIMSampler(eng, DDIMConfig())
    h = w = args.size // 8
    clips = [synthetic_inputs(cfg, args.frames, h, w, 1000 + rank * 100 + i, device) for i in range(args.warmup + args.steps)]

    ip = None
    if args.ip_tokens > 0:
        ip = torch.randn(2, args.ip_tokens, cfg.cross_attention_dim, generator=torch.Generator().manual_seed(4000)).to(device)

    def run(c):
        return sampler.sample(c["latents"], c["text"], args.ddim_steps, 8.0, c["first"], c["mask"], fps=[2], flow=[4], ip_tokens=ip,
                              use_graph=True if args.graph else None)

    for c in clips[: args.warmup]:
        run(c)
    D.barrier()
    sync()
    t0 = time.time()
    for c in clips[args.warmup:]:
        out = run(c)
    sync()
    D.barrier()
    elapsed = D.max_over_ranks(time.time() - t0, device)
    n_coll = rccl_ranks(device)
    if n_coll != world:
        raise SystemExit(f"[bench] an all-reduce of ones over the process group returned {n_coll}, expected {world} ranks")
    assert torch.isfinite(out).all(), "non-finite latents"
    # Host side of a DDIM step (~900 ctypes launches), outside the timed region: a SHORT run from an idle queue, so that the
    # host is timed queueing launches, not waiting for room in a full HIP queue (over a whole clip the call returns only a little
    # ahead of the GPU whatever the host's own speed: round 3 first reported that back-pressure as "host time").  Slowest rank.
    # Well below gpu_ms_per_ddim_step = the GPU is the bound and the launch path has headroom for N processes per node.
    n_host = min(4, args.ddim_steps)
    c = clips[-1]
    th = time.time()
    sampler.sample(c["latents"], c["text"], n_host, 8.0, c["first"], c["mask"], fps=[2], flow=[4], ip_tokens=ip)
    th = time.time() - th
    sync()
    host_ms = D.max_over_ranks(1000.0 * th / n_host, device)

    result = None
    if rank == 0:
        value = world * args.frames * args.steps / elapsed
        result = {
            "metric": f"denoised frames/sec, {args.frames}f x {args.size}^2 clip @ {args.ddim_steps} DDIM steps", "value": round(value, 3), "unit": "frames/s",
            "n_gpus": world, "rccl_ranks": n_coll, "collective_backend": (torch.distributed.get_backend() if torch.distributed.is_initialized() else None), "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * elapsed / args.steps, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "emulated" if emulate else "synthetic",
            "clips_per_sec": round(world * args.steps / elapsed, 4),
            "host_launch_ms_per_ddim_step": round(host_ms, 2), "gpu_ms_per_ddim_step": round(1000 * elapsed / (args.steps * args.ddim_steps), 2),
            "config": {"workload": f"{workload_label(args)}: AnimationPipeline DDIM loop, SD-1.5 UNet3D + mm_sd_v15-shaped motion modules "
                                   f"(random init), 1 clip/GPU of {args.frames} frames {args.size}x{args.size}, {args.ddim_steps} DDIM steps, "
                                   f"CFG 8.0, mask + first-frame concat, fps/flow conditioning"
                                   + (f", IP-Adapter decoupled cross-attention with {args.ip_tokens} image tokens" if args.ip_tokens > 0 else ""),
                       "frames": args.frames, "height": args.size, "width": args.size, "ddim_steps": args.ddim_steps, "ip_tokens": args.ip_tokens,
                       "parallelism": f"dp{world} (independent clips, weight broadcast {moved / 2**30:.2f} GiB in {t_b:.2f}s)"},
        }

    # ---- roofline leg: one instrumented clip step, per-launch HIP-event timing (rank 0) ---------------
    if rank == 0 and not args.no_roofline and not emulate:
        timed = TimedOps(eng.ops)
        eng.ops = timed
        c = clips[-1]
        st = sampler.prepare(c["text"], args.ddim_steps, 1, 8.0, [2], [4], ip)
        lat = c["latents"].clone()
        first, mask = c["first"].reshape(1, cfg.in_channels, h * w).contiguous(), c["mask"][:, :, 0].reshape(1, 1, h * w).contiguous()
        # one untimed instrumented step first (lazy host-side state), then n_inst timed ones.  Each timed step is queued behind
        # a ~50 ms device-side spin so that the host runs ahead of the GPU: the event pairs then bracket kernel execution only,
        # not the host's launch gaps (with ~900 ctypes launches per step the host is otherwise the slower side of this pass).
        sampler.step(st, 0, lat, first, mask)
        torch.cuda.synchronize()
        timed.reset()
        n_inst = 3
        for i in range(n_inst):
            torch.cuda._sleep(100_000_000)
            sampler.step(st, 1 + i, lat, first, mask)
        torch.cuda.synchronize()
        summ = timed.summary()
        eng.ops = timed.inner
        if os.environ.get("FYC_BENCH_SHAPES"):
            with open(os.environ["FYC_BENCH_SHAPES"], "w") as f:
                for key, n, ms_, tf in timed.by_shape():
                    f.write(f"{ms_ / n_inst:9.3f} ms/step  n={n // n_inst:4d}  {tf:7.1f} TF/s  {key}\n")
        mm = {k: v for k, v in summ.items() if k in ("gemm", "conv3x3")}
        fl = sum(v["flops"] for v in mm.values())
        ms = sum(v["ms"] for v in mm.values())
        launches = sum(v["launches"] for v in mm.values())
        ach = fl / (ms * 1e-3) / 1e12
        # HBM bytes per launch from PMC counters (collected offline with rocprofv3 --pmc, tools/collect_profiles.sh): reported only
        # when the profile was taken on THIS library binary or on a build of the same sources + flags (digests stamped by
        # tools/hbm_traffic.py; hipcc output is not bit-stable across output paths), otherwise null
        traffic, traffic_note = None, "no PMC profile of this library binary under profiles/ (tools/collect_profiles.sh regenerates it)"
        if args.frames == 16 and args.size == 512 and args.dtype == "bf16":
            import glob
            import hashlib
            from followyourclick_amd import _lib as L_
            from followyourclick_amd._build import family_digest, source_digest
            with open(L_.LIB_PATH, "rb") as f:
                digest = hashlib.sha256(f.read()).hexdigest()
            # the loaded library is a build of the sources in the tree unless FYC_LIB_PATH points elsewhere (A/B builds)
            src_digest = source_digest() if not os.environ.get("FYC_LIB_PATH") else None
            fam_digest = family_digest("gemm") if not os.environ.get("FYC_LIB_PATH") else None
            for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True):
                try:
                    doc = json.load(open(tpath))
                    same_binary = doc.get("lib_sha256") == digest
                    same_sources = bool(src_digest) and doc.get("source_sha256") == src_digest
                    same_family = bool(fam_digest) and doc.get("gemm_family_source_sha256") == fam_digest
                    if same_binary or same_sources or same_family:
                        traffic = round(doc["families"]["gemm"]["hbm_bytes_per_launch"])
                        ident = (f"this library binary (sha256 {digest[:12]})" if same_binary
                                 else f"a build of these same kernel sources and flags (source digest {src_digest[:12]})" if same_sources
                                 else f"a build whose GEMM-family sources and flags are those of the profiled library (family digest {fam_digest[:12]}; another family's source changed since)")
                        traffic_note = (f"avg HBM bytes per fyc_gemm_kernel launch, rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE on "
                                        f"{ident}, {os.path.relpath(tpath, ROOT)}")
                        break
                except Exception:
                    continue
        alg_bytes = sum(v["bytes"] for v in mm.values()) / launches
        result["roofline"] = {"kernel": "fyc_gemm_kernel (MFMA GEMM + implicit-GEMM conv3x3)", "bound": "mfma",
                              "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS if is16 else 157.3, "unit": "TFLOP/s",
                              "frac": round(ach / (PEAK_BF16_TFLOPS if is16 else 157.3), 4), "traffic": traffic,
                              "traffic_note": traffic_note,
                              "algorithmic_bytes_per_launch": round(alg_bytes),
                              "launches_per_ddim_step": launches // n_inst, "avg_launch_us": round(1e3 * ms / launches, 2),
                              "algorithmic_tflop_per_ddim_step": round(fl / n_inst / 1e12, 3)}
        fam = {}
        for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
            e = {"ms_per_ddim_step": round(v["ms"] / n_inst, 3), "launches": v["launches"] // n_inst}
            if v["flops"]:
                e["tflops"] = round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)
            if v["bytes"]:
                e["algorithmic_GBs"] = round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)
            fam[k] = e
        result["kernel_families"] = fam
        if "attn_self" in summ:
            a = summ["attn_self"]
            at = a["flops"] / (a["ms"] * 1e-3) / 1e12
            result["roofline_attention"] = {"kernel": "fyc_attn_kernel (spatial self-attention)", "bound": "mfma", "achieved": round(at, 1),
                                            "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(at / PEAK_BF16_TFLOPS, 4)}

        # the fused kernels that left the GEMM family this round, each against the same MFMA peak (algorithmic FLOPs of the
        # sub-block they replace ÷ summed launch time)
        for key, name in (("ff_block", "ff_block_kernel (fused GEGLU feed-forward block, C = 320)"),
                          ("temporal_block", "tblock_rr_kernel (fused temporal attention sub-block, C = 320)")):
            if key in summ and summ[key]["flops"] > 0 and is16:
                a = summ[key]
                at = a["flops"] / (a["ms"] * 1e-3) / 1e12
                result["roofline_" + key] = {"kernel": name, "bound": "mfma", "achieved": round(at, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                             "frac": round(at / PEAK_BF16_TFLOPS, 4), "launches_per_ddim_step": a["launches"] // n_inst,
                                             "avg_launch_us": round(1e3 * a["ms"] / a["launches"], 2)}

    if rank == 0 and args.vae and not emulate:
        from followyourclick_amd.engine import VAEDecoderConfig
        from followyourclick_amd.engine.schema import vae_decoder_schema
        from followyourclick_amd.engine.vae import VAEDecoderEngine
        from followyourclick_amd.engine.weights import pack_vae_decoder
        vcfg = VAEDecoderConfig()
        vae = VAEDecoderEngine(pack_vae_decoder(random_state_dict(vae_decoder_schema(vcfg), 1), vcfg, dtype, device))
        vae.decode_video(out)
        torch.cuda.synchronize()
        tv = time.time()
        vid = vae.decode_video(out)
        torch.cuda.synchronize()
        result["vae_decode_ms"] = round(1000 * (time.time() - tv), 1)
        result["end_to_end_frames_per_sec"] = round(args.frames / (elapsed / args.steps + result["vae_decode_ms"] / 1000), 3)
        if not args.no_roofline:     # the decode's dominant kernel = the same GEMM / implicit-conv family, timed per launch
            tvae = TimedOps(vae.ops)
            vae.ops = tvae
            torch.cuda._sleep(100_000_000)
            vae.decode_video(out)
            torch.cuda.synchronize()
            vs = tvae.summary()
            vae.ops = tvae.inner
            vm = {k: v for k, v in vs.items() if k in ("gemm", "conv3x3")}
            vfl, vms, vl = sum(v["flops"] for v in vm.values()), sum(v["ms"] for v in vm.values()), sum(v["launches"] for v in vm.values())
            peak = PEAK_BF16_TFLOPS if is16 else 157.3
            result["roofline_vae"] = {"kernel": "fyc_gemm_kernel inside VAEDecoderEngine.decode_video (implicit-GEMM conv3x3 + 1x1 / attention linears)",
                                      "bound": "mfma", "achieved": round(vfl / (vms * 1e-3) / 1e12, 1), "peak": peak, "unit": "TFLOP/s",
                                      "frac": round(vfl / (vms * 1e-3) / 1e12 / peak, 4), "traffic": None,
                                      "launches_per_decode": vl, "avg_launch_us": round(1e3 * vms / vl, 2), "family_ms_per_decode": round(vms, 2),
                                      "algorithmic_tflop_per_decode": round(vfl / 1e12, 3),
                                      "algorithmic_bytes_per_launch": round(sum(v["bytes"] for v in vm.values()) / vl),
                                      "kernel_families": {k: {"ms": round(v["ms"], 3), "launches": v["launches"]} for k, v in sorted(vs.items(), key=lambda kv: -kv[1]["ms"])}}
        assert torch.isfinite(vid).all()

    if rank == 0 and not emulate and not args.no_parity and (args.frames, args.size, args.ddim_steps, args.ip_tokens) == (16, 512, 25, 0):
        try:
            result["parity"] = parity_leg(args, dtype, device)
        except Exception as e:  # a reported extra, never a reason to lose the throughput number
            result["parity"] = {"value": None, "error": repr(e)[:200]}

    if rank == 0 and world == 1 and not emulate and not args.no_gpu_reference and args.ip_tokens == 0:
        result["gpu_reference"] = gpu_reference(args)
        if result["gpu_reference"].get("value"):
            result["gpu_reference"]["engine_over_reference"] = round(result["value"] / result["gpu_reference"]["value"], 2)

    if rank == 0 and world == 1 and not args.no_cpu_baseline and not emulate and os.environ.get("FYC_BENCH_CPU", "1") != "0":
        # the reference's OWN CPU path first (north_star: "the reference's own CPU path timed on the same box's host cores"); the oracle port
        # when the reference files are not there or its run fails
        ref, why = (None, "--cpu-port") if args.cpu_port else cpu_baseline_reference(args.frames, args.size, args.ddim_steps, args.cpu_steps, args.cpu_timeout)
        if ref is not None:
            result["cpu_baseline"] = ref
        else:
            try:
                result["cpu_baseline"] = cpu_baseline(sd, args.frames, h, w, args.ddim_steps, args.cpu_steps)
                result["cpu_baseline"]["reference_run"] = f"not used: {why}"
            except Exception as e:  # the baseline is a reported extra, never a reason to lose the GPU number
                result["cpu_baseline"] = {"value": None, "error": repr(e)[:200]}

    if rank == 0:
        print(json.dumps(result))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
