import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _fresh_library():
    """Rebuild libfyc_hip.so when a source changed (no-op when the object digests match)."""
    import shutil
    from followyourclick_amd import _build
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        _build.build(verbose=False)
    yield


class _GpuTurn:
    """this session's side of oracle/gpu_reference.py::GpuTurn: a GPU test holds flock(<ref dir>/gpu.lock) while it runs and does not start
    while the reference subprocess has asked for the chip (`ref_wants_gpu`); released around waits for the subprocess's outputs"""

    def __init__(self):
        self.fd = None

    def acquire(self):
        d = _REF_PROC.get("dir")
        if d is None or self.fd is not None:
            return
        import fcntl
        import time
        flag, proc = os.path.join(d, "ref_wants_gpu"), _REF_PROC.get("proc")
        while True:
            while os.path.exists(flag) and proc is not None and proc.poll() is None:
                time.sleep(0.1)
            fd = os.open(os.path.join(d, "gpu.lock"), os.O_CREAT | os.O_RDWR)
            fcntl.flock(fd, fcntl.LOCK_EX)
            if os.path.exists(flag) and proc is not None and proc.poll() is None:      # asked for between our check and our lock: its turn
                fcntl.flock(fd, fcntl.LOCK_UN)
                os.close(fd)
                continue
            self.fd = fd
            return

    def release(self):
        if self.fd is None:
            return
        import fcntl
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        except Exception:
            pass
        fcntl.flock(self.fd, fcntl.LOCK_UN)
        os.close(self.fd)
        self.fd = None


_TURN = _GpuTurn()


@pytest.fixture(autouse=True)
def _library_state_survives_the_test(request):
    """Around every GPU test: this session's turn on the chip (see _GpuTurn).  After it: the zero page fyc_init() was given (conv padding
    taps, rows past M, masked attention tiles all read it) still holds zeros - a test that broke it would change the numbers of every later
    test of the session without failing itself."""
    is_gpu = request.node.get_closest_marker("gpu") is not None
    if is_gpu:
        _TURN.acquire()
    try:
        yield
    finally:
        if is_gpu:
            _TURN.release()
    if not is_gpu:
        return
    from followyourclick_amd import ops
    o = ops.impl
    if o is None or getattr(o, "_zero", None) is None:
        return
    import torch
    torch.cuda.synchronize()
    assert not bool(o._zero.any()), f"{request.node.nodeid} left the library's zero page non-zero"


# ---- session-wide caches for the full-width GPU tests (round 6: the suite built the same 1.28 B-parameter seeded state dict 8 times
# - 22 s each - and packed the same engines up to 5 times) -------------------------------------------------------------------------------
class _FullWidthCache:
    """seeded oracle-side state dicts and the engines packed from them, keyed by configuration: test infrastructure"""

    def __init__(self):
        self.sd, self.eng, self._cfgs = {}, {}, {}

    def weights(self, ocfg, seed=0):
        from oracle import weights as W
        key = (repr(ocfg), int(seed))
        if key not in self.sd:
            # a configuration that differs from a cached one only in the length of the positional table has the SAME random tensors
            # (make_weights draws nothing for `pos_encoder.pe`): copy them, recompute the tables - 22 s of seeded randn saved
            import dataclasses
            shapes = W.unet_state_shapes(ocfg)
            for (orepr, oseed), osd in list(self.sd.items()):
                other = self._cfgs.get(orepr)
                if oseed == int(seed) and other is not None and dataclasses.replace(other, temporal_position_encoding_max_len=ocfg.temporal_position_encoding_max_len) == ocfg \
                        and list(osd) == list(shapes):
                    self.sd[key] = {k: (W.positional_encoding(shp[2], shp[1])[None].clone() if k.endswith("pos_encoder.pe") else osd[k]) for k, shp in shapes.items()}
                    break
            else:
                self.sd[key] = W.make_weights(shapes, int(seed))
            self._cfgs[repr(ocfg)] = ocfg
        return self.sd[key]

    def engine(self, ocfg, ecfg, dtype, seed=0, dev="cuda:0"):
        """UNet3DEngine over make_weights(unet_state_shapes(ocfg), seed) packed for ecfg in dtype.  Engines are shared between tests:
        every test calls prepare_context / the sampler's prepare before it runs a forward, nothing else of an engine is test state."""
        from followyourclick_amd.engine.unet3d import UNet3DEngine
        from followyourclick_amd.engine.weights import pack_unet
        key = (repr(ecfg), int(seed), str(dtype), dev)
        if key not in self.eng:
            self.eng[key] = UNet3DEngine(pack_unet(self.weights(ocfg, seed), ecfg, dtype, dev))
        return self.eng[key]


@pytest.fixture(scope="session")
def fullwidth():
    cache = _FullWidthCache()
    yield cache
    cache.eng.clear()
    cache.sd.clear()


# ---- the same-device reference (tests/test_reference_gpu.py): ONE background subprocess per session ---------------------------------------
# The reference's `animatediff` / `diffusers` packages and the drop-in packages of the same names cannot share an interpreter, so the
# reference side runs as `python -m oracle.gpu_reference --dump a,b,c --out-dir D`.  Round 5 started one such process per test and
# waited for it (270 s of the 744-s suite: three model builds, f32 forwards at 32f@768^2).  Now the process starts when the collection
# is done and computes while the kernel tests run; the tests that need a dump wait for its `<what>.pt`.
_REF_PROC = {}
_REF_WHAT = (("small", ("small_case", "test_device_reference", "test_real_autocast", "test_engine_vs_same_device")),
             ("cfg4ip", ("test_cfg4_full_shape_ip_trajectory",)), ("cfg3", ("test_cfg3_full_shape_trajectory",)))


def pytest_collection_finish(session):
    items = [it for it in session.items if "test_reference_gpu" in it.nodeid]
    if not items or session.config.option.collectonly:
        return
    staged = os.path.join(ROOT, "oracle", "_ref", "animatediff", "models", "unet.py")
    if not (os.path.exists(staged) or os.path.isdir("/root/reference/animatediff")):
        return
    try:
        import torch
        if not torch.cuda.is_available():
            return
    except Exception:
        return
    import subprocess
    import tempfile
    what = [w for w, keys in _REF_WHAT if any(any(k in it.nodeid for k in keys) for it in items)]
    if not what:
        return
    out_dir = tempfile.mkdtemp(prefix="fyc_ref_")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    # the host side of the subprocess (seeded weights of 1.28 B parameters, state-dict load) runs beside the tests: a bounded thread pool keeps it
    # from taking the cores the tests' CPU oracles need (round 6: CPU-side tests ran 3-5x slower beside an unbounded one); MIOpen's fast find mode:
    # the reference's F.conv2d calls are run once or twice per shape, an exhaustive search per shape is the larger part of their time
    env.setdefault("OMP_NUM_THREADS", "16")
    env.setdefault("MIOPEN_FIND_MODE", "FAST")
    log = open(os.path.join(out_dir, "log.txt"), "w")
    proc = subprocess.Popen([sys.executable, "-m", "oracle.gpu_reference", "--dump", ",".join(what), "--out-dir", out_dir], cwd=ROOT, env=env,
                            stdout=log, stderr=subprocess.STDOUT)
    _REF_PROC.update(proc=proc, dir=out_dir, what=what, log=log)


def pytest_sessionfinish(session, exitstatus):
    proc = _REF_PROC.get("proc")
    if proc is not None:
        if proc.poll() is None:
            proc.kill()            # the exact PID this session started
            proc.wait()
        _REF_PROC["log"].close()
        import shutil
        try:      # (what the reference side spent its time on: kept beside the parity report)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            shutil.copyfile(os.path.join(_REF_PROC["dir"], "log.txt"), os.path.join(ROOT, "gpurun_out", "reference_subprocess_log.txt"))
        except OSError:
            pass
        shutil.rmtree(_REF_PROC["dir"], ignore_errors=True)
        _REF_PROC.clear()


@pytest.fixture(scope="session")
def device_reference():
    """device_reference(what) -> the dict `oracle.gpu_reference --dump what` stored, from the session's background process"""
    import time

    def get(what, timeout=1500):
        if "proc" not in _REF_PROC:
            pytest.skip("reference model files not staged (python -m oracle.stage_ref_scripts, container only)")
        import torch
        proc, path = _REF_PROC["proc"], os.path.join(_REF_PROC["dir"], what + ".pt")
        assert what in _REF_PROC["what"], (what, _REF_PROC["what"])
        t0 = time.time()
        _TURN.release()                       # the subprocess may need the chip to produce what we wait for
        while not os.path.exists(path):
            if proc.poll() is not None and not os.path.exists(path):
                _REF_PROC["log"].flush()
                raise AssertionError("the reference subprocess ended without " + what + ".pt:\n" + open(os.path.join(_REF_PROC["dir"], "log.txt")).read()[-3000:])
            assert time.time() - t0 < timeout, f"no {what}.pt after {timeout} s"
            time.sleep(1.0)
        _TURN.acquire()
        return torch.load(path)
    return get
