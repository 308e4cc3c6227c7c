"""CPU checks of the test infrastructure behind tests/test_reference_gpu.py (oracle/gpu_reference.py): the stand-in `xformers` module's
chunked attention equals the unchunked one, and `reference_trajectory` - the loop body of AnimationPipeline.__call__ around the REAL
UNet3DConditionModel + DDIMScheduler - reproduces the oracle's pinned denoising loop.  The reference is imported in a SUBPROCESS (its
`animatediff` / `diffusers` packages and the drop-in's cannot share an interpreter with the rest of the suite)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAVE_REF = os.path.isdir("/root/reference/animatediff") or os.path.exists(os.path.join(ROOT, "oracle", "_ref", "animatediff", "models", "unet.py"))

SCRIPT = r"""
import sys, torch
from oracle import gpu_reference as G, functional as Fn, weights as W
cfg, unet = G.build_reference_unet("cpu", attention="eager", ocfg=Fn.tiny_unet_config())
inp = W.seeded_inputs(cfg, 1, 3, 8, 8, seed=7)
tr = G.reference_trajectory(unet, inp, 5, 3, None)
sd = W.make_weights(W.unet_state_shapes(cfg), seed=0)
got = {}
with torch.no_grad():
    Fn.denoise(sd, cfg, Fn.DDIMConfig(), inp["latents"].clone(), inp["text"], 5, 8.0, inp["first_image_latents"], inp["first_images_mask"],
               torch.tensor([2]), torch.tensor([4]), callback=lambda i, t, l: got.__setitem__(i, l.clone()))
print("REL", max(float((tr[i] - got[i]).norm() / got[i].norm()) for i in range(3)))
"""


def test_chunked_sdpa_stand_in_equals_unchunked(monkeypatch):
    from oracle import gpu_reference as G
    monkeypatch.delitem(sys.modules, "xformers", raising=False)
    monkeypatch.delitem(sys.modules, "xformers.ops", raising=False)
    G.install_sdpa_xformers()
    import xformers.ops as xo
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(6, 40, 16, generator=g) for _ in range(3))
    whole = torch.nn.functional.scaled_dot_product_attention(q, k, v)
    assert torch.allclose(xo.memory_efficient_attention(q, k, v), whole)
    # force the chunked path: two (batch x head) slices per call
    real = torch.nn.functional.scaled_dot_product_attention
    calls = []
    monkeypatch.setattr(torch.nn.functional, "scaled_dot_product_attention", lambda a, b, c, **kw: (calls.append(a.shape[0]), real(a, b, c, **kw))[1])
    monkeypatch.setattr(G, "SCORE_BYTES_LIMIT", 2 * 40 * 40 * 4 * 2)
    bias = torch.randn(6, 40, 40, generator=g)
    out = xo.memory_efficient_attention(q, k, v, attn_bias=bias)
    assert calls == [2, 2, 2]
    assert torch.allclose(out, real(q, k, v, attn_mask=bias), atol=1e-6)
    monkeypatch.delitem(sys.modules, "xformers", raising=False)
    monkeypatch.delitem(sys.modules, "xformers.ops", raising=False)


@pytest.mark.skipif(not HAVE_REF, reason="reference tree / staged copy not present")
def test_reference_trajectory_reproduces_the_oracle_loop():
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, "-c", SCRIPT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rel = float([l for l in r.stdout.splitlines() if l.startswith("REL")][-1].split()[1])
    assert rel < 2e-5, rel


def test_reference_subprocess_and_session_take_turns_on_the_chip(tmp_path):
    """tests/conftest.py::_GpuTurn vs oracle/gpu_reference.py::GpuTurn (two processes, no GPU needed): the subprocess's turn waits for the
    test that holds the lock, and the session does not start its next test while the subprocess has asked for the chip"""
    import subprocess
    import sys
    import time
    import conftest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time, os; sys.path.insert(0, %r)\n"
            "from oracle.gpu_reference import GpuTurn\n"
            "d = sys.argv[1]\n"
            "with GpuTurn(d):\n"
            "    open(os.path.join(d, 'ref_started'), 'w').close(); time.sleep(1.0)\n"
            "open(os.path.join(d, 'ref_done'), 'w').close()\n") % root
    old = dict(conftest._REF_PROC)
    turn = conftest._GpuTurn()
    try:
        conftest._REF_PROC.clear()
        conftest._REF_PROC.update(dir=str(tmp_path), proc=None)
        turn.acquire()                                       # "a GPU test is running"
        proc = subprocess.Popen([sys.executable, "-c", code, str(tmp_path)])
        conftest._REF_PROC["proc"] = proc
        t0 = time.time()
        while not os.path.exists(tmp_path / "ref_wants_gpu"):
            assert time.time() - t0 < 120 and proc.poll() is None
            time.sleep(0.05)
        time.sleep(0.5)
        assert not os.path.exists(tmp_path / "ref_started")  # it waits for the running test
        turn.release()
        turn.acquire()                                       # "the next test": must not start before the subprocess's turn is over
        assert os.path.exists(tmp_path / "ref_started")
        assert not os.path.exists(tmp_path / "ref_wants_gpu")
        turn.release()
        assert proc.wait(timeout=60) == 0 and os.path.exists(tmp_path / "ref_done")
    finally:
        turn.release()
        conftest._REF_PROC.clear()
        conftest._REF_PROC.update(old)
