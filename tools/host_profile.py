"""Where does the HOST spend a DDIM step?  (bench.py reports host_launch_ms_per_ddim_step close to the GPU's time per step.)
cProfile over a few steps of the bench workload + the same steps with the launches themselves stubbed out (pure Python cost)."""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench as B
from followyourclick_amd.engine import DDIMConfig, UNet3DConfig
from followyourclick_amd.engine.sampler import DDIMSampler
from followyourclick_amd.engine.schema import random_state_dict, unet_schema
from followyourclick_amd.engine.unet3d import UNet3DEngine
from followyourclick_amd.engine.weights import pack_unet

device = torch.device("cuda", 0)
cfg = UNet3DConfig()
sd = random_state_dict(unet_schema(cfg), seed=0, materialize=True)
eng = UNet3DEngine(pack_unet(sd, cfg, torch.bfloat16, device))
sampler = DDIMSampler(eng, DDIMConfig())
c = B.synthetic_inputs(cfg, 16, 64, 64, 1000, device)
steps = 6
run = lambda: sampler.sample(c["latents"], c["text"], steps, 8.0, c["first"], c["mask"], fps=[2], flow=[4])
run(); torch.cuda.synchronize()
t = time.time(); run(); th = time.time() - t; torch.cuda.synchronize(); tg = time.time() - t
print(f"eager: host returns after {1000 * th / steps:.1f} ms/step, GPU done after {1000 * tg / steps:.1f} ms/step")
pr = cProfile.Profile(); pr.enable(); run(); pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
# launches stubbed: what is left is Python + ctypes marshalling + torch.empty
lib = eng.ops.lib if hasattr(eng.ops, "lib") else None
import followyourclick_amd.ops as O
calls = [0]
orig = O.HipOps._call
def fake(self, name, args):
    calls[0] += 1
O.HipOps._call = fake
t = time.time(); run(); tp = time.time() - t
O.HipOps._call = orig
torch.cuda.synchronize()
print(f"launches stubbed out: {1000 * tp / steps:.1f} ms/step of Python for {calls[0] / steps:.0f} fyc_* launches/step (torch's own kernels still launch)")
