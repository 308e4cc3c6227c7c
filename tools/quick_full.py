# quick timing probe of a full-size (cfg2) UNet forward with random weights
import sys, time, torch
sys.path.insert(0, "/root/repo")
from followyourclick_amd.engine import UNet3DConfig
from followyourclick_amd.engine.unet3d import UNet3DEngine
from followyourclick_amd.engine.weights import pack_unet
from followyourclick_amd.engine.schema import random_state_dict, unet_schema
t0 = time.time()
cfg = UNet3DConfig()
sd = random_state_dict(unet_schema(cfg), seed=0)
print("weights", time.time() - t0)
eng = UNet3DEngine(pack_unet(sd, cfg, torch.bfloat16, "cuda:0"))
print("packed", time.time() - t0)
B, F, H, Wd = 2, 16, 64, 64
g = torch.Generator().manual_seed(0)
eng.prepare_context(torch.randn(B, 77, 768, generator=g))
_, temb = eng.prepare_time_embeddings([961], [2, 2], [4, 4], B)
x = torch.randn(B * F * H * Wd, 64, generator=g).to(torch.bfloat16).cuda()
for it in range(3):
    torch.cuda.synchronize(); t = time.time()
    out = eng.forward(x, temb, B, F, H, Wd)
    torch.cuda.synchronize()
    print(f"forward {it}: {(time.time() - t) * 1e3:.1f} ms  finite={torch.isfinite(out.float()).all().item()} std={out.float().std().item():.3f}")
print("mem GB", torch.cuda.max_memory_allocated() / 2**30)
