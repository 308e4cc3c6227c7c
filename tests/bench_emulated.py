"""tests-side wrapper of bench.py: the SAME entry point (rendezvous from the torch.distributed.run environment, barrier,
max-over-ranks timing, one JSON line from rank 0) on CPU ranks over gloo, with the op emulator (tests/emu_ops.py) at tiny widths
injected through bench.main(emulation=...).  Test infrastructure: bench.py itself knows nothing of tests/; the line this prints
says "data": "emulated" and is no measurement.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tests/bench_emulated.py --gpus 2 ...
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import bench  # noqa: E402
from emu_ops import EmuOps  # noqa: E402
from followyourclick_amd.engine import UNet3DConfig  # noqa: E402

if __name__ == "__main__":
    bench.main(emulation={"ops": EmuOps(), "cfg": UNet3DConfig(block_out_channels=(64, 128, 256, 256), cross_attention_dim=64, sample_size=8)})
