"""One rank, RCCL backend: the accumulator leaves the library through pt_copy_accum_to_device, goes through
dist.reduce and comes back through pt_set_accum_from_device, as bench.py does it with N ranks."""
import os
import sys
import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import scenes
from pathtracer_amd.core import HostScene, Renderer

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
r = Renderer(HostScene(scenes.cornell(64, 16)))
r.render(0, 8, seed=1)
before, n = r.read_accum()
buf = torch.empty(r.npix * 3, dtype=torch.float32, device="cuda")
r.copy_accum_to_device(buf.data_ptr())
dist.reduce(buf, dst=0, op=dist.ReduceOp.SUM)
dist.barrier()
torch.cuda.synchronize()
r.set_accum_from_device(buf.data_ptr(), n)
after, n2 = r.read_accum()
assert n2 == n and np.array_equal(before.view(np.uint32), after.view(np.uint32))
print("ok: reduce over RCCL with one rank leaves the accumulator as it was;", n, "passes, mean", float(after.mean()))
dist.destroy_process_group()
