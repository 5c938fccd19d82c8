"""Two processes, each with its own context of the HIP renderer, shard the strata and merge their accumulators the
way bench.py does on N GPUs (pt_copy_accum_to_device -> reduce onto rank 0 -> pt_set_accum_from_device), here with
both ranks on the one GPU of the test box and the reduce over gloo on host copies.  The merged image equals the
one-context image to rounding (the two halves are summed in another association)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WIDTH, TOTAL, SEED = 64, 36, 11


def _worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pathtracer_amd import scenes
    from pathtracer_amd.core import HostScene, Renderer
    from pathtracer_amd.sharding import reduce_accumulator, stratum_ranges
    torch.cuda.set_device(0)
    r = Renderer(HostScene(scenes.small_mixed(WIDTH, TOTAL)), device=0)
    b, e = stratum_ranges(TOTAL, world)[rank]
    r.render(b, e, seed=SEED)
    acc_dev = torch.empty(r.npix * 3, dtype=torch.float32, device="cuda")
    r.copy_accum_to_device(acc_dev.data_ptr())
    acc = acc_dev.cpu()
    n = torch.tensor([e - b], dtype=torch.int64)
    reduce_accumulator(acc, dst=0)
    dist.reduce(n, dst=0, op=dist.ReduceOp.SUM)
    if rank == 0:
        acc_dev.copy_(acc)
        torch.cuda.synchronize()
        r.set_accum_from_device(acc_dev.data_ptr(), int(n.item()))
        merged, count = r.read_accum()
        np.save(out_path, np.concatenate([merged.reshape(-1), [float(count)]]))
        np.save(out_path + ".rgb.npy", r.read_rgb8())
    dist.barrier()
    r.close()
    dist.destroy_process_group()


def test_two_processes_on_one_gpu_shard_and_merge(tmp_path):
    import torch.multiprocessing as mp
    from pathtracer_amd import scenes
    from pathtracer_amd.core import HostScene, Renderer
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    assert got[-1] == TOTAL
    r = Renderer(HostScene(scenes.small_mixed(WIDTH, TOTAL)), device=0)
    r.render(0, TOTAL, seed=SEED)
    want = np.nan_to_num(r.read_accum()[0].reshape(-1))
    img_want = r.read_rgb8()
    r.close()
    merged = np.nan_to_num(got[:-1])
    assert np.abs(merged - want).max() <= 2e-5 * max(1.0, float(np.abs(want).max()))
    diff = np.abs(np.load(out + ".rgb.npy").astype(np.int32) - img_want.astype(np.int32))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3
