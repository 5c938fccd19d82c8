"""N > 1: stratum sharding plus one reduce gives the single-process image (gloo, world size 2, CPU).

The renderer used here is the CPU oracle (tests may use it); what is under test is the host logic
that bench.py runs on GPUs: the split of the stratum list and the reduce of the accumulators."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pathtracer_amd.sharding import reduce_accumulator, stratum_ranges


def test_stratum_ranges_cover_everything_once():
    for total, world in [(10201, 8), (64, 1), (64, 3), (5, 8), (0, 2)]:
        r = stratum_ranges(total, world)
        assert len(r) == world and r[0][0] == 0 and r[-1][1] == total
        assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
        sizes = [e - b for b, e in r]
        assert max(sizes) - min(sizes) <= 1
    assert stratum_ranges(10201, 8)[0] == (0, 1276) and stratum_ranges(10201, 8)[7] == (8926, 10201)
    with pytest.raises(ValueError):
        stratum_ranges(4, 0)


def _worker(rank, world, port, total, seed, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.pyoracle import OracleScene
    from pathtracer_amd import scenes
    o = OracleScene(scenes.cornell(32, total))
    b, e = stratum_ranges(total, world)[rank]
    acc = torch.from_numpy(o.render(b, e, seed=seed, threads=2)["accum"].reshape(-1).copy())
    n = torch.tensor([e - b], dtype=torch.int64)
    reduce_accumulator(acc, dst=0)
    dist.reduce(n, dst=0, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.save(out_path, np.concatenate([acc.numpy(), [float(n.item())]]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one(tmp_path):
    from oracle.pyoracle import OracleScene
    from pathtracer_amd import scenes
    total, seed = 16, 21
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(2, port, total, seed, out), nprocs=2, join=True)
    got = np.load(out)
    assert got[-1] == total
    ref = OracleScene(scenes.cornell(32, total)).render(0, total, seed=seed, threads=2)["accum"].reshape(-1)
    # the two halves are summed in a different association than the single pass loop: tolerance, not bits
    assert np.allclose(got[:-1], ref, rtol=1e-5, atol=1e-5)
    assert np.abs(got[:-1] - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
