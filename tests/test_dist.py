"""Multi-process (gloo, world_size 2) test of the batch-sharding path used at N > 1:
ranks own contiguous batch ranges, no data-path collective, optional all-gather of results."""
import os
import socket

import numpy as np
import pytest

from lol_amd.dist import shard_range, shard_sizes


def test_shard_ranges_partition_the_batch():
    for batch in (0, 1, 7, 8, 4096, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard_range(batch, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(shard_sizes(batch, world)) - min(shard_sizes(batch, world)) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _worker(rank, world, port, batch, n, T, out_dir):
    import torch
    import torch.distributed as dist
    from lol_amd.dist import allgather_batch, shard_range as sr
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = torch.arange(batch * n * T, dtype=torch.int64).reshape(batch, n, T)   # same on every rank
    lo, hi = sr(batch, rank, world)
    local = full[lo:hi].clone() * 3 + 1          # stand-in for this rank's independent batch of results
    got = allgather_batch(local, batch)
    ok = torch.equal(got, full * 3 + 1)
    # weak-scaling bookkeeping as bench.py does it: max over ranks of the local time
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = ok and float(t.item()) == float(world)
    np.save(os.path.join(out_dir, f"ok{rank}.npy"), np.array([ok]))
    dist.destroy_process_group()


@pytest.mark.parametrize("batch", [8, 7])
def test_allgather_two_ranks_gloo(tmp_path, batch):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, batch, 4, 2, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert bool(np.load(tmp_path / f"ok{r}.npy")[0])
