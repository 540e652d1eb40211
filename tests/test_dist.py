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


def _run_bench(*argv, env_extra=None, timeout=240):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE: bench.py itself starts two ranks, they rendezvous
    (gloo here), run the barrier / MAX-over-ranks protocol and rank 0 reports n_gpus = 2."""
    import json
    r = _run_bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "16", "--launcher-selftest")
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                       # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size"] == 2 and out["backend"] == "gloo"
    assert out["items_all_ranks"] == 2 * 16 * 3  # both ranks did their own shard, every step
    assert "NOT a measurement" in out["metric"]


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    r = _run_bench("--gpus", "2", "--launcher-selftest", env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_bench_fails_when_any_rank_fails():
    """No GPU here: every real rank exits non-zero, and so must the launcher (no JSON line)."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present: the real ranks would run")
    r = _run_bench("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "ranks failed" in r.stderr
