"""N>1 path on CPU: two gloo ranks shard a candidate list round-robin, each evaluates its shard with the CPU oracle's
getError (stand-in for the device refinement, which needs a GPU), and one all_gather rebuilds the full result."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

N_CAND = 7      # odd on purpose: ragged shards


def _energies(indices):
    """deterministic per-candidate 'energy' and 'pose' records"""
    rng = np.random.default_rng(1234)
    e = rng.normal(size=N_CAND).astype(np.float32)
    p = rng.normal(size=(N_CAND, 5)).astype(np.float32)
    return e[indices], p[indices], e, p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dart_amd import sharding
    idx = sharding.shard_indices(N_CAND, rank, world)
    e, p, full_e, full_p = _energies(idx)
    ge, gp = sharding.gather_candidates(N_CAND, rank, world, torch.from_numpy(e), torch.from_numpy(p))
    ok = np.array_equal(ge.numpy(), full_e) and np.array_equal(gp.numpy(), full_p)
    best = sharding.best_candidates(ge, 2)
    ok = ok and np.array_equal(best, np.argsort(full_e, kind="stable")[:2])
    # timing reduction used by bench.py: max over ranks
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = ok and float(t.item()) == world
    q.put((rank, bool(ok), len(idx)))
    dist.destroy_process_group()


def test_round_robin_shards_and_all_gather_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert sorted(n for _, _, n in res) == [3, 4]


def test_shard_helpers_single_rank():
    from dart_amd import sharding
    assert list(sharding.shard_indices(10, 1, 4)) == [1, 5, 9]
    assert sharding.shard_size(4096, 8) == 512 and sharding.shard_size(7, 2) == 4
    e = torch.tensor([3.0, 1.0, 2.0])
    p = torch.zeros((3, 2))
    ge, gp = sharding.gather_candidates(3, 0, 1, e, p)
    assert ge is e and gp is p
    assert list(sharding.best_candidates(e, 2)) == [1, 2]
    # every candidate belongs to exactly one rank
    for world in (1, 2, 3, 8):
        allidx = np.concatenate([sharding.shard_indices(4096, r, world) for r in range(world)])
        assert np.array_equal(np.sort(allidx), np.arange(4096))
