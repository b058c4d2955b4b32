"""Multi-GPU sharding of independent problems (frames, grasp candidates): one process per GPU, static round-robin
partition by candidate index, NO collective inside an iteration; a single all-gather of {energy, pose} per candidate
at the end (RCCL over xGMI with backend "nccl", gloo on CPU). The reference has no multi-device path."""
import numpy as np
import torch
import torch.distributed as dist


def shard_indices(n, rank, world):
    """Round-robin: candidate i belongs to rank i % world."""
    return np.arange(rank, n, world)


def shard_size(n, world):
    return (n + world - 1) // world


def gather_candidates(n, rank, world, energy, poses):
    """energy [n_local], poses [n_local, P] (tensors on this rank's device) -> full (energy [n], poses [n, P]) on every
    rank, in candidate order. One all_gather of a [shard, 1+P] record block; shards are padded to equal length."""
    if world == 1:
        return energy, poses
    per = shard_size(n, world)
    P = poses.shape[1]
    rec = torch.full((per, 1 + P), float("nan"), dtype=torch.float32, device=energy.device)
    k = energy.shape[0]
    rec[:k, 0] = energy
    rec[:k, 1:] = poses
    if dist.get_backend() == "gloo" and rec.is_cuda:       # gloo has no all_gather for device tensors (CPU tests, rehearsals)
        host = rec.cpu()
        out = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(out, host)
        out = [o.to(rec.device) for o in out]
    else:
        out = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(out, rec)
    full_e = torch.empty(n, dtype=torch.float32, device=energy.device)
    full_p = torch.empty((n, P), dtype=torch.float32, device=energy.device)
    for r in range(world):
        idx = torch.as_tensor(shard_indices(n, r, world), device=energy.device)
        full_e[idx] = out[r][:len(idx), 0]
        full_p[idx] = out[r][:len(idx), 1:]
    return full_e, full_p


def best_candidates(energy, k=1):
    """grasp_analyzer sorts its results by energy (grasp_analyzer.cpp:814-849): indices of the k lowest energies."""
    e = energy.detach().cpu().numpy()
    return np.argsort(e, kind="stable")[:k]
