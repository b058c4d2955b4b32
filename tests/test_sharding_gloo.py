"""N>1 path on CPU (gloo, world_size 2). Three layers: (i) the round-robin partition + all-gather plumbing on deterministic
records; (ii) grasp candidates: every rank evaluates ITS shard of a real (small) ContactGrasp scene with the CPU oracle's
refinement (stand-in for the device refinement, which needs a GPU) and the gathered result equals the single-process one;
(iii) frames: every rank tracks ITS frames (oracle optimizePose on small synthetic frames) and the gathered
{energy, T_mc, joints} records equal the single-process ones -- the shard-by-frame path of bench.py's 256-frame leg."""
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


# ---------------------------------------------------------------------------------------------------------------------------
# real work per shard: the CPU oracle stands in for the device loops
N_GRASPS = 5
N_FRAMES = 5


def _grasp_setup():
    import oracle_lib as O
    from dart_amd import _abi, configs
    sc = configs.contact_grasp_scene(N_GRASPS, n_attract=30, n_repulse=60)
    oms = [O.OModel(sc.hand, 0), O.OModel(sc.obj, 1)]
    clouds = [sc.hand.collision_cloud(2e4, seed=11), sc.obj.collision_cloud(2e4, seed=12)]
    pots = [np.asarray(sc.potential, dtype=np.int32).reshape(-1), np.ones(1, dtype=np.int32)]
    o = _abi.NoObsOptions()
    o.numIterations, o.nModels = 2, 2
    for m in range(2):
        o.regularization[m], o.regularizationScaled[m] = sc.regularization, sc.lm_damping
    o.lambdaIntersection[0], o.lambdaIntersection[1], o.lambdaIntersection[2] = sc.lambda_intra, sc.lambda_inter, sc.lambda_inter
    T, j = configs.grasp_batch_arrays(sc)
    return O, sc, oms, clouds, pots, o, T, j


def _refine_grasps(indices, setup):
    O, sc, oms, clouds, pots, o, T, j = setup
    e = np.zeros(len(indices), dtype=np.float32)
    rec = np.zeros((len(indices), 12 + j.shape[1]), dtype=np.float32)
    for k, c in enumerate(indices):
        oT, oj, oe, _, _ = O.optimize_poses_noobs(oms, [T[c, 0], T[c, 1]], [j[c], np.zeros(0, dtype=np.float32)], o, clouds, pots, sc.priors)
        e[k] = oe
        rec[k] = np.concatenate([oT[0], oj[0]])
    return e, rec


def _frame_setup():
    import oracle_lib as O
    import scenes
    from dart_amd import configs, se3
    sc = configs.config2(64, 48)
    m = sc.host_model
    om = O.OModel(m, 0, sc.obs_layout)
    sc.opts.numIterations = 2
    sc.opts.lambdaModToObs = 0.0
    o = sc.opts.c_struct(0, 32, 24, (sc.fl[0] / 2, sc.fl[1] / 2))
    T_mf, _ = om.fk(sc.true_joints)
    frames = [scenes.points_near_model(m, T_mf, se3.invert(sc.true_T_mc), 64, 48, 100 + k) for k in range(N_FRAMES)]
    return O, sc, om, o, frames


def _track_frames(indices, setup):
    O, sc, om, o, frames = setup
    e = np.zeros(len(indices), dtype=np.float32)
    rec = np.zeros((len(indices), 12 + om.n_joints), dtype=np.float32)
    for k, f in enumerate(indices):
        v = frames[f]
        T, j, info, _ = O.optimize_pose(om, sc.init_T_mc, sc.init_joints, v, np.zeros_like(v), o)
        e[k] = info[-1, 0]
        rec[k] = np.concatenate([T, j])
    return e, rec


def _real_worker(rank, world, port, q, what):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dart_amd import sharding
    n, setup, work = (N_GRASPS, _grasp_setup(), _refine_grasps) if what == "grasps" else (N_FRAMES, _frame_setup(), _track_frames)
    idx = sharding.shard_indices(n, rank, world)
    e, rec = work(idx, setup)
    ge, gp = sharding.gather_candidates(n, rank, world, torch.from_numpy(e), torch.from_numpy(rec))
    out = (ge.numpy().copy(), gp.numpy().copy()) if rank == 0 else None
    q.put((rank, out, [int(i) for i in idx]))
    dist.destroy_process_group()


def _run_real(what, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + (7 if what == "frames" else 0)
    procs = [ctx.Process(target=_real_worker, args=(r, 2, port, q, what)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        r, out, idx = q.get(timeout=300)
        res[r] = (out, idx)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == list(range(0, n, 2)) and res[1][1] == list(range(1, n, 2))
    return res[0][0]


def test_two_ranks_refine_their_grasp_shards_with_the_oracle():
    ge, gp = _run_real("grasps", N_GRASPS)
    e, rec = _refine_grasps(list(range(N_GRASPS)), _grasp_setup())
    assert np.array_equal(ge, e) and np.array_equal(gp, rec)            # sharded result == single-process result, bit for bit
    assert len(set(ge.tolist())) == N_GRASPS and np.isfinite(gp).all()


def test_two_ranks_track_their_frame_shards_with_the_oracle():
    ge, gp = _run_real("frames", N_FRAMES)
    e, rec = _track_frames(list(range(N_FRAMES)), _frame_setup())
    assert np.array_equal(ge, e) and np.array_equal(gp, rec)
    assert len({r.tobytes() for r in gp}) == N_FRAMES                   # distinct frames, distinct results
