#!/usr/bin/env python3
"""bench.py -- headline benchmark of BASELINE.json: Gauss-Newton iterations/s on one 640x480 Allegro frame
(config 2: obs->mod + mod->obs), synthetic depth generated on device.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one Optimizer::optimizePose call = observation-SDF build + opts.numIterations (5, the reference default)
Gauss-Newton iterations starting from the same perturbed pose; `value` = GN iterations/s over the whole job. The
path does not split one frame across GPUs (north_star): with N ranks every rank tracks its own frame (independent
units, weak scaling) and a single RCCL all-gather collects {energy, pose} per frame at the end.

Extra objects on the JSON line:
  roofline      dominant kernel's ALGORITHMIC bytes per launch / its average launch duration, measured with HIP events
                on the launch stream in a second pass over the same K steps (events add overhead, so the throughput
                pass runs without them); peak = 8 TB/s HBM (MI355X_MICROARCH.md). traffic = PMC HBM bytes (null here;
                collected with rocprofv3 --pmc, see profiles/).
  cpu_baseline  the CPU oracle (plain C restatement, "port") timed on this host on a bounded sample: whole steps of
                the same workload, single thread.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0
PROF_NAMES = ["obs_sdf_build", "raycast_prediction", "fused_assoc_obs2mod", "ne_mod2obs", "ne_self_intersection", "gn_solve", "refine", "other"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--norm-threshold", type=float, default=None, help="override the workload's normal gate (reference default -0.1)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from dart_amd import configs, device as D
    from dart_amd._abi import check, lib, fptr, iptr
    from dart_amd.optimizer import Optimizer, synthetic_observation

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    D.init(local_rank)
    dev = torch.device("cuda", local_rank)

    # ---- workload: BASELINE config 2, one frame per rank (rank r perturbs the seed so frames differ)
    sc = configs.config2()
    if args.norm_threshold is not None:
        sc.opts.normThreshold = args.norm_threshold
    if rank:
        rng = np.random.default_rng(configs.SEED + rank)
        lim = np.asarray(sc.host_model.joint_limits, dtype=np.float64)
        sc.true_joints = np.clip(sc.true_joints + rng.normal(0, 0.05, len(lim)), lim[:, 0], lim[:, 1]).astype(np.float32)
    dm = D.DeviceModel(sc.host_model, model_id=0, obs_layout=sc.obs_layout)
    dm.set_pose(sc.true_T_mc, sc.true_joints)
    depth, vert, norm = synthetic_observation([dm], sc.width, sc.height, sc.fl, sc.pp)
    opt = Optimizer(sc.width, sc.height, sc.fl, sc.pred_width, sc.pred_height)
    iters = sc.opts.numIterations

    def step(debug=False):
        dm.set_pose(sc.init_T_mc, sc.init_joints)
        opt.optimizePose(dm, vert, norm, sc.opts, debug=debug)

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- result gather (the path's only exchange): {energy, T_mc[12], joints} per frame
    step(debug=True)
    T_mc, _, _, joints = dm.get_transforms()
    info = opt.iter_info.cpu().numpy()
    rec = torch.tensor(np.concatenate([[info[-1, 0] + info[-1, 1]], T_mc, joints]).astype(np.float32), device=dev)
    if distributed:
        out = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(out, rec)

    # ---- roofline leg (rank 0): per-kernel durations from HIP events over the same K steps
    roofline = None
    cpu_baseline = None
    if rank == 0:
        check(lib().dartk_profile_begin())
        for _ in range(args.steps):
            step()
        ms = np.zeros(8, dtype=np.float32)
        cnt = np.zeros(8, dtype=np.int32)
        check(lib().dartk_profile_end(fptr(ms), iptr(cnt)))
        per_launch_us = {PROF_NAMES[i]: (1000.0 * ms[i] / cnt[i] if cnt[i] else 0.0) for i in range(8)}
        per_step_us = {PROF_NAMES[i]: 1000.0 * ms[i] / args.steps for i in range(8)}
        dom = int(np.argmax(ms))

        # algorithmic bytes from exact counts of one step of the CPU oracle on this workload
        import oracle_lib as O
        om = O.OModel(sc.host_model, 0, sc.obs_layout)
        o_struct = sc.opts.c_struct(0, opt.pred_width, opt.pred_height, opt.pred_fl)
        v_np, n_np = vert.cpu().numpy(), norm.cpu().numpy()
        O.lib().orc_stats_reset()
        O.lib().orc_stats_enable(1)
        tc0 = time.perf_counter()
        _, _, o_info, _ = O.optimize_pose(om, sc.init_T_mc, sc.init_joints, v_np, n_np, o_struct)
        cpu_one = time.perf_counter() - tc0
        O.lib().orc_stats_enable(0)
        st = (C.c_longlong * 8)()
        O.lib().orc_stats_get(st)
        A = sc.width * sc.height
        P = opt.pred_width * opt.pred_height
        S = 22 + 22 * 23 // 2 + 1
        n_assoc = float(o_info[:, 3].mean())
        n_pred = float(o_info[:, 4].mean())
        bytes_per_launch = {
            # 16 B vertex per pixel + 32 B per trilinear lookup of the association scan + (16 B normal + 192 B gradient)
            # per associated candidate + one partial system per workgroup
            "fused_assoc_obs2mod": 16.0 * A + 32.0 * st[O_ASSOC] / iters + (16.0 + 32.0) * st[O_ASSOC_GRAD] / 6 / iters * 1.0
                                   + 32.0 * st[O_ASSOC_GRAD] / iters + 4.0 * S * 256,
            # 32 B per marching-step lookup + 16 B prediction write per pixel
            "raycast_prediction": 32.0 * st[O_RAYCAST] / iters + 16.0 * P,
            # 16 B per prediction pixel + 32 B per lookup (value + gradient) + partial systems
            "ne_mod2obs": 16.0 * P + 32.0 * st[O_M2O] / iters + 4.0 * S * 256,
            "gn_solve": 4.0 * S * 256 * 2,
            "obs_sdf_build": 4.0 * 64 ** 3 * 9 + 16.0 * 64 ** 3,
        }
        name = PROF_NAMES[dom]
        ach = bytes_per_launch.get(name, 0.0) / (per_launch_us[name] * 1e-6) / 1e9 if per_launch_us[name] else 0.0
        roofline = {"bound": "hbm", "kernel": name, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None,
                    "algorithmic_bytes_per_launch": round(bytes_per_launch.get(name, 0.0)),
                    "avg_launch_us": round(per_launch_us[name], 2),
                    "per_step_us": {k: round(v, 2) for k, v in per_step_us.items() if v},
                    "counts": {"associated_per_iter": n_assoc, "predicted_per_iter": n_pred,
                               "assoc_lookups_per_iter": st[O_ASSOC] / iters, "raycast_lookups_per_iter": st[O_RAYCAST] / iters}}
        if not args.no_cpu_baseline:
            tc0 = time.perf_counter()
            for _ in range(args.cpu_steps):
                O.optimize_pose(om, sc.init_T_mc, sc.init_joints, v_np, n_np, o_struct)
            cpu_t = time.perf_counter() - tc0
            cpu_baseline = {"value": round(args.cpu_steps * iters / cpu_t, 3), "unit": "GN iterations/s", "cores": 1, "kind": "port",
                            "sample": "%d whole steps (obs-SDF build + %d iterations) of the same 640x480 Allegro workload, CPU oracle, 1 thread"
                                      % (args.cpu_steps, iters)}

    if rank == 0:
        value = world * args.steps * iters / elapsed
        line = {
            "metric": "Gauss-Newton iters/sec (640x480 Allegro)", "value": round(value, 2), "unit": "GN iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000.0 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: Allegro 22-DOF (17 SDFs, proxy geometry), synthetic 640x480 depth, "
                                   "prediction 320x240, 64^3 obs SDF, obs->mod + mod->obs, %d GN iterations per optimizePose call, "
                                   "normThreshold %.1f; one frame per GPU" % (iters, sc.opts.normThreshold),
                       "iterations_per_step": iters, "frames_per_gpu": 1, "parallelism": "independent frames, 1 per GPU"},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line, default=float))
    if distributed:
        dist.destroy_process_group()


O_ASSOC, O_ASSOC_GRAD, O_NE_O2M, O_RAYCAST, O_M2O = 1, 2, 3, 4, 5

if __name__ == "__main__":
    main()
