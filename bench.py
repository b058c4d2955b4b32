#!/usr/bin/env python3
"""bench.py -- headline benchmark of BASELINE.json: Gauss-Newton iterations/s on one 640x480 Allegro frame
(config 2: obs->mod + mod->obs, synthetic depth generated on device) + batched ContactGrasp refinement (config 4).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Headline (`metric`/`value`): a "step" is one Optimizer::optimizePose call = observation-SDF build + opts.numIterations
(5, the reference default) Gauss-Newton iterations from the same perturbed pose; `value` = GN iterations/s over the
whole job. One frame is never split across GPUs (north_star): with N ranks every rank tracks its own frame
(independent units, weak scaling) and one RCCL all-gather collects {energy, pose} per frame at the end.

Extra objects on the same JSON line:
  roofline        the dominant kernel of the headline path: ALGORITHMIC bytes per launch / its average launch duration,
                  measured with HIP events on the launch stream in a second pass over the same K steps (events add
                  overhead, so the throughput pass runs without them); peak = 8 TB/s HBM (MI355X_MICROARCH.md). The
                  algorithmic bytes price the work the kernel PERFORMS: its trilinear lookups are counted on the device by the
                  diagnostic twin of the library (libdartk_count.so, dartk_lookup_counters) on the same inputs -- not the
                  lookups the reference's un-pruned loops would make. `traffic` = PMC HBM bytes per launch (profiles/),
                  `frac_hbm` = traffic / duration / peak, `l2_gather_GBs` = 32 B x lookups / duration.
  cpu_baseline    the CPU oracle (plain C restatement, "port") timed on this host on a bounded sample with all of this
                  process's cores (OpenMP / threads; `cores` says how many) and, nested, with one thread; CPU model printed.
  repeats         the timed K-step loop repeated (default 24 more times): median / min / max ms per step.
  batched_grasps  BASELINE config 4: 4096 HumanHand grasp candidates x 40 iterations of OptimizerNoObs (contact priors
                  + intersection terms), candidates sharded round-robin over the N ranks (strong scaling for this leg),
                  one all-gather of {energy, pose}; grasps/s, its own roofline and CPU baseline.
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
TRACE_BYTES_PER_PAIR = 8        # ray-cast trace record per traced (prediction pixel, SDF) pair: {detection t, hit t}
PROF_NAMES = ["obs_sdf_build", "raycast_trace_kernel", "fused_obs2mod_kernel", "ne_mod2obs_kernel", "ne_self_intersection_kernel",
              "gn_solve_single_kernel", "raycast_replay_kernel", "valid_pixel_compaction", "gn_reduce_kernel",
              "noobs_terms_kernel<intersection>", "noobs_solve_kernel", "noobs_terms_kernel<priors>", "assoc_multi_kernel"]
O_ASSOC, O_ASSOC_GRAD, O_NE_O2M, O_RAYCAST, O_M2O, O_INTER, O_PRIOR = 1, 2, 3, 4, 5, 6, 7
N_CLASSES = 13


def _profile_json(name):
    path = os.path.join(ROOT, "profiles", name)
    if os.path.exists(path):
        with open(path) as f:
            return json.load(f)
    return {}


# What the committed rocprofv3 passes of THIS bench command measured (profiles/README.md says which pass wrote which file);
# the live run prices its own HIP-event durations and prints these beside them. Empty = not collected.
PMC_TRAFFIC = _profile_json("pmc_traffic.json")                 # HBM bytes per launch (FETCH_SIZE / WRITE_SIZE passes, corrected)
PMC_TRAFFIC_BATCH = _profile_json("pmc_traffic_batch.json")     # the same for the 64-frame launches of the config-5 batch
ROCPROF = _profile_json("kernel_durations.json")                # --kernel-trace: calls / median / average duration per kernel
ROCPROF_POSE_BATCH = _profile_json("kernel_durations_pose_batch.json")   # the same of a pass over the single-model frame batch alone
PMC_TRAFFIC_POSE_BATCH = _profile_json("pmc_traffic_pose_batch.json")
L2_RATES = _profile_json("l2_hit_rate.json")                    # TCC_HIT / MISS / REQ pass
LIMITERS = _profile_json("kernel_limiters.json")                # what the counters say bounds each kernel (+ the evidence)
# bench kernel class -> (kernel name in the rocprof tables, regular expression on its template arguments or None)
ROCPROF_KERNELS = {
    "trace_and_obs2mod_kernel": [("trace_and_obs2mod_kernel", None)],
    "fused_obs2mod_kernel": [("fused_obs2mod_kernel", None), ("ne_obs2mod_kernel", None)],
    "raycast_trace_kernel": [("raycast_trace_kernel", None)],
    "ne_mod2obs_kernel": [("ne_mod2obs_kernel", r"(true$)|(@batch)")],      # single frame: the instantiation with the fused replay (config 2)
    "gn_reduce_kernel": [("gn_reduce_kernel", None)],
    "gn_solve_single_kernel": [("gn_solve_single_kernel", None)],
    "noobs_terms_kernel<priors>": [("noobs_terms_kernel", r"^\d+, [234]")],
    "noobs_terms_kernel<intersection>": [("noobs_inter_union_kernel", None), ("noobs_terms_kernel", r"^\d+, [01]")],
    "noobs_solve_kernel": [("noobs_solve_kernel", None)],
    "assoc_multi_kernel": [("assoc_multi_kernel", None)],
    "raycast_replay_kernel": [("raycast_replay_multi_rows_kernel", None), ("raycast_replay_multi_kernel", None), ("raycast_replay_kernel", None)],
    "ne_self_intersection_kernel": [("ne_self_intersection_kernel", None), ("ne_intersection_kernel", None)],
    "multi_solve_kernel": [("multi_solve_kernel", None)],
}


def rocprof_stats(kernel, batch, retry=True, table=None):
    """calls / median / average launch duration of `kernel` (a ROCPROF_KERNELS class) in the committed rocprofv3 kernel trace;
    batch: the launches with more than one grid row (frames / candidates) or the single-row ones. None when not collected."""
    import re
    tot_calls, avg_sum, meds = 0, 0.0, []
    for base, targs in ROCPROF_KERNELS.get(kernel, [(kernel, None)]):
        for key, v in (ROCPROF if table is None else table).items():
            name, is_batch = (key[:-6], True) if key.endswith("@batch") else (key, False)
            if is_batch != batch:
                continue
            b, _, rest = name.partition("<")
            if b != base or (targs and not re.search(targs, rest.rstrip(">") + ("@batch" if is_batch else ""))):
                continue
            tot_calls += v["calls"]
            avg_sum += v["avg_us"] * v["calls"]
            meds.append((v["median_us"], v["calls"]))
    if not tot_calls:
        # kernels whose second grid dimension is not a frame (gn_reduce_kernel: term; noobs_solve_kernel: candidates in x)
        return rocprof_stats(kernel, not batch, False, table) if retry else None
    return {"calls": tot_calls, "avg_us": round(avg_sum / tot_calls, 2),
            "median_us": round(sum(m * c for m, c in meds) / tot_calls, 2),
            "source": "profiles/kernel_durations.json" if table is None else "profiles/kernel_durations_pose_batch.json"}


def l2_hit_rate(kernel):
    """launch-weighted L2 (TCC) hit rate of the kernel class in the committed counter pass (all its instantiations)"""
    import re
    hit, n = 0.0, 0
    for base, targs in ROCPROF_KERNELS.get(kernel, [(kernel, None)]):
        for key, v in L2_RATES.items():
            b, _, rest = key.partition("<")
            if b == base and (not targs or "@batch" in targs or re.search(targs, rest.rstrip(">"))):
                hit += v["hit_rate"] * v["launches"]
                n += v["launches"]
    return round(hit / n, 4) if n else None


def pmc_traffic(kernel, table):
    """HBM bytes per launch of the kernel class in a pmc_traffic*.json table (keyed by kernel name without template arguments)"""
    if kernel in table:
        return table[kernel]
    for base, _ in ROCPROF_KERNELS.get(kernel, []):
        if base in table:
            return table[base]
    return None


def prof_collect(lib, fptr, iptr, check, fn, reps):
    """HIP-event durations per kernel class over `reps` calls of fn: (us per launch, us per call, launches per call)"""
    check(lib.dartk_profile_begin())
    for _ in range(reps):
        fn()
    ms = np.zeros(N_CLASSES, dtype=np.float32)
    cnt = np.zeros(N_CLASSES, dtype=np.int32)
    check(lib.dartk_profile_end(fptr(ms), iptr(cnt)))
    per_launch_us = {PROF_NAMES[i]: float(1000.0 * ms[i] / cnt[i]) for i in range(N_CLASSES) if cnt[i]}
    per_step_us = {PROF_NAMES[i]: float(1000.0 * ms[i] / reps) for i in range(N_CLASSES) if cnt[i]}
    launches = {PROF_NAMES[i]: float(cnt[i]) / reps for i in range(N_CLASSES) if cnt[i]}
    return per_launch_us, per_step_us, launches


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_threads(args):
    if args.cpu_threads > 0:
        return args.cpu_threads
    env = os.environ.get("OMP_NUM_THREADS", "")
    return int(env) if env.isdigit() and int(env) > 0 else min(os.cpu_count() or 1, 16)


def count_lookups(fn):
    """Run fn() through libdartk_count.so (the diagnostic build with one counter per trilinear lookup) and return the
    counts by category [other, assoc, assoc gradient, obs->mod gradient, ray-cast, mod->obs, intersection, priors]."""
    from dart_amd import _abi
    cnt = (C.c_uint64 * 8)()
    with _abi.counting_library() as L:
        _abi.check(L.dartk_lookup_counters(cnt, 1))
        fn()
        _abi.check(L.dartk_lookup_counters(cnt, 1))
    return [int(v) for v in cnt]


def roofline_entry(name, bytes_per_launch, lookups_per_launch, per_launch_us, batch=False, traffic_table=None, rocprof_table=None,
                   limiter_suffix="@batch"):
    """One kernel's roofline figures. achieved = algorithmic bytes of the work performed / the launch duration measured live with
    HIP events on the launch stream. `bound` names the roof the fraction is priced against (HBM bandwidth: every kernel of this
    path moves bytes, none is a GEMM); `limiter` says what the committed counter passes show actually bounds the kernel
    (latency of a dependent-load chain, instruction issue, or HBM), with the evidence. `rocprof`: the same kernel's duration in
    the committed rocprofv3 --kernel-trace pass of this command (two event records per launch make the live figure 2-4 us
    longer); `frac_rocprof` prices the same bytes with that median."""
    us = per_launch_us[name]
    ach = bytes_per_launch / (us * 1e-6) / 1e9
    traffic = pmc_traffic(name, PMC_TRAFFIC if traffic_table is None else traffic_table)
    lim = LIMITERS.get(name + (limiter_suffix if batch and (name + limiter_suffix) in LIMITERS else ""), {})
    rp = rocprof_stats(name, batch, True, rocprof_table)
    e = {"bound": "hbm", "limiter": lim.get("limiter"), "limiter_evidence": lim.get("evidence"), "kernel": name,
         "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
         "frac_hbm": round(traffic / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5) if traffic else None,
         "l2_gather_GBs": round(32.0 * lookups_per_launch / (us * 1e-6) / 1e9, 2), "l2_hit_rate": l2_hit_rate(name),
         "algorithmic_bytes_per_launch": round(bytes_per_launch), "lookups_per_launch": round(lookups_per_launch),
         "avg_launch_us": round(us, 2), "rocprof": rp,
         "frac_rocprof": round(bytes_per_launch / (rp["median_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 5) if rp else None}
    assert e["frac"] <= 1.0, "a roofline fraction above 1 means the priced work is not the work performed: %r" % e
    return e


def per_step_from_rocprof(launches, batch, chain, step_seconds, table=None):
    """us per step per kernel class = committed rocprofv3 median x launches per step (live launch counts); `chain` lists the classes
    that run back to back on the caller's stream: their sum must fit into the measured step."""
    out = {}
    for k, n in launches.items():
        rp = rocprof_stats(k, batch, True, table)
        if rp:
            out[k] = round(rp["median_us"] * n, 2)
    if not out:
        return None
    s = sum(out.get(k, 0.0) for k in chain)
    return {"source": "rocprofv3 --kernel-trace medians (profiles/kernel_durations.json) x live launch counts", "us": out,
            "critical_chain": list(chain), "critical_chain_sum_us": round(s, 2), "step_us": round(1e6 * step_seconds, 2),
            "chain_fits_step": bool(s <= 1e6 * step_seconds)}


def tr_slab_floats(tr):
    """floats of partial systems one candidate writes per pass of the term kernels (for the algorithmic-bytes figure):
    (intersection terms, contact-prior term)."""
    D_ = sum(m.dims for m in tr.models)

    def S(d):
        return d + d * (d + 1) // 2 + 1
    total = 0
    mb = tr.max_batch       # noobs_blocks_per_term (dartk_noobs.hip)
    want = 1 << 30 if mb < 64 else 16 if mb < 128 else 8 if mb < 512 else 4 if mb < 2048 else 2
    n = len(tr.models)
    lam = tr.opts.lambdaIntersection
    for m in range(n):
        tiles = (tr.clouds[m].shape[0] + 127) // 128
        blocks = min(tiles, want)
        if lam[m + m * n] > 0 and tr.models[m].dims > 6:
            total += blocks * S(tr.models[m].dims - 6)
        for d in range(n):
            if d != m and lam[m + d * n] > 0:
                total += blocks * S(tr.models[m].dims)
    tiles = (len(tr.priors) + 127) // 128
    return total, min(tiles, want) * S(D_)


def spawn_ranks(args):
    """Launch `--gpus N` ranks of this script on this node (one process per GPU, RCCL between them) and return the launcher's
    exit code. Nothing here touches the GPU: torch.cuda.device_count() does not initialise it."""
    import socket
    import subprocess
    import torch
    n_dev = torch.cuda.device_count()
    if not args.rehearse_on_one_gpu and n_dev < args.gpus:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible (--rehearse-on-one-gpu runs every rank on cuda:0 over gloo: "
                         "a test of the sharding code, not a measurement)\n" % (args.gpus, n_dev))
        return 2
    with socket.socket() as sk:      # a free port for the rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--repeats", type=int, default=24, help="further repeats of the timed K-step loop for the median / min / max line")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the all-cores CPU baseline (0 = OMP_NUM_THREADS or min(cpu_count, 16))")
    ap.add_argument("--cpu-steps", type=int, default=12, help="single-thread CPU sample: whole steps of the headline workload (about 5 s)")
    ap.add_argument("--norm-threshold", type=float, default=None, help="override the workload's normal gate (reference default -0.1)")
    ap.add_argument("--grasp-candidates", type=int, default=4096)
    ap.add_argument("--grasp-steps", type=int, default=3)
    ap.add_argument("--no-grasps", action="store_true")
    ap.add_argument("--no-multi-model", action="store_true", help="skip the config-5 (two-model tracking) leg")
    ap.add_argument("--multi-steps", type=int, default=2)
    ap.add_argument("--frames", type=int, default=256, help="frames of the config-5 batch (sharded by index over the ranks)")
    ap.add_argument("--no-tracking", action="store_true", help="profiling aid: skip the headline leg's timed loop")
    ap.add_argument("--no-pose-batch", action="store_true", help="skip the single-model frame-batch leg (dartk_optimize_pose_batch)")
    ap.add_argument("--pose-frames", type=int, default=256, help="frames of the single-model batch (sharded by index over the ranks)")
    ap.add_argument("--pose-steps", type=int, default=3)
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="test aid for a 1-GPU box: every rank uses cuda:0 and the ranks talk over gloo (RCCL refuses two ranks on one device); "
                         "exercises the sharding / gather / max-over-ranks code of the multi-GPU launch, the numbers mean nothing")
    args = ap.parse_args()

    # `python bench.py --gpus N` (N > 1) without a launcher: start N ranks of this script under torch.distributed.run -- a
    # child process, started before anything in this one touches the GPU -- and leave with its exit code. Rank 0 of the
    # children prints the JSON line. Under a launcher (WORLD_SIZE set, the driver's form) this process IS a rank.
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    import torch
    import torch.distributed as dist
    from dart_amd import configs, device as D, sharding
    from dart_amd._abi import check, lib, fptr, iptr
    from dart_amd.optimizer import Optimizer, synthetic_observation

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); start it as `python bench.py --gpus N` or "
                 "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (args.gpus, world))
    n_dev = torch.cuda.device_count()       # (counting devices does not initialise the GPU)
    if not args.rehearse_on_one_gpu and n_dev < world:
        sys.exit("bench.py: %d rank(s) need %d GPUs, %d visible (--rehearse-on-one-gpu puts every rank on cuda:0 over gloo to "
                 "exercise the sharding code on a 1-GPU box)" % (world, world, n_dev))
    distributed = world > 1
    if args.rehearse_on_one_gpu:
        local_rank = 0
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.rehearse_on_one_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    D.init(local_rank)
    dev = torch.device("cuda", local_rank)

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not distributed:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ================================================================== headline: BASELINE config 2, one frame per rank
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

    n_steps = 0 if args.no_tracking else args.steps
    for _ in range(0 if args.no_tracking else args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        step()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)

    # result gather (the path's only exchange): {energy, T_mc[12], joints} per frame
    step(debug=True)
    T_mc, _, _, joints = dm.get_transforms()
    info = opt.iter_info.cpu().numpy()
    rec = torch.tensor(np.concatenate([[info[-1, 0] + info[-1, 1]], T_mc, joints]).astype(np.float32), device=dev)
    if distributed:
        if args.rehearse_on_one_gpu:
            rec = rec.cpu()
        out = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(out, rec)

    # repeats of the timed loop (same K steps): spread of the measurement, local to this rank
    rep = []
    for _ in range(0 if args.no_tracking else args.repeats):
        torch.cuda.synchronize()
        tr0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        torch.cuda.synchronize()
        rep.append(1000.0 * (time.perf_counter() - tr0) / max(n_steps, 1))
    repeats = None
    if rep:
        rep_all = sorted(rep + [1000.0 * elapsed / max(n_steps, 1)])
        repeats = {"n": len(rep_all), "ms_per_step_median": round(float(np.median(rep_all)), 4), "ms_per_step_min": round(rep_all[0], 4),
                   "ms_per_step_max": round(rep_all[-1], 4), "value_median": round(world * iters / (float(np.median(rep_all)) * 1e-3), 1)}

    # the same workload at the reference's default normal gate (normThreshold -0.1, optimization.h:44): with the reference's own
    # normals and nPred = -gradient the gate drops most camera-facing points (DESIGN.md section 4 iii), so it does LESS work than
    # the headline (gate off); reported beside it
    value_default_gate = None
    if not args.no_tracking and args.norm_threshold is None:
        keep_thr = sc.opts.normThreshold
        sc.opts.normThreshold = -0.1
        for _ in range(args.warmup):
            step()
        barrier()
        tg0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        barrier()
        g_el = max_over_ranks(time.perf_counter() - tg0)
        step(debug=True)
        value_default_gate = {"value": round(world * n_steps * iters / g_el, 2), "normThreshold": -0.1, "ms_per_step": round(1000.0 * g_el / max(n_steps, 1), 4),
                              "associated_per_iter": float(opt.iter_info.cpu().numpy()[:, 3].mean())}
        sc.opts.normThreshold = keep_thr

    roofline = None
    cpu_baseline = None
    if rank == 0 and not args.no_tracking:
        import oracle_lib as O
        per_launch_us, per_step_us, launches = prof_collect(lib(), fptr, iptr, check, step, args.steps)
        # lookups the kernels perform in one step (device counters of the diagnostic library, same inputs, same poses)
        cnt = count_lookups(lambda: step(debug=True))
        om = O.OModel(sc.host_model, 0, sc.obs_layout)
        o_struct = sc.opts.c_struct(0, opt.pred_width, opt.pred_height, opt.pred_fl)
        v_np, n_np = vert.cpu().numpy(), norm.cpu().numpy()
        info_np = opt.iter_info.cpu().numpy()
        P = opt.pred_width * opt.pred_height
        S = 22 + 22 * 23 // 2 + 1
        n_valid = int((v_np[..., 3] > 0).sum())
        n_assoc = float(info_np[:, 3].mean())
        n_pred = float(info_np[:, 4].mean())
        n_sdfs = len(sc.host_model.sdfs)
        lk = {"fused_obs2mod_kernel": (cnt[1] + cnt[2]) / iters, "raycast_trace_kernel": cnt[4] / iters, "ne_mod2obs_kernel": cnt[5] / iters}
        trace_bytes = float(TRACE_BYTES_PER_PAIR) * cnt[0] / iters      # records written: the (pixel, SDF) pairs inside the SDFs' screen rectangles
        bytes_per_launch = {
            # 4 B index + 16 B vertex per valid pixel + 32 B per trilinear lookup performed (association scan of the SDFs
            # whose keep box holds the point + the 6-lookup gradient of the associated ones) + 16 B normal per candidate
            # + one partial system per workgroup
            "fused_obs2mod_kernel": 20.0 * n_valid + 32.0 * lk["fused_obs2mod_kernel"] + 16.0 * cnt[2] / 6 / iters
                                    + 4.0 * S * ((n_valid + 127) // 128),
            # 32 B per marching-step lookup performed + the trace records written for the replay
            "raycast_trace_kernel": 32.0 * lk["raycast_trace_kernel"] + trace_bytes,
            # (the replay of the traces is part of this kernel) trace records read + prediction read and
            # write + 32 B per lookup (value + gradient) + one partial system per 128-pixel tile
            "ne_mod2obs_kernel": trace_bytes + 32.0 * P + 32.0 * lk["ne_mod2obs_kernel"] + 4.0 * S * min((P + 127) // 128, 1024),
            "gn_reduce_kernel": 4.0 * S * (min((P + 127) // 128, 1024) + (n_valid + 127) // 128) + 8.0 * S,
            "gn_solve_single_kernel": 4.0 * 2 * S + 96.0 * 17 * 2,
            "obs_sdf_build": 4.0 * 64 ** 3 * 9 + 16.0 * 64 ** 3,
        }
        if "fused_obs2mod_kernel" not in per_launch_us and "raycast_trace_kernel" in per_launch_us:
            # the trace and the obs->mod term share one launch (trace_and_obs2mod_kernel, timed under the ray-cast class): its
            # work is the sum of the two
            U = "trace_and_obs2mod_kernel"
            bytes_per_launch[U] = bytes_per_launch.pop("raycast_trace_kernel") + bytes_per_launch.pop("fused_obs2mod_kernel")
            lk[U] = lk.pop("raycast_trace_kernel") + lk.pop("fused_obs2mod_kernel")
            per_launch_us[U] = per_launch_us.pop("raycast_trace_kernel")
            per_step_us[U] = per_step_us.pop("raycast_trace_kernel")
            launches[U] = launches.pop("raycast_trace_kernel")
        cands = {k: v for k, v in per_step_us.items() if k in bytes_per_launch}
        name = max(cands, key=cands.get)
        roofline = roofline_entry(name, bytes_per_launch[name], lk.get(name, 0.0), per_launch_us)
        it_bytes = sum(bytes_per_launch[k] for k in ("fused_obs2mod_kernel", "raycast_trace_kernel", "trace_and_obs2mod_kernel",
                                                     "ne_mod2obs_kernel", "gn_reduce_kernel", "gn_solve_single_kernel") if k in bytes_per_launch)
        chain = [k for k in ("trace_and_obs2mod_kernel", "raycast_trace_kernel", "ne_mod2obs_kernel", "gn_reduce_kernel", "gn_solve_single_kernel")
                 if k in launches]
        roofline.update({
            "per_step_us": per_step_from_rocprof(launches, False, chain, elapsed / max(n_steps, 1)),
            "per_step_us_events": {k: round(v, 2) for k, v in per_step_us.items()},
            "launches_per_step": {k: round(v, 2) for k, v in launches.items()},
            "per_kernel": {k: roofline_entry(k, bytes_per_launch[k], lk.get(k, 0.0), per_launch_us) for k in bytes_per_launch if k in per_launch_us},
            "whole_iteration": {"algorithmic_bytes": round(it_bytes), "achieved_GBs": round(it_bytes / (elapsed / max(n_steps * iters, 1)) / 1e9, 2),
                                "frac": round(it_bytes / (elapsed / max(n_steps * iters, 1)) / 1e9 / HBM_PEAK_GBS, 5)},
            "counts": {"valid_pixels": n_valid, "associated_per_iter": n_assoc, "predicted_per_iter": n_pred,
                       "trace_records_per_iter": cnt[0] / iters, "dense_pairs_per_iter": P * n_sdfs,
                       "lookups_performed_per_iter": {"association_scan": cnt[1] / iters, "association_gradient": cnt[2] / iters,
                                                      "raycast_march": cnt[4] / iters, "mod2obs": cnt[5] / iters}},
            "note": "latency-bound at batch 1: one frame's working set (SDFs L2/MALL-resident) cannot approach the HBM roof; an "
                    "iteration is four kernels back to back on one stream (trace + obs->mod in one launch, mod->obs, reduce, solve), "
                    "the observation-SDF build of the call runs beside the first of them on a second stream. per_step_us_events: HIP "
                    "events (two records per launch stretch every kernel by 2-4 us, so their sum exceeds the step); per_step_us: the "
                    "committed rocprofv3 medians"})
        if not args.no_cpu_baseline:
            nthr = cpu_threads(args)

            def time_cpu(threads, n):
                O.set_threads(threads)
                tc0 = time.perf_counter()
                for _ in range(n):
                    O.optimize_pose(om, sc.init_T_mc, sc.init_joints, v_np, n_np, o_struct)
                dt_ = time.perf_counter() - tc0
                O.set_threads(1)
                return n * iters / dt_
            one = time_cpu(1, args.cpu_steps)
            many = time_cpu(nthr, max(args.cpu_steps, 10 * nthr))
            sample = "whole steps (obs-SDF build + %d iterations) of the same 640x480 Allegro workload, CPU oracle (OpenMP over pixels / rays)" % iters
            cpu_baseline = {"value": round(many, 3), "unit": "GN iterations/s", "cores": nthr, "kind": "port", "cpu_model": cpu_model_name(),
                            "sample": "%d %s, %d threads" % (max(args.cpu_steps, 10 * nthr), sample, nthr),
                            "single_thread": {"value": round(one, 3), "cores": 1, "sample": "%d %s, 1 thread" % (args.cpu_steps, sample)}}

    # ================================================================== the headline workload as a BATCH of independent frames
    # (dartk_optimize_pose_batch: Optimizer::optimizePose per frame, every launch covers a chunk of 64 frames; what each GPU runs
    # when a recorded sequence is sharded by frame index) -- the form in which this workload can fill the chip
    pose_batch = None
    if not args.no_pose_batch:
        pf_idx = sharding.shard_indices(args.pose_frames, rank, world)
        nPF = len(pf_idx)
        tT, tj, iT, ij = configs.tracking_frames(sc, args.pose_frames)
        pv, pn = [], []
        for f in pf_idx:
            dm.set_pose(tT[f], tj[f])
            _, v_, n_ = synthetic_observation([dm], sc.width, sc.height, sc.fl, sc.pp)
            pv.append(v_); pn.append(n_)
        pvert, pnorm = torch.stack(pv).contiguous(), torch.stack(pn).contiguous()
        del pv, pn
        iT_d, ij_d = torch.from_numpy(iT[pf_idx]).to(dev), torch.from_numpy(ij[pf_idx]).to(dev)

        def pstep(debug=False):
            T_ = iT_d.clone(); j_ = ij_d.clone()
            opt.optimizePoseBatch(dm, pvert, pnorm, T_, j_, sc.opts, debug=debug)
            return T_, j_

        pstep()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.pose_steps):
            T_o, j_o = pstep()
        barrier()
        p_elapsed = max_over_ranks(time.perf_counter() - t0)
        rec_ = torch.cat([T_o, j_o], dim=1)
        sharding.gather_candidates(args.pose_frames, rank, world, torch.zeros(nPF, device=dev), rec_)      # the poses: the leg's only exchange
        p_roof = None
        if rank == 0:
            p_launch_us, p_step_us, p_launches = prof_collect(lib(), fptr, iptr, check, pstep, 1)
            n_c = min(nPF, 8)

            def psample(debug=False):
                T_ = iT_d[:n_c].clone(); j_ = ij_d[:n_c].clone()
                opt.optimizePoseBatch(dm, pvert[:n_c], pnorm[:n_c], T_, j_, sc.opts, debug=debug)
                torch.cuda.synchronize()
            pcnt = count_lookups(psample)
            psample(debug=True)
            pinfo = opt.iter_info.cpu().numpy()          # [n_c, iters, 8]
            per_fi = 1.0 / (n_c * iters)
            Pp = opt.pred_width * opt.pred_height
            Sp = 22 + 22 * 23 // 2 + 1
            Vp = float((pvert[:n_c, :, :, 3] > 0).sum().item()) / n_c
            recs = pcnt[0] * per_fi
            lkp = {"fused_obs2mod_kernel": (pcnt[1] + pcnt[2]) * per_fi, "raycast_trace_kernel": pcnt[4] * per_fi, "ne_mod2obs_kernel": pcnt[5] * per_fi}
            # bytes per frame and iteration: the single-frame leg's formulas with the lookups this batch performed
            bp = {"fused_obs2mod_kernel": 20.0 * Vp + 32.0 * lkp["fused_obs2mod_kernel"] + 16.0 * pcnt[2] / 6 * per_fi + 4.0 * Sp * ((Vp + 127) // 128),
                  "raycast_trace_kernel": 32.0 * lkp["raycast_trace_kernel"] + float(TRACE_BYTES_PER_PAIR) * recs,
                  "ne_mod2obs_kernel": float(TRACE_BYTES_PER_PAIR) * recs + 32.0 * Pp + 32.0 * lkp["ne_mod2obs_kernel"] + 4.0 * Sp * min((Pp + 127) // 128, 1024),
                  "gn_reduce_kernel": 4.0 * Sp * (min((Pp + 127) // 128, 1024) + (Vp + 127) // 128) + 8.0 * Sp,
                  "gn_solve_single_kernel": 4.0 * 2 * Sp + 96.0 * 17 * 2}
            ents = {}
            for k in bp:
                if k in p_launch_us:
                    n_l = p_launches[k]
                    ents[k] = roofline_entry(k, bp[k] * nPF * iters / n_l, lkp.get(k, 0.0) * nPF * iters / n_l, p_launch_us, batch=True,
                                             traffic_table=PMC_TRAFFIC_POSE_BATCH, rocprof_table=ROCPROF_POSE_BATCH, limiter_suffix="@pose_batch")
            if ents:
                dom = max(ents, key=lambda k: p_step_us.get(k, 0.0))
                p_roof = dict(ents[dom])
                it_b = sum(bp[k] for k in ents) * nPF * iters
                chain = [k for k in ("raycast_trace_kernel", "ne_mod2obs_kernel", "fused_obs2mod_kernel", "gn_reduce_kernel", "gn_solve_single_kernel") if k in p_launches]
                p_roof.update({
                    "per_step_us": per_step_from_rocprof(p_launches, True, chain, p_elapsed / args.pose_steps, ROCPROF_POSE_BATCH),
                    "per_step_us_events": {k: round(v, 2) for k, v in p_step_us.items()},
                    "launches_per_step": {k: round(v, 2) for k, v in p_launches.items()},
                    "per_kernel": ents,
                    "whole_step": {"algorithmic_bytes": round(it_b), "achieved_GBs": round(it_b / (p_elapsed / args.pose_steps) / 1e9, 2),
                                   "frac": round(it_b / (p_elapsed / args.pose_steps) / 1e9 / HBM_PEAK_GBS, 5)},
                    "counts_per_frame_and_iteration": {"valid_pixels": Vp, "associated": float(pinfo[:, :, 3].sum()) * per_fi,
                                                       "predicted": float(pinfo[:, :, 4].sum()) * per_fi, "trace_records": recs,
                                                       "lookups_performed": lkp, "sample_frames": n_c},
                    "note": "one stream, five launches per iteration and chunk of 64 frames (trace, replay + mod->obs, obs->mod, reduce, "
                            "solve); the step also holds the per-call work (pose import / export, valid-pixel lists, observation-SDF builds)"})
            pose_batch = {"metric": "single-model tracking, batch of independent frames (dartk_optimize_pose_batch)", "unit": "GN iterations/s",
                          "value": round(args.pose_frames * args.pose_steps * iters / p_elapsed, 2),
                          "frames_per_s": round(args.pose_frames * args.pose_steps / p_elapsed, 2), "scaling": "strong",
                          "n_frames": args.pose_frames, "frames_per_gpu": nPF, "steps": args.pose_steps,
                          "ms_per_step": round(1000.0 * p_elapsed / args.pose_steps, 3), "iterations_per_step": iters, "dtype": "f32",
                          "data": "synthetic", "roofline": p_roof,
                          "cpu_baseline": "the headline leg's: the same per-frame workload (cpu_baseline above)",
                          "config": {"workload": "BASELINE configs[1] per frame (Allegro 22-DOF, 640x480, prediction 320x240, 64^3 obs SDF, obs->mod + "
                                                 "mod->obs, %d GN iterations), %d frames with their own true and start poses, sharded by index over "
                                                 "%d GPU(s), one all-gather of the poses; every frame equals the single call bit for bit "
                                                 "(tests/test_gpu_pose_batch.py)" % (iters, args.pose_frames, world)}}
        del pvert, pnorm
        torch.cuda.empty_cache()

    # ================================================================== BASELINE config 4: batched ContactGrasp refinement
    grasps = None
    if not args.no_grasps:
        n_cand = args.grasp_candidates
        gs = configs.contact_grasp_scene(n_cand)
        idx = sharding.shard_indices(n_cand, rank, world)
        tr = configs.make_grasp_tracker(gs, max_batch=len(idx))
        T_all, j_all = configs.grasp_batch_arrays(gs)
        Td = torch.from_numpy(T_all[idx]).to(dev)
        jd = torch.from_numpy(j_all[idx]).to(dev)
        g_iters = gs.num_iterations

        def grefine():
            return tr.optimizePosesBatch(Td, jd, device_arrays=True)

        def gstep():
            T_o, j_o, e_o, _, _ = grefine()
            poses = torch.cat([T_o.reshape(len(idx), -1), j_o], dim=1)
            return sharding.gather_candidates(n_cand, rank, world, e_o, poses)     # RCCL all-gather of {energy, pose}

        gstep()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.grasp_steps):
            full_e, full_p = gstep()
        barrier()
        g_elapsed = max_over_ranks(time.perf_counter() - t0)
        if rank == 0:
            import oracle_lib as O
            best = sharding.best_candidates(full_e, 5)
            per_launch_us, per_step_us, launches = prof_collect(lib(), fptr, iptr, check, grefine, 1)
            # algorithmic bytes of one terms-kernel launch (contact priors + intersection) with the lookups the kernels
            # perform, counted by the diagnostic library on a sample of this rank's candidates (2 iterations + getError = 3
            # passes of every term kernel per candidate)
            oms = [O.OModel(gs.hand, 0), O.OModel(gs.obj, 1)]
            clouds = [c.cpu().numpy() for c in tr.clouds]
            pots = [p.cpu().numpy() for p in tr.potentials]
            n_cs = min(len(idx), 512)
            from dart_amd import _abi

            def counted():
                trc = configs.make_grasp_tracker(gs, max_batch=n_cs)
                trc.getOptions().numIterations = 2
                trc.optimizePosesBatch(Td[:n_cs].clone(), jd[:n_cs].clone(), device_arrays=True)
                torch.cuda.synchronize()
            cnt = count_lookups(counted)
            lk_inter, lk_prior = cnt[6] / (3.0 * n_cs), cnt[7] / (3.0 * n_cs)
            Q = len(gs.priors)
            Cn = sum(len(c) for c in clouds) + len(clouds[0])      # the hand cloud is walked twice (self + object)
            slab_inter, slab_prior = tr_slab_floats(tr)
            K_I, K_P = "noobs_terms_kernel<intersection>", "noobs_terms_kernel<priors>"
            # per candidate and pass: 40 B per contact prior / 16 B per cloud point + 32 B per trilinear lookup performed + partial
            # systems + the candidate's pose record (transforms of both models, read by every workgroup)
            pose_bytes = 4.0 * sum(24 + max(m.n_joints, 1) + 24 * m.n_frames for m in (gs.hand, gs.obj))
            # intersection launches per pass: one (the terms share a launch, noobs_inter_union_kernel) or one per term; the class's
            # figure is the terms' sum over the class's launches
            n_il = max(launches.get(K_I, 1.0) / max(launches.get(K_P, 1.0), 1.0), 1.0)
            bytes_per_launch = {K_P: (40.0 * Q + 32.0 * lk_prior + 4.0 * slab_prior + pose_bytes) * len(idx),
                                K_I: (16.0 * Cn + 32.0 * lk_inter + 4.0 * slab_inter + 3 * pose_bytes) * len(idx) / n_il}
            lk_launch = {K_P: lk_prior * len(idx), K_I: lk_inter * len(idx) / n_il}
            g_name = max((K_I, K_P), key=lambda k: per_step_us.get(k, 0.0))
            g_roof = roofline_entry(g_name, bytes_per_launch[g_name], lk_launch[g_name], per_launch_us, batch=True)
            g_roof.update({"per_step_us": per_step_from_rocprof(launches, True, [k for k in (K_I, K_P, "noobs_solve_kernel") if k in launches],
                                                                g_elapsed / args.grasp_steps),
                           "per_step_us_events": {k: round(v, 2) for k, v in per_step_us.items()},
                           "per_kernel": {k: roofline_entry(k, bytes_per_launch[k], lk_launch[k], per_launch_us, batch=True)
                                          for k in bytes_per_launch if k in per_launch_us},
                           "lookups_performed_per_candidate_per_pass": {"intersection": lk_inter, "priors": lk_prior},
                           "note": "the term kernels are instruction-issue bound (sphere / box rejects and exact bound tests per "
                                   "(point, SDF) pair) and their SDF working set is L2-resident: neither HBM nor gather bandwidth "
                                   "is the limiter, see DESIGN.md"})
            g_cpu = None
            if not args.no_cpu_baseline:
                o2 = tr._options_struct()
                o2.numIterations = 10
                nthr = cpu_threads(args)

                def one_candidate(k):
                    c = idx[k % len(idx)]
                    O.optimize_poses_noobs(oms, [T_all[c, 0], T_all[c, 1]], [j_all[c], np.zeros(0, dtype=np.float32)], o2, clouds, pots, gs.priors)

                def time_cpu(threads, n):
                    tc0 = time.perf_counter()
                    if threads == 1:
                        for k in range(n):
                            one_candidate(k)
                    else:       # candidates are independent: one per thread (ctypes releases the GIL during the call)
                        from concurrent.futures import ThreadPoolExecutor
                        with ThreadPoolExecutor(threads) as ex:
                            list(ex.map(one_candidate, range(n)))
                    return n * 10 / g_iters / (time.perf_counter() - tc0)
                one = time_cpu(1, 96)
                many = time_cpu(nthr, 64 * nthr)
                sample = "candidates x 10 of the %d iterations (+ getError), CPU oracle, scaled to %d iterations" % (g_iters, g_iters)
                g_cpu = {"value": round(many, 4), "unit": "grasps/s", "cores": nthr, "kind": "port", "cpu_model": cpu_model_name(),
                         "sample": "%d %s, %d threads (one candidate per thread)" % (64 * nthr, sample, nthr),
                         "single_thread": {"value": round(one, 4), "cores": 1, "sample": "96 %s, 1 thread" % sample}}
            grasps = {"metric": "batched ContactGrasp refinement", "value": round(n_cand * args.grasp_steps / g_elapsed, 2), "unit": "grasps/s",
                      "gn_iterations_per_s": round(n_cand * args.grasp_steps * g_iters / g_elapsed, 1),
                      "scaling": "strong", "n_candidates": n_cand, "candidates_per_gpu": len(idx), "iterations": g_iters,
                      "steps": args.grasp_steps, "ms_per_step": round(1000.0 * g_elapsed / args.grasp_steps, 3),
                      "config": {"workload": "BASELINE configs[3]: %d HumanHand (26 DOF, 16 SDFs, proxy geometry) + ellipsoid object candidates, "
                                             "%d contact priors (500 attract / 2000 repulse / thumb), hand self- and hand<->object "
                                             "intersection, %d iterations + getError, round-robin over %d GPU(s), one all-gather of "
                                             "{energy, pose}" % (n_cand, Q, g_iters, world)},
                      "best_candidates": [int(b) for b in best],
                      "roofline": g_roof,
                      "cpu_baseline": g_cpu}

    # third leg: BASELINE configs[4] (Barrett 8->4 reduced + box object, 1280x720, multi-model tracking with observations), a
    # batch of independent frames sharded by frame index over the ranks: every launch of the device-resident
    # Optimizer::optimizePoses loop covers all frames of the rank (dartk_optimize_poses_batch); one all-gather of the poses
    multi = None
    if not args.no_multi_model:
        m_iters = 10
        n_frames = args.frames
        f_idx = sharding.shard_indices(n_frames, rank, world)
        msc = configs.config5(1280, 720, 0)
        mtr = configs.make_multi_tracker(msc)
        mtr.getOptions().numIterations = m_iters
        mvert, mnorm, mT0, ma0 = configs.config5_batch(mtr, msc, f_idx)
        mT0_d, ma0_d = torch.from_numpy(mT0).to(dev), torch.from_numpy(ma0).to(dev)

        def mstep():
            T_o, a_o = mtr.optimizePosesBatch(mvert, mnorm, mT0_d, ma0_d)
            rec_ = torch.cat([T_o.reshape(len(f_idx), -1), a_o], dim=1)
            return sharding.gather_candidates(n_frames, rank, world, torch.zeros(len(f_idx), device=dev), rec_)    # RCCL all-gather of the poses

        mstep()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.multi_steps):
            mstep()
        barrier()
        m_elapsed = max_over_ranks(time.perf_counter() - t0)
        # the same frames one at a time through the single-frame device loop (what round 1 measured), on a sample
        single_its = single_its_carried = None
        if rank == 0:
            n_s = min(len(f_idx), 8)
            for m in range(2):
                mtr.getPose(m).setTransformCameraToModel(mT0[0, m])
            mtr.getPose(0).getReducedArticulation()[:] = ma0[0]
            mtr.optimizePosesDevice(mvert[0], mnorm[0], readback=False)         # one untimed frame (first-call set-up of this path)
            torch.cuda.synchronize()
            ts0 = time.perf_counter()
            for k in range(n_s):
                for m in range(2):
                    mtr.getPose(m).setTransformCameraToModel(mT0[k, m])
                mtr.getPose(0).getReducedArticulation()[:] = ma0[k]
                mtr.optimizePosesDevice(mvert[k], mnorm[k], readback=False)     # poses stay on the device between frames
            torch.cuda.synchronize()
            single_its = n_s * m_iters / (time.perf_counter() - ts0)
            # the same calls as a tracker makes them: every frame starts from the poses the previous call left on the device
            ts1 = time.perf_counter()
            for k in range(n_s):
                mtr.optimizePosesDevice(mvert[k], mnorm[k], readback=False, from_device=True)
            torch.cuda.synchronize()
            single_its_carried = n_s * m_iters / (time.perf_counter() - ts1)
            # ---- roofline of this leg: HIP-event durations of one batch step with the three chains on ONE stream (side by side
            # on three streams every kernel's duration is stretched by the others and says nothing), the lookups its kernels
            # perform counted by the diagnostic library on a sample of the frames, algorithmic bytes per kernel class and step
            import oracle_lib as O
            nF = len(f_idx)
            os.environ["DARTK_MULTI_ONE_STREAM"] = "1"
            try:
                mb = lambda: mtr.optimizePosesBatch(mvert, mnorm, mT0_d, ma0_d)      # noqa: E731
                m_launch_us, m_step_us, m_launches = prof_collect(lib(), fptr, iptr, check, mb, 1)
            finally:
                del os.environ["DARTK_MULTI_ONE_STREAM"]
            n_c = min(nF, 8)
            mcnt = count_lookups(lambda: (mtr.optimizePosesBatch(mvert[:n_c], mnorm[:n_c], mT0_d[:n_c], ma0_d[:n_c]), torch.cuda.synchronize()))
            mtr.optimizePosesBatch(mvert[:n_c], mnorm[:n_c], mT0_d[:n_c], ma0_d[:n_c], debug=True)
            minfo = mtr.iteration_summaries            # [n_c, iters, nModels, 8]
            per_fi = 1.0 / (n_c * m_iters)             # per frame and iteration
            V = float((mvert[:n_c, :, :, 3] > 0).sum().item()) / n_c
            N_a = float(minfo[:, :, :, 3].sum()) * per_fi
            N_p = float(minfo[:, :, :, 4].sum()) * per_fi
            rec = mcnt[0] * per_fi
            lk5 = {"assoc_multi_kernel": mcnt[1] * per_fi, "fused_obs2mod_kernel": mcnt[3] * per_fi, "raycast_trace_kernel": mcnt[4] * per_fi,
                   "ne_mod2obs_kernel": mcnt[5] * per_fi, "ne_self_intersection_kernel": mcnt[6] * per_fi}
            Cn = float(sum(int(c.shape[0]) for c in mtr.clouds) + int(mtr.clouds[0].shape[0]))     # hand cloud: self + pairwise
            # bytes per frame and iteration (SURVEY 8d with the lookups PERFORMED); the association's 16 B per valid pixel are read
            # from the vertex map once per call and listed (4 B index), then every iteration walks the list
            b5 = {"assoc_multi_kernel": 16.0 * V / m_iters + 20.0 * V + 32.0 * lk5["assoc_multi_kernel"] + 12.0 * N_a,
                  "fused_obs2mod_kernel": 28.0 * N_a + 32.0 * lk5["fused_obs2mod_kernel"],
                  "raycast_trace_kernel": 32.0 * lk5["raycast_trace_kernel"] + float(TRACE_BYTES_PER_PAIR) * rec,
                  "raycast_replay_kernel": float(TRACE_BYTES_PER_PAIR) * rec + 16.0 * N_p,
                  "ne_mod2obs_kernel": 16.0 * N_p * 2 + 32.0 * lk5["ne_mod2obs_kernel"],
                  "ne_self_intersection_kernel": 16.0 * Cn + 32.0 * lk5["ne_self_intersection_kernel"]}
            ents = {}
            for k in b5:
                if k not in m_launch_us:
                    continue
                n_l = m_launches[k]
                ents[k] = roofline_entry(k, b5[k] * nF * m_iters / n_l, lk5.get(k, 0.0) * nF * m_iters / n_l, m_launch_us, batch=True,
                                         traffic_table=PMC_TRAFFIC_BATCH)
            m_roof = None
            if ents:
                dom = max(ents, key=lambda k: m_step_us.get(k, 0.0))
                m_roof = dict(ents[dom])
                it_b = sum(b5[k] for k in ents) * nF * m_iters
                m_roof.update({
                    "per_step_us": per_step_from_rocprof(m_launches, True, [], m_elapsed / args.multi_steps),
                    "per_step_us_events": {k: round(v, 2) for k, v in m_step_us.items()},
                    "launches_per_step": {k: round(v, 2) for k, v in m_launches.items()},
                    "per_kernel": ents,
                    "whole_step": {"algorithmic_bytes": round(it_b), "achieved_GBs": round(it_b / (m_elapsed / args.multi_steps) / 1e9, 2),
                                   "frac": round(it_b / (m_elapsed / args.multi_steps) / 1e9 / HBM_PEAK_GBS, 5)},
                    "counts_per_frame_and_iteration": {"valid_pixels": V, "associated": N_a, "predicted": N_p, "trace_records": rec,
                                                       "lookups_performed": lk5, "sample_frames": n_c},
                    "note": "event durations from one batch step with the loop's three chains on one stream (DARTK_MULTI_ONE_STREAM); "
                            "the timed steps run them side by side on three streams, so the summed durations exceed the step; "
                            "a launch covers a chunk of up to 64 frames"})
            # ---- CPU baseline of this leg: the oracle's Optimizer::optimizePoses loop on the same frames (host copies of the maps)
            m_cpu = None
            if not args.no_cpu_baseline:
                nthr = cpu_threads(args)
                n_cpu = min(nF, 3)
                v_h = [mvert[k].cpu().numpy() for k in range(n_cpu)]
                n_h = [mnorm[k].cpu().numpy() for k in range(n_cpu)]

                def cpu_frames(threads, n):
                    O.set_threads(threads)
                    tc0 = time.perf_counter()
                    for k in range(n):
                        kk = k % n_cpu
                        O.optimize_poses(mtr, v_h[kk], n_h[kk], [mT0[kk, 0], mT0[kk, 1]], [ma0[kk], np.zeros(0, dtype=np.float32)])
                    dt_ = time.perf_counter() - tc0
                    O.set_threads(1)
                    return n * m_iters / dt_
                one = cpu_frames(1, 2)
                n_many = max(n_cpu, min(24, int(4 * one)))
                many = cpu_frames(nthr, n_many)
                smp = "frames x %d iterations (obs-SDF build + loop) of the same 1280x720 two-model workload, CPU oracle" % m_iters
                m_cpu = {"value": round(many, 3), "unit": "GN iterations/s", "cores": nthr, "kind": "port", "cpu_model": cpu_model_name(),
                         "sample": "%d %s, %d threads (OpenMP over pixels / rays / points)" % (n_many, smp, nthr),
                         "single_thread": {"value": round(one, 3), "cores": 1, "sample": "2 %s, 1 thread" % smp}}
            multi = {"metric": "multi-model tracking, batch of independent frames (dartk_optimize_poses_batch)", "unit": "GN iterations/s",
                     "value": round(n_frames * args.multi_steps * m_iters / m_elapsed, 2),
                     "frames_per_s": round(n_frames * args.multi_steps / m_elapsed, 2), "scaling": "strong", "n_frames": n_frames,
                     "frames_per_gpu": len(f_idx), "steps": args.multi_steps, "ms_per_step": round(1000.0 * m_elapsed / args.multi_steps, 3),
                     "iterations_per_step": m_iters, "single_frame_loop_GN_its": round(single_its, 2), "single_frame_loop_from_device_GN_its": round(single_its_carried, 2) if single_its_carried else None, "dtype": "f32", "data": "synthetic",
                     "roofline": m_roof, "cpu_baseline": m_cpu,
                     "config": {"workload": "BASELINE configs[4]: %d frames of Barrett hand (8 joints reduced to 4, proxy geometry) + box object, "
                                            "synthetic 1280x720 depth per frame, obs->mod + mod->obs + self / pairwise intersection in reduced "
                                            "dimensions, %d GN iterations per frame, frames sharded by index over %d GPU(s), one all-gather of "
                                            "the poses; single_frame_loop_GN_its = the same frames one call at a time, each from its own uploaded start pose; "
                                            "..._from_device_... = each from the poses the previous call left on the device" % (n_frames, m_iters, world)}}

    if rank == 0:
        value = world * n_steps * iters / elapsed if n_steps else 0.0
        line = {
            "metric": "Gauss-Newton iters/sec (640x480 Allegro)", "value": round(value, 2), "unit": "GN iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000.0 * elapsed / max(n_steps, 1), 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: Allegro 22-DOF (17 SDFs, proxy geometry), synthetic 640x480 depth, "
                                   "prediction 320x240, 64^3 obs SDF, obs->mod + mod->obs, %d GN iterations per optimizePose call, "
                                   "normThreshold %.1f; one frame per GPU" % (iters, sc.opts.normThreshold),
                       "iterations_per_step": iters, "frames_per_gpu": 1, "parallelism": "independent frames, 1 per GPU"},
            "value_default_gate": value_default_gate, "repeats": repeats, "roofline": roofline, "cpu_baseline": cpu_baseline, "tracking_batch": pose_batch, "batched_grasps": grasps, "multi_model_tracking": multi,
        }
        print(json.dumps(line, default=float))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
