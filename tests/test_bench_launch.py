"""bench.py's launch path: `python bench.py --gpus N` starts N ranks itself (torch.distributed.run as a child process, before
the parent touches the GPU), refuses when fewer than N GPUs are visible, and a rank refuses a launcher whose world size is not
--gpus. On the 1-GPU box the two-rank form is rehearsed with --rehearse-on-one-gpu (both ranks on cuda:0, gloo between them):
the sharding / gather / max-over-ranks code of the multi-GPU run, not a measurement."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_gpus_flag_without_enough_devices_is_refused():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the refusal needs fewer devices than ranks")
    r = run(["--gpus", "2", "--steps", "1", "--warmup", "0"], timeout=120)
    assert r.returncode == 2 and "--gpus 2 but only" in r.stderr and not r.stdout.strip()
    r = run(["--gpus", "0"], timeout=120)
    assert r.returncode != 0 and "at least 1" in r.stderr


def test_rank_refuses_a_world_size_that_is_not_gpus():
    r = run(["--gpus", "2", "--steps", "1"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, timeout=120)
    assert r.returncode != 0 and "--gpus 2 but the launcher started 1 rank" in r.stderr
    r = run(["--gpus", "1", "--steps", "1"], env_extra={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"}, timeout=120)
    assert r.returncode != 0 and "--gpus 1 but the launcher started 4 rank" in r.stderr


@pytest.mark.gpu
def test_two_ranks_rehearsed_on_one_gpu(gpu):
    r = run(["--gpus", "2", "--rehearse-on-one-gpu", "--steps", "2", "--warmup", "1", "--repeats", "0", "--no-cpu-baseline",
             "--grasp-steps", "1", "--multi-steps", "1", "--frames", "6", "--pose-frames", "10", "--pose-steps", "1"], timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0: %r" % r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["value"] > 0 and j["scaling"] == "weak"
    assert j["batched_grasps"]["n_candidates"] == 4096 and j["batched_grasps"]["candidates_per_gpu"] == 2048
    assert j["multi_model_tracking"]["n_frames"] == 6 and j["multi_model_tracking"]["frames_per_gpu"] == 3
    assert len(j["batched_grasps"]["best_candidates"]) == 5
    assert j["tracking_batch"]["n_frames"] == 10 and j["tracking_batch"]["frames_per_gpu"] == 5 and j["tracking_batch"]["value"] > 0
