"""Trajectory parity in segments. A Gauss-Newton run is a sequence of discrete decisions (which pixels associate, which rays
hit) and continuous steps. The HIP path and the oracle add their partial sums in different orders, so their systems agree to
~1e-7, and as long as every discrete decision is the same the poses stay together. When a borderline pixel flips in some
iteration, the two runs legitimately take slightly different steps from there on and may drift apart on weakly observed
joints. Instead of loosening the bar after a flip, the oracle is RESTARTED from the GPU's pose at the flip and the comparison
goes on: every segment -- from a common pose up to and including the next iteration whose decisions differ -- is held to
the strict bar (north_star: 1e-3 rad / 1e-3 m), over ALL iterations. Test infrastructure only."""
import numpy as np


def segmented_parity(name, n_iter, gpu_prefix, oracle_from, decisions_equal, pose_err, bar=1e-3):
    """gpu_prefix(k) -> (pose after k iterations from the common start, info rows [>= k]);  k = 0 gives the start pose.
    oracle_from(pose, k0, n) -> (pose after n more iterations, info rows [n]) with `pose` = the GPU's pose after k0 iterations.
    decisions_equal(gpu_row, oracle_row) -> bool.  pose_err(a, b) -> float (max of metres / radians).
    Returns the list of (first iteration, end iteration, error) per segment."""
    cache = {}

    def gp(k):
        if k not in cache:
            cache[k] = gpu_prefix(k)
        return cache[k]

    g_info = gp(n_iter)[1]
    k0, segments = 0, []
    while k0 < n_iter:
        start = gp(k0)[0]
        o_pose, o_info = oracle_from(start, k0, n_iter - k0)
        flips = [i for i in range(n_iter - k0) if not decisions_equal(g_info[k0 + i], o_info[i])]
        if not flips:
            end = n_iter
        else:
            end = k0 + flips[0] + 1            # the flipped iteration's step is part of the segment
            if end < n_iter:
                o_pose, _ = oracle_from(start, k0, end - k0)
        err = pose_err(gp(end)[0], o_pose)
        segments.append((k0, end, err))
        assert err < bar, "%s: iterations %d..%d differ by %.3e (bar %.1e)" % (name, k0, end, err, bar)
        k0 = end
    if len(segments) == 1:
        print("%s: branch NO-FLIP: all %d iterations in one free-running segment, pose error %.2e (bar %.0e)" % (name, n_iter, segments[0][2], bar))
    else:
        print("%s: branch RESTART: %d segments %s, each held to %.0e (oracle restarted from the GPU pose after every decision flip)"
              % (name, len(segments), ["%d-%d: %.1e" % s for s in segments], bar))
    return segments
