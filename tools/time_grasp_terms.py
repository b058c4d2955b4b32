"""Times the batched NoObs loop with subsets of the cost terms, to see where noobs_terms_kernel spends its time."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from dart_amd import configs, device as D
from dart_amd.tracker_no_obs import TrackerNoObs
D.init(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sc = configs.contact_grasp_scene(B)
T, j = configs.grasp_batch_arrays(sc)
Td = torch.from_numpy(T).cuda(); jd = torch.from_numpy(j).cuda()
print("hand sdfs", len(sc.hand.sdfs), "priors", len(sc.priors))


def make(self_on, inter_on, priors):
    tr = TrackerNoObs(max_batch=B)
    h = tr.addModel(sc.hand, cloud_seed=11)
    tr.setIntersectionPotentialMatrix(h, sc.potential)
    o_ = tr.addModel(sc.obj, cloud_seed=12)
    for p in priors:
        tr.addPrior(p)
    n = 2
    o = tr.getOptions()
    o.numIterations = sc.num_iterations
    if self_on: o.lambdaIntersection[h + n * h] = sc.lambda_intra
    if inter_on & 1: o.lambdaIntersection[o_ + n * h] = sc.lambda_inter
    if inter_on & 2: o.lambdaIntersection[h + n * o_] = sc.lambda_inter
    o.regularization = [sc.regularization] * n
    o.regularizationScaled = [sc.lm_damping] * n
    return tr


attract = [p for p in sc.priors if not p.invert]
repulse = [p for p in sc.priors if p.invert]
for name, args in (("full", (1, 3, sc.priors)), ("self only", (1, 0, [])), ("inter a only", (0, 1, [])), ("inter b only", (0, 2, [])),
                   ("priors only", (0, 0, sc.priors)), ("attract only", (0, 0, attract)), ("repulse only", (0, 0, repulse)),
                   ("one prior", (0, 0, sc.priors[:1]))):
    tr = make(*args)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = tr.optimizePosesBatch(Td, jd, device_arrays=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-14s B=%d: %.2f ms / 40 it = %.3f ms/it" % (name, B, dt * 1e3, dt * 1e3 / 40))

# the same subsets, one iteration from the poses the full loop converged to (rows are dense there)
full = make(1, 3, sc.priors)
out = full.optimizePosesBatch(Td, jd, device_arrays=True)
Tc, jc = out[0].clone(), out[1].clone()
print("-- one iteration at the converged poses")
for name, args in (("full", (1, 3, sc.priors)), ("self only", (1, 0, [])), ("inter a only", (0, 1, [])), ("inter b only", (0, 2, [])),
                   ("priors only", (0, 0, sc.priors)), ("attract only", (0, 0, attract)), ("repulse only", (0, 0, repulse)),
                   ("one prior", (0, 0, sc.priors[:1]))):
    tr = make(*args)
    tr.getOptions().numIterations = 1
    ts = []
    for rep in range(5):
        a, b = Tc.clone(), jc.clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tr.optimizePosesBatch(a, b, device_arrays=True)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("%-14s B=%d: %.3f ms/it" % (name, B, min(ts) * 1e3))
