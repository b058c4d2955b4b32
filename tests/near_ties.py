"""Documentation of discrete-decision flips (SURVEY 7 iii / north_star: mismatches are listed and each is shown to be a near-tie).
Two runs whose poses differ by rounding (sums added in a different order) decide differently only for items that sit on a
threshold: the functions here take the two poses, evaluate every item's deciding quantity at both with the CPU oracle, list the
items whose decision differs and check that the threshold lies BETWEEN the two values -- i.e. the item's margin is no larger than
the change the pose difference causes. Test infrastructure only."""
import numpy as np

import oracle_lib as O

F = np.float32


def _assoc_maps(om, vert, norm, T_mc, joints, dist_thr, norm_thr, plane_offset, plane_normal):
    T_mf, T_fm = om.fk(joints)
    _, gated = O.assoc_single(vert, norm, om, T_mc, T_fm, dist_thr=dist_thr, norm_thr=norm_thr, plane_offset=plane_offset, plane_normal=plane_normal)
    _, free = O.assoc_single(vert, norm, om, T_mc, T_fm, dist_thr=1e9, norm_thr=-1e9, plane_offset=plane_offset, plane_normal=plane_normal)
    dots = np.einsum("hwk,hwk->hw", free["norm"][..., :3].astype(np.float64), norm[..., :3].astype(np.float64))
    return gated["da"].reshape(-1), free["da"].reshape(-1), free["err"].reshape(-1), dots.reshape(-1)


def association_near_ties(om, vert, norm, opts, pose_a, pose_b, model_index=0):
    """Pixels whose single-model association (obsToMod.cu:12-114) differs between pose_a and pose_b = (T_mc, joints), one record
    each: the signed distance to the nearest SDF and the normal-gate dot product at both poses, the margins to the distance
    threshold / normal gate, and which criterion the pose difference carried across its threshold."""
    m = model_index
    thr, nthr = float(opts.distThreshold[m]), float(opts.normThreshold)
    args = (thr, nthr, float(opts.planeOffset[m]), tuple(opts.planeNormal[m]))
    da_a, near_a, d_a, dot_a = _assoc_maps(om, vert, norm, pose_a[0], pose_a[1], *args)
    da_b, near_b, d_b, dot_b = _assoc_maps(om, vert, norm, pose_b[0], pose_b[1], *args)
    out = []
    for px in np.nonzero(da_a != da_b)[0]:
        dA, dB, nA, nB = float(d_a[px]), float(d_b[px]), float(dot_a[px]), float(dot_b[px])
        rec = {"pixel": int(px), "association": [int(da_a[px]), int(da_b[px])], "nearest_sdf": [int(near_a[px]), int(near_b[px])],
               "distance_m": [dA, dB], "distance_threshold_m": thr, "normal_dot": [nA, nB], "normal_threshold": nthr}
        if np.isfinite(dA) and np.isfinite(dB) and (abs(dA) <= thr) != (abs(dB) <= thr):
            rec["kind"] = "distance threshold"
            rec["margin"] = min(abs(abs(dA) - thr), abs(abs(dB) - thr))
            rec["change"] = abs(abs(dA) - abs(dB))
        elif np.isfinite(nA) and np.isfinite(nB) and (nA < nthr) != (nB < nthr):
            rec["kind"] = "normal gate"
            rec["margin"] = min(abs(nA - nthr), abs(nB - nthr))
            rec["change"] = abs(nA - nB)
        elif near_a[px] != near_b[px] and near_a[px] >= 0 and near_b[px] >= 0:
            rec["kind"] = "nearest SDF"          # two SDFs equally near: the arg-min changes hands (strict <, lowest index wins)
            rec["margin"] = rec["change"] = abs(dA - dB)
        else:
            rec["kind"] = "bounds"               # the point enters / leaves an SDF's interpolation bounds (or the plane test)
            rec["margin"] = rec["change"] = float("nan")
        out.append(rec)
    return out


def assert_explained(records, what):
    """every listed flip is a near-tie: its threshold lies between the values at the two poses (margin <= change)"""
    for r in records:
        assert r["kind"] != "bounds", "%s: item %r changed its decision without a threshold between the two poses" % (what, r)
        assert r["margin"] <= r["change"] * (1 + 1e-6) + 1e-12, "%s: %r is not a near-tie of the two poses" % (what, r)


def prior_near_ties(oms, priors, pose_a, pose_b, dims, offs):
    """Contact priors (contact_prior.cpp:17-95) whose state differs between pose_a and pose_b = ([T_mc per model], [joints per
    model]): inactive (constant error sign*w*threshold) / active (error sign*w*|distance|) / destination SDF. One record each with
    the error at both poses."""
    def evaluate(pose):
        T_mcs, joints = pose
        fk = [om.fk(j) for om, j in zip(oms, joints)]
        T_cms = [O.se3_invert(T) for T in T_mcs]
        T_mfs, T_fms = [x[0] for x in fk], [x[1] for x in fk]
        D = sum(dims)
        res = []
        for p in priors:
            JTJ = np.zeros((D + 3, D + 3), dtype=F)
            JTe = np.zeros(D + 3, dtype=F)
            res.append(O.contact_prior_contribution(oms, p, T_mcs, T_cms, T_fms, T_mfs, offs, dims, D, JTJ, JTe))
        return res
    ra, rb = evaluate(pose_a), evaluate(pose_b)
    out = []
    for i, (p, (ea, sa), (eb, sb)) in enumerate(zip(priors, ra, rb)):
        const = (-1.0 if p.invert else 1.0) * p.weight * p.contactThreshold
        act_a, act_b = ea != F(const), eb != F(const)
        if act_a != act_b or (act_a and act_b and sa != sb):
            w = p.weight if p.weight else 1.0
            out.append({"prior": i, "active": [bool(act_a), bool(act_b)], "destination_sdf": [sa, sb], "error": [ea, eb],
                        "inactive_error": const, "invert": bool(p.invert),
                        "kind": "contact threshold / cylinder gate" if act_a != act_b else "nearest SDF",
                        "margin_m": abs(abs(ea if act_a else eb) / w - p.contactThreshold) if act_a != act_b else abs(ea - eb) / w})
    return out
