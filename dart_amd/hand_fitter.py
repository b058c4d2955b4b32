"""HandFitter (src/hand_fitter.{hpp,cpp}) without the GUI: fits the articulated hand to 3-D keypoint targets with
Point3D3DPrior terms on a TrackerNoObs, from several DOF initialisations.

The reference runs the starts one after the other (hand_fitter.cpp:334-356); they are independent problems, so here they
are one batch of candidates on the device path (TrackerNoObs.optimizePosesBatch)."""
import numpy as np

from . import se3
from .tracker_no_obs import Point3D3DPrior, TrackerNoObs

F = np.float32

# geometry name -> frame id of HumanHand (hand_fitter.cpp:104-120)
GEOM_NAME2ID = {"palm": 0, "index1": 2, "index2": 3, "index3": 4, "mid1": 6, "mid2": 7, "mid3": 8, "ring1": 10, "ring2": 11,
                "ring3": 12, "pinky1": 14, "pinky2": 15, "pinky3": 16, "thumb1": 18, "thumb2": 19, "thumb3": 20}


def read_keypoints(path, offsets=None):
    """keypoints.txt: `joint geom x y z` in millimetres, relative to the geometry's mesh; + meshes/<geom>/offset.txt
    (hand_fitter.cpp:121-152). offsets: {geom: (x, y, z)} in mm (the mesh directory is not part of the reference tree;
    missing entries count as zero). Returns {joint: (frame id, [outer point, inner point (0,0,0)])} in metres."""
    out = {}
    with open(path) as f:
        for line in f:
            t = line.split()
            if len(t) < 5:
                continue
            joint, geom = t[0], t[1]
            off = (offsets or {}).get(geom, (0.0, 0.0, 0.0))
            p = tuple((float(t[2 + k]) + off[k]) / 1000.0 for k in range(3))
            out[joint] = (GEOM_NAME2ID[geom], [p, (0.0, 0.0, 0.0)])
    return out


def read_dof_samples(path, n_dofs):
    """dof_samples.txt: one comma-separated DOF initialisation per line (hand_fitter.cpp:155-171)."""
    rows = []
    with open(path) as f:
        for line in f:
            v = [float(x) for x in line.replace(",", " ").split()]
            if v:
                rows.append((v + [0.0] * n_dofs)[:n_dofs])
    return np.asarray(rows, dtype=F).reshape(-1, n_dofs)


def umeyama(src, dst, with_scaling=True):
    """Eigen::umeyama: least-squares similarity transform dst ~ c R src + t for 3xN point sets. Returns a 4x4 matrix."""
    src, dst = np.asarray(src, dtype=np.float64), np.asarray(dst, dtype=np.float64)
    n = src.shape[1]
    mu_s, mu_d = src.mean(1, keepdims=True), dst.mean(1, keepdims=True)
    xs, xd = src - mu_s, dst - mu_d
    var_s = (xs ** 2).sum() / n
    sigma = xd @ xs.T / n
    U, d, Vt = np.linalg.svd(sigma)
    S = np.ones(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2] = -1
    R = U @ np.diag(S) @ Vt
    c = (d * S).sum() / var_s if with_scaling else 1.0
    T = np.eye(4)
    T[:3, :3] = c * R
    T[:3, 3] = (mu_d - c * R @ mu_s)[:, 0]
    return T


def estimate_srt(src, dst, n_iters=25, n_sample_pts=3, inlier_thresh=1e-3, rng=None):
    """HandFitter::estimate_srt (hand_fitter.cpp:205-270): RANSAC over 3-point Umeyama fits (squared-distance inlier test),
    then a fit on the largest inlier set. The reference draws its samples with std::random_shuffle (rand()); here a seeded
    generator."""
    src, dst = np.asarray(src, dtype=np.float64), np.asarray(dst, dtype=np.float64)
    if src.shape[1] < n_sample_pts:
        raise ValueError("need at least %d points, have %d" % (n_sample_pts, src.shape[1]))
    rng = rng or np.random.default_rng(0)
    best = []
    hs = np.vstack([src, np.ones((1, src.shape[1]))])
    for _ in range(n_iters):
        order = rng.permutation(src.shape[1])[:n_sample_pts]
        T = umeyama(src[:, order], dst[:, order], True)
        d2 = (((T @ hs)[:3] - dst) ** 2).sum(0)
        inl = np.nonzero(d2 < inlier_thresh)[0]
        if len(inl) > len(best):
            best = inl
    if len(best) < n_sample_pts:
        best = np.arange(src.shape[1])
    return umeyama(src[:, best], dst[:, best], True)


class HandFitter:
    def __init__(self, hand, potential, keypoints, dof_inits, max_batch=64):
        """hand: voxelised HostModel; keypoints: read_keypoints(); dof_inits: [n, n_joints]."""
        self.hand = hand
        self.keypoints = keypoints
        self.dof_inits = np.asarray(dof_inits, dtype=F).reshape(-1, hand.n_joints)
        # GUI defaults (hand_fitter.cpp:80-85)
        self._max_iterations, self._attract_weight = 40, 20.0
        self._intra_model_intersection, self._lm_damping, self._log10_regularization = 20.0, 60.0, -1.0
        # create_tracker (:379-393)
        self.tracker = TrackerNoObs(max_batch=max_batch)
        self.tracker.addModel(hand, cloud_seed=11)
        self.tracker.getOptions().lambdaIntersection = [0.0]
        self.tracker.setIntersectionPotentialMatrix(0, potential)
        self.joint_targets, self.joint2inout, self.priors = {}, {}, {}
        self.T_mc = se3.identity()
        self.joints = np.zeros(hand.n_joints, dtype=F)
        self.set_max_iterations(self._max_iterations)
        self.set_intra_model_intersection(self._intra_model_intersection)
        self.set_lm_damping(self._lm_damping)
        self.set_log10_regularization(self._log10_regularization)

    # ---- setters (the GUI callbacks)
    def set_max_iterations(self, m):
        self._max_iterations = m
        self.tracker.getOptions().numIterations = m

    def set_attract_weight(self, w):
        self._attract_weight = w
        for p in self.priors.values():
            p.setWeight(w)

    def set_intra_model_intersection(self, w):
        self._intra_model_intersection = w
        self.tracker.getOptions().lambdaIntersection[0] = w

    def set_lm_damping(self, w):
        self._lm_damping = w
        self.tracker.getOptions().regularizationScaled = [w]

    def set_log10_regularization(self, w):
        self._log10_regularization = w
        self.tracker.getOptions().regularization = [float(F(10.0) ** F(w))]

    # ---- set_targets (:181-192)
    def set_targets(self, joint_locations, inner_surface):
        self.joint_targets = {k: tuple(float(x) for x in v) for k, v in joint_locations.items()}
        self.joint2inout = {k: (1 if inner_surface[k] else 0) for k in joint_locations}

    def _frame_point(self, joint):
        frame, pts = self.keypoints[joint]
        return frame, pts[self.joint2inout[joint]]

    def src_frame_points_in_camera(self, T_mc, joints):
        """src_frame_point_in_camera (:25-34) for every target joint at the given pose."""
        T_cm = se3.invert(np.asarray(T_mc, dtype=F))
        T_mf = np.asarray(self.hand.forward_kinematics(np.asarray(joints, dtype=F), "dh"), dtype=F).reshape(-1, 12)
        out = {}
        for name in self.joint_targets:
            frame, p = self._frame_point(name)
            out[name] = se3.apply(se3.mul(T_cm, T_mf[frame]), np.asarray([p], dtype=F))[0]
        return out

    # ---- fit_rigid_joints + init (:273-331)
    def init(self, rng=None):
        self.T_mc = se3.identity()
        self.joints = np.zeros(self.hand.n_joints, dtype=F)
        rigid = [n for n in self.joint_targets if self.keypoints[n][0] == 0]        # palm points only
        cur = self.src_frame_points_in_camera(self.T_mc, self.joints)
        src = np.asarray([self.joint_targets[n] for n in rigid], dtype=np.float64).T
        dst = np.asarray([cur[n] for n in rigid], dtype=np.float64).T
        T = estimate_srt(src, dst, rng=rng)
        for name, cam in self.joint_targets.items():
            frame, frame_pt = self._frame_point(name)
            cam_T = (T @ np.asarray([cam[0], cam[1], cam[2], 1.0]))[:3]
            if name in self.priors:
                self.priors[name].setTargetCameraPoint(cam_T)
            else:
                pr = Point3D3DPrior(0, frame, tuple(float(v) for v in cam_T), tuple(frame_pt), self._attract_weight)
                self.priors[name] = pr
                self.tracker.addPrior(pr)
        return T

    # ---- calculate_error (:366-376)
    def calculate_error(self, T_mc, joints):
        cur = self.src_frame_points_in_camera(T_mc, joints)
        e = [np.linalg.norm(np.asarray(cur[n], dtype=F) - np.asarray(self.priors[n].getTargetCameraPoint(), dtype=F)) for n in self.priors]
        return float(np.mean(e))

    # ---- fit (:334-356): every DOF initialisation is one candidate of a batch
    def fit(self):
        n = len(self.dof_inits)
        T = np.tile(se3.identity().astype(F), (n, 1, 1)).reshape(n, 1, 12)
        T_o, j_o, _, _, _ = self.tracker.optimizePosesBatch(T, self.dof_inits)
        errors = [self.calculate_error(T_o[c, 0], j_o[c]) for c in range(n)]
        best = int(np.argmin(errors))
        self.T_mc, self.joints = T_o[best, 0].copy(), j_o[best].copy()
        self.errors = errors
        return self.T_mc, self.joints, float(errors[best])
