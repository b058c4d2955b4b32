"""CPU: the damping matrices in the oracle's NoObs loop (optimizer.cpp:895-896: JTJ += *dampingMatrices[m] on the model's block
before the priors contribute) checked against a system assembled in numpy from the oracle's undamped first system."""
import numpy as np

import oracle_lib as O
from dart_amd import _abi, se3
from dart_amd.tracker import solve_upper
from test_oracle_contact_update import scene

F = np.float32


def test_noobs_damping_enters_the_upper_triangle_before_regularisation():
    sc, oms, priors = scene()
    priors = [type(p)(**{**p.__dict__, "update_contact_point": False}) for p in priors]     # frozen: the system is the model blocks
    T0, j0 = [sc.init_T_mc[0], se3.identity()], [sc.init_joints[0], np.zeros(0, dtype=F)]
    o = _abi.NoObsOptions()
    o.numIterations, o.nModels = 1, 2
    reg, lm = 1e-4, 25.0
    for m in range(2):
        o.regularization[m], o.regularizationScaled[m] = reg, lm
    clouds = [np.zeros((0, 4), dtype=F)] * 2
    pots = [np.ones(len(oms[m].host.sdfs) ** 2, dtype=np.int32) for m in range(2)]
    dims = [om.dims for om in oms]
    D = sum(dims)
    r = np.random.RandomState(7)
    mats = []
    for d in dims:
        B = r.standard_normal((d, d))
        mats.append((3.0 * (B @ B.T / d + np.eye(d))).astype(F))
    _, _, _, _, first0 = O.optimize_poses_noobs(oms, [T.copy() for T in T0], [j.copy() for j in j0], o, clouds, pots, priors)
    with O.damping_matrices(mats):
        oT, oj, _, _, first1 = O.optimize_poses_noobs(oms, [T.copy() for T in T0], [j.copy() for j in j0], o, clouds, pots, priors)
    # after the with-block the matrices are gone again
    _, _, _, _, first2 = O.optimize_poses_noobs(oms, [T.copy() for T in T0], [j.copy() for j in j0], o, clouds, pots, priors)
    assert np.array_equal(first0, first2)
    U0, U1 = first0[:D * D].reshape(D, D), first1[:D * D].reshape(D, D)
    expect = np.zeros((D, D), dtype=F)
    off = 0
    for d, M in zip(dims, mats):
        expect[off:off + d, off:off + d] = np.triu(M)
        off += d
    assert np.allclose(U1 - U0, expect, rtol=1e-4, atol=1e-3)
    assert np.array_equal(first0[D * D:], first1[D * D:])               # J^T e is untouched
    # the step the loop took is the back substitution of that damped system, regularised after the priors (:919-931)
    U = U1.copy()
    for i in range(D):
        U[i, i] += F(reg)
        U[i, i] += F(lm) * U[i, i]
    x = -solve_upper(U, first1[D * D:].copy())
    nj = oms[0].n_joints
    lim = np.asarray(sc.hand.joint_limits, dtype=F).reshape(-1, 2)
    j_new = np.clip(j0[0] + x[6:6 + nj].astype(F), lim[:, 0], lim[:, 1])
    T_new = O.se3_mul(O.se3_exp(x[0:6].astype(F)), T0[0])
    assert np.abs(oj[0] - j_new).max() < 1e-5 and np.abs(oT[0] - T_new).max() < 1e-5
    assert np.abs(oT[0] - T0[0]).max() > 1e-5
