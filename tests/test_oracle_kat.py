"""Known-answer / invariant tests of the CPU oracle's pieces (no GPU).  The reference holds no golden vectors for this
path, so the oracle is pinned by analytic properties and by independent numpy restatements."""
import ctypes

import numpy as np
import pytest

import human_robot_gym_amd as hrg
from human_robot_gym_amd._cstruct import CONST, LTT
from human_robot_gym_amd.model import robot_fk_numpy

NV, NARM = CONST["HRG_NV"], CONST["HRG_NARM"]


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _robot(lib, d, q, qd):
    M, b, eef = np.zeros((NV, NV)), np.zeros(NV), np.zeros(3)
    lib.hrgo_test_robot(ctypes.byref(d), _p(np.ascontiguousarray(q)), _p(np.ascontiguousarray(qd)), _p(M), _p(b), _p(eef))
    return M, b, eef


def _numpy_M_and_V(d, q):
    """Independent restatement: M = sum_b m Jv'Jv + Jw' I Jw (+ armature), V = sum_b m g.com."""
    R, p = robot_fk_numpy(d, q)
    M, V = np.zeros((NV, NV)), 0.0
    for b in range(NV):
        com = p[b] + R[b] @ np.asarray(d.body_com[b][:])
        Jv, Jw = np.zeros((3, NV)), np.zeros((3, NV))
        k = b
        while k >= 0:
            ax = R[k] @ np.asarray(d.jnt_axis[k][:])
            if d.jnt_type[k] == 0:
                Jw[:, k] = ax
                Jv[:, k] = np.cross(ax, com - p[k])
            else:
                Jv[:, k] = ax
            k = d.body_parent[k]
        I = np.asarray(d.body_inertia[b][:])
        Ib = np.array([[I[0], I[3], I[4]], [I[3], I[1], I[5]], [I[4], I[5], I[2]]])
        M += d.body_mass[b] * Jv.T @ Jv + Jw.T @ (R[b] @ Ib @ R[b].T) @ Jw
        V += -d.body_mass[b] * np.dot(np.asarray(d.gravity[:]), com)
    return M + np.diag([d.jnt_armature[i] for i in range(NV)]), V


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_crba_matches_jacobian_form(oracle_lib, seed):
    d = hrg.build_model_desc()
    rng = np.random.RandomState(seed)
    q = np.concatenate([rng.uniform(-2, 2, NARM), rng.uniform(-0.01, 0.02, 2)])
    M, _, _ = _robot(oracle_lib, d, q, np.zeros(NV))
    Mn, _ = _numpy_M_and_V(d, q)
    np.testing.assert_allclose(M, M.T, atol=1e-13)
    assert np.linalg.eigvalsh(M).min() > 0
    np.testing.assert_allclose(M, Mn, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("seed", [0, 1])
def test_bias_gravity_is_potential_gradient(oracle_lib, seed):
    d = hrg.build_model_desc()
    rng = np.random.RandomState(seed)
    q = np.concatenate([rng.uniform(-2, 2, NARM), rng.uniform(-0.01, 0.02, 2)])
    _, g, _ = _robot(oracle_lib, d, q, np.zeros(NV))
    eps = 1e-6
    gn = np.zeros(NV)
    for i in range(NV):
        dq = np.zeros(NV); dq[i] = eps
        gn[i] = (_numpy_M_and_V(d, q + dq)[1] - _numpy_M_and_V(d, q - dq)[1]) / (2 * eps)
    np.testing.assert_allclose(g, gn, rtol=1e-6, atol=1e-7)


def test_bias_coriolis_power_balance(oracle_lib):
    """qd'(c - g) = 1/2 qd' (dM/dt) qd  (passivity of the Coriolis term)."""
    d = hrg.build_model_desc()
    rng = np.random.RandomState(3)
    q = np.concatenate([rng.uniform(-2, 2, NARM), [0.0, 0.0]])
    qd = np.concatenate([rng.uniform(-1, 1, NARM), rng.uniform(-0.05, 0.05, 2)])
    _, c, _ = _robot(oracle_lib, d, q, qd)
    _, g, _ = _robot(oracle_lib, d, q, np.zeros(NV))
    eps = 1e-6
    Mp, _, _ = _robot(oracle_lib, d, q + eps * qd, np.zeros(NV))
    Mm, _, _ = _robot(oracle_lib, d, q - eps * qd, np.zeros(NV))
    lhs = qd @ (c - g)
    rhs = 0.5 * qd @ ((Mp - Mm) / (2 * eps)) @ qd
    assert abs(lhs - rhs) < 1e-6 * (1 + abs(rhs))


def test_pendulum_closed_form(oracle_lib):
    """Single link about a horizontal axis: M = I + m l^2, bias = m g l sin(theta) — the Schunk joint-2 column
    with every other joint locked reproduces the rigid-pendulum formula of the distal composite body."""
    d = hrg.build_model_desc()
    q = np.zeros(NV); q[1] = 0.7
    M, g, _ = _robot(oracle_lib, d, q, np.zeros(NV))
    R, p = robot_fk_numpy(d, q)
    ax = R[1] @ np.asarray(d.jnt_axis[1][:])
    I, tau = 0.0, 0.0
    for b in range(1, NV):
        com = p[b] + R[b] @ np.asarray(d.body_com[b][:])
        r = com - p[1]
        rp = r - ax * (ax @ r)
        Ib = np.asarray(d.body_inertia[b][:])
        Iw = R[b] @ np.array([[Ib[0], Ib[3], Ib[4]], [Ib[3], Ib[1], Ib[5]], [Ib[4], Ib[5], Ib[2]]]) @ R[b].T
        I += ax @ Iw @ ax + d.body_mass[b] * (rp @ rp)
        tau += -ax @ np.cross(r, d.body_mass[b] * np.asarray(d.gravity[:]))
    assert abs(M[1, 1] - I) < 1e-10
    assert abs(g[1] - tau) < 1e-10


def test_segment_segment_against_brute_force(oracle_lib):
    rng = np.random.RandomState(0)
    t = np.linspace(0, 1, 201)
    for k in range(200):
        P = rng.uniform(-1, 1, (4, 3))
        if k % 10 == 0:
            P[1] = P[0]  # degenerate: point
        if k % 15 == 0:
            P[3] = P[2]
        c1, c2 = np.zeros(3), np.zeros(3)
        d2 = oracle_lib.hrgo_test_segseg(_p(P[0].copy()), _p(P[1].copy()), _p(P[2].copy()), _p(P[3].copy()), _p(c1), _p(c2))
        A = P[0] + t[:, None] * (P[1] - P[0])
        B = P[2] + t[:, None] * (P[3] - P[2])
        bf = ((A[:, None, :] - B[None, :, :]) ** 2).sum(-1).min()
        assert d2 <= bf + 1e-12
        assert d2 >= bf - 2e-2 * (1 + bf)  # the grid can only overestimate
        assert abs(np.sum((c1 - c2) ** 2) - d2) < 1e-12


def test_segment_segment_known_values(oracle_lib):
    c1, c2 = np.zeros(3), np.zeros(3)
    f = lambda *v: _p(np.array(v, float))  # noqa: E731
    assert oracle_lib.hrgo_test_segseg(f(0, 0, 0), f(1, 0, 0), f(0, 1, 0), f(1, 1, 0), _p(c1), _p(c2)) == pytest.approx(1.0)
    assert oracle_lib.hrgo_test_segseg(f(0, 0, 0), f(1, 0, 0), f(0.5, -1, 0.3), f(0.5, 1, 0.3), _p(c1), _p(c2)) == pytest.approx(0.09)
    assert oracle_lib.hrgo_test_segseg(f(0, 0, 0), f(1, 0, 0), f(2, 0, 0), f(3, 0, 0), _p(c1), _p(c2)) == pytest.approx(1.0)


@pytest.mark.parametrize("seed", range(5))
def test_long_term_trajectory_end_conditions_and_limits(oracle_lib, seed):
    d = hrg.build_model_desc()
    rng = np.random.RandomState(seed)
    q0 = rng.uniform(-2, 2, NARM)
    v0 = rng.uniform(-1, 1, NARM) * (seed > 0)
    a0 = rng.uniform(-2, 2, NARM) * (seed > 1)
    goal = rng.uniform(-2.5, 2.5, NARM)
    L = LTT()
    oracle_lib.hrgo_test_ltt(ctypes.byref(d), _p(q0), _p(v0), _p(a0), _p(goal), ctypes.byref(L))
    out = np.zeros(3)
    for j in range(NARM):
        T = sum(L.dur[j][:])
        assert T <= L.T + 1e-12 and all(x >= 0 for x in L.dur[j][:])
        oracle_lib.hrgo_test_ltt_eval(ctypes.byref(L), j, 0.0, _p(out))
        np.testing.assert_allclose(out, [q0[j], v0[j], a0[j]], atol=1e-12)
        oracle_lib.hrgo_test_ltt_eval(ctypes.byref(L), j, T - 1e-9, _p(out))
        np.testing.assert_allclose(out, [goal[j], 0, 0], atol=1e-6)
        prev = None
        for s in np.linspace(0, T, 400):
            oracle_lib.hrgo_test_ltt_eval(ctypes.byref(L), j, float(s), _p(out))
            assert abs(out[2]) <= max(d.a_max_ltt[j], abs(a0[j])) + 1e-9
            assert abs(out[1]) <= max(d.v_max_ltt[j], abs(v0[j]) + a0[j] ** 2 / (2 * d.j_max_ltt[j])) + 1e-9
            if prev is not None:  # C1 continuity of the position
                assert abs(out[0] - prev[0]) < (abs(out[1]) + abs(prev[1]) + 1.0) * T / 399
            prev = out.copy()


@pytest.mark.parametrize("seed", range(4))
def test_long_term_trajectory_is_time_synchronised(oracle_lib, seed):
    """sara-shield's LongTermPlanner synchronises the joints of a long-term trajectory to the slowest one: every joint that moves arrives at its goal
    at the same instant L.T (here: within 1e-9 s, far inside one 4 ms sample) -- from rest, with initial velocities, with initial accelerations; a joint already at
    its goal stays there.  Without the synchronisation (shield_params ltt_time_sync=False, the planner of rounds 1-2) the joints arrive one by one."""
    rng = np.random.RandomState(100 + seed)
    for trial in range(50):
        q0 = rng.uniform(-2, 2, NARM)
        v0 = rng.uniform(-1, 1, NARM) * (seed > 0)
        a0 = rng.uniform(-2, 2, NARM) * (seed > 1)
        goal = rng.uniform(-2.5, 2.5, NARM)
        if seed == 3:
            goal[2] = q0[2]; v0[2] = a0[2] = 0.0     # a joint that is where it has to be
        ends = {}
        for sync in (True, False):
            d = hrg.build_model_desc(shield_params=dict(ltt_time_sync=sync))
            L = LTT()
            oracle_lib.hrgo_test_ltt(ctypes.byref(d), _p(q0), _p(v0), _p(a0), _p(goal), ctypes.byref(L))
            ends[sync] = (np.array([sum(L.dur[j][:]) for j in range(NARM)]), L.T)
            out = np.zeros(3)
            for j in range(NARM):
                assert all(x >= 0 for x in L.dur[j][:])
                oracle_lib.hrgo_test_ltt_eval(ctypes.byref(L), j, float(L.T), _p(out))
                np.testing.assert_allclose(out, [goal[j], 0, 0], atol=1e-9)
                for sx in np.linspace(0, L.T, 60):   # the stretched profiles stay inside the planner's limits
                    oracle_lib.hrgo_test_ltt_eval(ctypes.byref(L), j, float(sx), _p(out))
                    assert abs(out[2]) <= max(d.a_max_ltt[j], abs(a0[j])) + 1e-9
                    assert abs(out[1]) <= max(d.v_max_ltt[j], abs(v0[j]) + a0[j] ** 2 / (2 * d.j_max_ltt[j])) + 1e-9
        (t_sync, T_sync), (t_own, T_own) = ends[True], ends[False]
        assert T_sync == pytest.approx(T_own, abs=1e-12)                       # the slowest joint sets the duration either way
        moving = t_own > 0
        assert np.all(np.abs(t_sync[moving] - T_sync) < 1e-9), (t_sync, T_sync)
        assert np.all(t_sync[~moving] == 0)
        assert t_own[moving].max() - t_own[moving].min() > 1e-3                # (unsynchronised: they do not)


@pytest.mark.parametrize("v0,a0,ve", [(1.0, 0.0, 0.0), (0.0, 0.0, 1.0), (0.6, 3.0, 0.0), (0.4, -5.0, 1.0), (0.9, 7.9, 1.0), (0.0, 0.0, 0.0)])
def test_path_profile_reaches_target_velocity(oracle_lib, v0, a0, ve):
    d = hrg.build_model_desc()
    out = np.zeros(4)
    oracle_lib.hrgo_test_path(0.0, v0, a0, ve, d.path_amax, d.path_jmax, 1e9, _p(out))
    T = out[3]
    oracle_lib.hrgo_test_path(0.0, v0, a0, ve, d.path_amax, d.path_jmax, max(T - 1e-12, 0.0), _p(out))
    assert abs(out[1] - ve) < 1e-8 and abs(out[2]) < 1e-6
    last = -1e9
    for t in np.linspace(0, T, 50):  # s never runs backwards while braking / recovering within [0, 1]
        oracle_lib.hrgo_test_path(0.0, v0, a0, ve, d.path_amax, d.path_jmax, float(t), _p(out))
        assert out[0] >= last - 1e-12
        assert abs(out[2]) <= max(d.path_amax, abs(a0)) + 1e-9
        last = out[0]


def test_counter_rng_is_deterministic_and_uniform(oracle_lib):
    u = np.array([oracle_lib.hrgo_test_u01(7, 3, 2, 1, i) for i in range(4000)])
    assert u.min() >= 0 and u.max() < 1
    assert abs(u.mean() - 0.5) < 0.02 and abs(u.var() - 1 / 12) < 0.01
    assert oracle_lib.hrgo_test_u01(7, 3, 2, 1, 5) == oracle_lib.hrgo_test_u01(7, 3, 2, 1, 5)
    assert oracle_lib.hrgo_test_u01(7, 3, 2, 1, 5) != oracle_lib.hrgo_test_u01(7, 4, 2, 1, 5)
    # pinned values (integer hash -> exact doubles on every platform)
    np.testing.assert_array_equal(u[:3], np.load(__file__.replace("test_oracle_kat.py", "golden/rng_u01.npy")))


def test_human_kinematics_rest_pose_and_bone_lengths(oracle_lib):
    d = hrg.build_model_desc()
    nhj, nhb = CONST["HRG_NHJ"], CONST["HRG_NHB"]
    sites, caps = np.zeros((nhj, 3)), np.zeros((nhb, 6))
    mp, mq = np.array([0.3, -0.2, 0.1]), np.array([1.0, 0, 0, 0])
    oracle_lib.hrgo_test_human_fk(ctypes.byref(d), _p(mp), _p(mq), _p(np.zeros(69)), _p(sites), _p(caps))
    for k in range(nhj):
        np.testing.assert_allclose(sites[k], mp + np.asarray(d.hb_anchor[d.meas_body[k]][:]), atol=1e-14)
    rng = np.random.RandomState(0)
    qh = rng.uniform(-1, 1, 69)
    s2 = np.zeros((nhj, 3))
    yaw = 0.7
    oracle_lib.hrgo_test_human_fk(ctypes.byref(d), _p(mp), _p(np.array([np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)])), _p(qh), _p(s2), _p(caps))
    body_of = {d.meas_body[k]: k for k in range(nhj)}
    for k in range(nhj):  # distance between a joint and its parent joint is pose-invariant
        par = d.hb_parent[d.meas_body[k]]
        if par in body_of:
            assert abs(np.linalg.norm(s2[k] - s2[body_of[par]]) - np.linalg.norm(sites[k] - sites[body_of[par]])) < 1e-12


def test_study_free_human_joint_dynamics_diverges(oracle_lib):
    """restating the 69 human hinges as free dynamic DoFs whose qpos is overwritten every substep
    (human_env.py:1766-1767; unit masses/inertias, armature 0.01, no damping: human.xml:4,46) makes the joint velocities
    run away within a few simulated seconds even for a static T-pose — articulated-body dynamics under gravity with the
    configuration pinned.  The reference's episodes last 10-40 s, so this cannot be what its MuJoCo run looks like; the
    human is therefore kept kinematic until the real engine's behaviour can be observed."""
    d = hrg.build_model_desc()
    F = np.ascontiguousarray(hrg.static_clip(100).frames)
    n = 1000
    mv, ma = np.zeros(n), np.zeros(n)
    oracle_lib.hrgo_test_human_dyn(ctypes.byref(d), _p(F), ctypes.c_int(len(F)), ctypes.c_int(n), ctypes.c_double(0.01), _p(mv), _p(ma))
    assert 5.0 < ma[0] < 20.0            # gravity alone: ~12 rad/s^2 on the worst joint of the T-pose
    assert mv[250] > 8.0                 # 1 s: > 8 rad/s
    assert mv[:750].max() > 40.0         # within 3 s: > 40 rad/s somewhere in the tree
    assert not (mv[-1] < 1e6)            # 4 s: diverged (huge or NaN)
