"""Cartesian action front-end (IKPositionDeltaWrapper, wrappers/ik_position_delta_wrapper.py:93-142) on the CPU oracle.
pybullet's IK is absent here ([UPSTREAM]); what is pinned is what the wrapper promises: the end-effector link moves by the
clipped position delta within residual_threshold, its orientation stays the initial one, the gripper entry passes through."""
import ctypes

import numpy as np
import pytest

import human_robot_gym_amd as hrg
from oracle.oracle import OracleBatch, load

PP = dict(env_id="PickPlaceHumanCart")


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _fk(lib, d, q):
    p, R = np.zeros(3), np.zeros(9)
    lib.hrgo_test_ik_fk(ctypes.byref(d), _p(np.ascontiguousarray(q, np.float64)), _p(p), _p(R))
    return p, R.reshape(3, 3)


def test_ik_moves_the_end_effector_by_the_clipped_delta_at_fixed_orientation():
    lib = load()
    d = hrg.build_model_desc(dict(seed=1), ik_position_delta={}, **PP)
    q0 = np.array(list(d.init_qpos))
    p0, R0 = _fk(lib, d, q0)
    np.testing.assert_allclose(R0.reshape(-1), list(d.ik_target_rot), atol=1e-12)    # orientation target = pose at init_qpos
    rng = np.random.RandomState(0)
    for k in range(50):  # reachable targets (the start pose is 3 cm from full stretch: large outward deltas are out of reach)
        delta = rng.uniform(-0.06, 0.06, 3) if k else np.array([-0.2, 0.03, -0.2])
        act = np.array(list(delta) + [rng.uniform(-2, 2), 9, 9, 9])
        grip = np.clip(act[3], -1, 1)
        lib.hrgo_test_ik(ctypes.byref(d), _p(q0), _p(act))
        p1, R1 = _fk(lib, d, q0 + act[:6])
        np.testing.assert_allclose(p1 - p0, np.clip(delta, -0.15, 0.15), atol=1e-3 + 1e-9)   # residual_threshold
        assert np.abs(R1 - R0).max() < 5e-3
        assert act[6] == grip
    zero = np.zeros(7)
    lib.hrgo_test_ik(ctypes.byref(d), _p(q0), _p(zero))
    assert np.abs(zero[:6]).max() < 1e-12                                               # no delta, orientation on target: no motion


def test_ik_options_and_limits():
    lib = load()
    lim = [[0.2, -0.1, 1.0], [0.4, 0.1, 1.3]]
    d = hrg.build_model_desc(None, ik_position_delta=dict(action_limit=0.05, x_output_max=2, x_position_limits=lim, max_iter=100, residual_threshold=1e-4), **PP)
    assert d.ik_enabled == 1 and d.ik_max_iter == 100 and d.ik_use_pos_limits == 1 and d.ik_action_limit == 0.05
    q0 = np.array(list(d.init_qpos))
    p0, _ = _fk(lib, d, q0)
    act = np.array([0.2, 0.0, 0.0, 0, 0, 0, 0.0])                       # clipped to 0.05, scaled by x_output_max 2 -> 0.1, then the box limit
    lib.hrgo_test_ik(ctypes.byref(d), _p(q0), _p(act))
    p1, _ = _fk(lib, d, q0 + act[:6])
    want = np.clip(p0 + [0.1, 0, 0], lim[0], lim[1])
    np.testing.assert_allclose(p1, want, atol=2e-4)
    assert hrg.build_model_desc(None, **PP).ik_enabled == 0
    with pytest.raises(ValueError):
        hrg.build_model_desc(None, ik_position_delta=dict(bogus=1))


def test_env_step_converts_the_cartesian_action_before_collision_prevention():
    """Through env.step: the action row is rewritten to the joint action that was executed, and the arm follows."""
    clips = hrg.synthetic_clips(2, seed=0, min_frames=200, max_frames=300)
    kw = dict(shield_type="OFF", horizon=100, seed=2)
    d = hrg.build_model_desc(kw, n_clips=clips.n_clips, ik_position_delta={}, collision_prevention=dict(replace_type=0, n_resamples=20), **PP)
    B = OracleBatch(d, clips, 3)
    B.reset()
    e0 = np.array([list(B.get_state(e).eef_pos) for e in range(3)])
    for _ in range(15):
        a = np.zeros((3, 7))
        a[:, 0] = [0.15, -0.15, 0.0]
        a[:, 2] = [0.0, 0.0, 0.1]
        a[:, 3] = 1.0
        B.step(a)
        ex = B.last_actions
        assert np.all(ex[:, 6] == 1.0) and np.abs(ex[:, :6]).max() > 1e-3            # joint actions + gripper passed through
    e1 = np.array([list(B.get_state(e).eef_pos) for e in range(3)])
    dx = e1 - e0
    assert dx[0, 0] > 0.08 and abs(dx[0, 1]) < 0.03 and abs(dx[0, 2]) < 0.03             # +x
    assert dx[1, 0] < -0.08 and abs(dx[1, 1]) < 0.03
    assert dx[2, 2] > 0.05 and abs(dx[2, 0]) < 0.03
    B.close()


def test_vec_env_cartesian_action_space():
    from helpers import OracleBackend
    from human_robot_gym_amd.vec_env import HipVecEnv
    clips = hrg.synthetic_clips(2, seed=0, min_frames=200, max_frames=300)
    kw = dict(shield_type="SSM", horizon=20, seed=2)
    ik = dict(action_limit=0.15)
    desc = hrg.build_model_desc(kw, n_clips=clips.n_clips, ik_position_delta=ik, **PP)
    env = HipVecEnv(2, env_id="PickPlaceHumanCart", env_kwargs=kw, clips=clips, ik_position_delta=ik, backend=OracleBackend(desc, clips, 2))
    assert env.action_space.shape == (4,)
    np.testing.assert_allclose(env.action_space.high, [0.15, 0.15, 0.15, 1.0])
    env.reset()
    obs, rew, done, infos = env.step(np.array([[0.1, 0, 0, -1], [0, 0.1, 0, 1]]))
    assert obs.shape == (2, 11) and infos[0]["action"].shape == (7,) and infos[0]["action"][6] == -1 and infos[1]["action"][6] == 1
