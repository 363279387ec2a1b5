"""Static/self collision pre-check (HumanEnv.check_collision_action, human_env.py:588-627) and the
CollisionPreventionWrapper semantics (wrappers/collision_prevention_wrapper.py:38-103) folded into the stepper."""
import ctypes

import numpy as np
import pytest

import human_robot_gym_amd as hrg
from human_robot_gym_amd.vec_env import HipVecEnv
from helpers import OracleBackend, RTOL, assert_state_close, make_pair


def _collides(lib, d, q):
    q = np.ascontiguousarray(q, np.float64)
    return lib.hrgo_test_config_collides(ctypes.byref(d), q.ctypes.data_as(ctypes.c_void_p)) != 0


def test_precheck_known_configurations(oracle_lib):
    d = hrg.build_model_desc()
    assert not _collides(oracle_lib, d, np.zeros(6))                       # upright home pose is free
    assert _collides(oracle_lib, d, [0, 1.8, -0.8, 0, 0, 0])               # shoulder folded over, elbow down: link 4 in the table volume
    assert _collides(oracle_lib, d, [0, 0, 2.6, 0, 1.85, 0])               # elbow and wrist folded back: link 6 meets link 1
    lo, hi = np.array(d.qpos_limits[0][:]), np.array(d.qpos_limits[1][:])
    rng = np.random.RandomState(0)
    free = np.mean([not _collides(oracle_lib, d, lo + (hi - lo) * rng.rand(6)) for _ in range(500)])
    assert 0.5 < free < 0.95                                                # most of the joint box is reachable


def test_goals_are_collision_free_or_zero(oracle_lib):
    from oracle.oracle import OracleBatch
    clips = hrg.synthetic_clips(1, seed=0, min_frames=100, max_frames=120)
    d = hrg.build_model_desc(dict(horizon=10), n_clips=1)
    B = OracleBatch(d, clips, 32)
    B.reset()
    for e in range(32):
        g = np.array(B.get_state(e).cur_goal)
        assert np.all(g == 0) or not _collides(oracle_lib, d, g)          # reach_human_env.py:535-546
    d2 = hrg.build_model_desc(dict(horizon=10), n_clips=1, goal_check=False)
    B2 = OracleBatch(d2, clips, 32)
    B2.reset()
    assert any(_collides(oracle_lib, d2, np.array(B2.get_state(e).cur_goal)) for e in range(32))


@pytest.mark.parametrize("replace_type", [0, 1, 2])
def test_wrapper_semantics_on_the_oracle_backend(oracle_lib, replace_type):
    """action replaced <=> the goal configuration of the ORIGINAL action fails the pre-check; type 0 -> zero action;
    types 1/2 -> a sampled action whose goal passes the pre-check (or zero); the counter counts replacements."""
    kw = dict(shield_type="OFF", horizon=80)
    cp = dict(replace_type=replace_type, n_resamples=20)
    clips = hrg.synthetic_clips(1, seed=0, min_frames=200, max_frames=300)
    desc = hrg.build_model_desc(kw, n_clips=1, collision_prevention=cp)
    env = HipVecEnv(6, env_kwargs=kw, clips=clips, backend=OracleBackend(desc, clips, 6), collision_prevention=cp)
    env.reset()
    B = env._backend.B
    lo, hi = np.array(desc.qpos_limits[0][:]), np.array(desc.qpos_limits[1][:])
    goal_of = lambda q, a: np.clip(q + 0.2 * np.clip(a[:6], -1, 1), lo, hi)  # noqa: E731  failsafe.json output range
    rng = np.random.RandomState(0)
    replaced, count = 0, np.zeros(6, int)
    for k in range(80):
        a = rng.uniform(-1, 1, (6, 7))
        a[:, 1] = 1.0                                                       # keep folding the shoulder towards the table (the joints of a trajectory arrive
                                                                            # together, so the fold advances at the pace of the slowest of the random moves)
        q0 = [np.array(B.get_state(i).qpos[:6]) for i in range(6)]
        obs, rew, done, infos = env.step(a.copy())
        for i in range(6):
            bad = _collides(oracle_lib, desc, goal_of(q0[i], a[i]))
            ex = infos[i]["action"]
            assert bad == (not np.array_equal(ex, a[i]))
            if bad:
                replaced += 1
                count[i] += 1
                if replace_type == 0:
                    assert np.all(ex == 0)                                  # collision_prevention_wrapper.py:53-61
                else:
                    assert np.all(np.abs(ex) <= 1) and (np.all(ex == 0) or not _collides(oracle_lib, desc, goal_of(q0[i], ex)))
            if done[i]:
                count[i] = 0                                                # wrapper.reset(): action_resamples = 0
            else:
                assert infos[i]["action_resamples"] == count[i]
    assert replaced > 0
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("replace_type", [0, 1, 2])
def test_hip_matches_oracle_with_collision_prevention(replace_type):
    import torch
    kw = dict(shield_type="OFF", horizon=90)
    O, G = make_pair(16, kw, collision_prevention=dict(replace_type=replace_type, n_resamples=12))
    np.testing.assert_allclose(G.reset().cpu().numpy(), O.reset(), rtol=RTOL, atol=1e-6)
    rng = np.random.RandomState(3)
    tot = 0
    for k in range(90):
        a = rng.uniform(-1, 1, (16, 7))
        a[:, 1] = np.where(np.arange(16) % 4 != 3, 1.0, a[:, 1])  # most envs keep folding the shoulder towards the table (at the pace of the slowest joint: the
                                                                  # joints of a trajectory arrive together)
        ag = torch.from_numpy(a.copy()).cuda()
        o_o, r_o, d_o, i_o = O.step(a)
        o_g, r_g, d_g, i_g = G.step(ag)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(i_g.cpu().numpy(), i_o, err_msg=f"step {k}")
        np.testing.assert_array_equal(ag.cpu().numpy(), O.last_actions, err_msg=f"executed actions, step {k}")  # bit-exact: same hash
        np.testing.assert_allclose(o_g.cpu().numpy(), o_o, rtol=RTOL, atol=1e-6)
        for e in range(16):
            so = O.get_state(e)
            assert_state_close(so, G.get_state(e), f"step {k} env {e}")
            G.set_state(e, so)
        tot += int(i_o[:, 12].max())
    assert tot > 0
    O.close(); G.close()


@pytest.mark.gpu
def test_batched_check_collision_action_matches_the_oracle():
    """hrg_batch_check_actions = HumanEnv.check_collision_action (human_env.py:588-627) for every env at its current joint angles."""
    import torch
    from helpers import make_pair
    from human_robot_gym_amd.vec_env import HipVecEnv
    O, G = make_pair(256, dict(shield_type="OFF", horizon=50, seed=3))
    O.reset(); G.reset()
    rng = np.random.RandomState(0)
    for e in range(0, 256, 2):                                    # every other env close to the folded-over posture that meets the table
        st = O.get_state(e)
        for j, q in enumerate([0.0, 1.65, -0.75, 0.0, 0.0, 0.0] + rng.uniform(-0.1, 0.1, 6) * np.array([1, 1, 1, 1, 1, 1])):
            st.qpos[j] = float(q)
        O.set_state(e, st); G.set_state(e, st)
    hits = 0
    for k in range(6):
        probe = rng.uniform(-1, 1, (256, 7)) * 3.0                # (clipped to [-1, 1] by the controller's input scaling)
        co, cg = O.check_actions(probe), G.check_actions(torch.from_numpy(probe).cuda()).cpu().numpy()
        np.testing.assert_array_equal(cg, co, err_msg=f"step {k}")
        hits += int(co.sum())
    assert 100 < hits < 6 * 200                                   # goal configurations near the table collide, the ones from the initial posture do not
    O.close(); G.close()
    env = HipVecEnv(8, env_kwargs=dict(seed=1))
    env.reset()
    acts = rng.uniform(-1, 1, (8, 7))
    flags = env.check_collision_action(acts)
    assert flags.dtype == bool and flags.shape == (8,)
    assert env.env_method("check_collision_action", acts[3], indices=[3]) == [bool(flags[3])]
    env.close()
