"""VecEnv / gym facade semantics (SB3 1.5.0 VecEnv contract, Monitor, TimeLimit) — host logic, backed by the oracle."""
import numpy as np
import pytest

import human_robot_gym_amd as hrg
from human_robot_gym_amd.vec_env import HipGymEnv, HipVecEnv, INFO_KEYS
from helpers import OracleBackend


def _vec(n, kw, **extra):
    clips = hrg.synthetic_clips(2, seed=0, min_frames=200, max_frames=300)
    desc = hrg.build_model_desc(kw, n_clips=clips.n_clips)
    return HipVecEnv(n, env_kwargs=kw, clips=clips, backend=OracleBackend(desc, clips, n), **extra)


def test_vec_env_contract_and_autoreset():
    kw = dict(shield_type="OFF", horizon=5, reward_shaping=True)
    env = _vec(4, kw)
    assert env.num_envs == 4 and env.observation_space.shape == (18,) and env.action_space.shape == (7,)
    obs = env.reset()
    assert obs.shape == (4, 18) and obs.dtype == np.float32  # default obs_keys: object-state + goal_difference
    rng = np.random.RandomState(0)
    rets = np.zeros(4)
    for k in range(5):
        obs, rew, done, infos = env.step(rng.uniform(-1, 1, (4, 7)))
        rets += rew
        assert rew.shape == (4,) and done.dtype == bool and len(infos) == 4
        for key in ["n_goal_reached", "collision", "collision_type", "n_collisions", "n_collisions_static", "n_collisions_robot",
                    "n_collisions_human", "n_collisions_critical", "timeout", "failsafe_interventions", "action_resamples"]:
            assert key in infos[0]                      # training/config/run/default_training.yaml:18-29
        if k < 4:
            assert not done.any() and "terminal_observation" not in infos[0] and "TimeLimit.truncated" not in infos[0]
    assert done.all()                                   # TimeLimit: elapsed >= horizon (time_limit.py:40-43)
    for i in range(4):
        assert infos[i]["TimeLimit.truncated"] is True and infos[i]["timeout"] is True
        assert infos[i]["terminal_observation"].shape == (18,)
        assert infos[i]["episode"]["l"] == 5 and infos[i]["episode"]["r"] == pytest.approx(rets[i], rel=1e-6)
        assert not np.allclose(infos[i]["terminal_observation"], obs[i])   # obs is already the next episode's first obs
    obs2, _, done2, infos2 = env.step(rng.uniform(-1, 1, (4, 7)))
    assert not done2.any() and "episode" not in infos2[0]
    env.close()


def test_gym_env_four_tuple_and_stepping_finished_episode():
    kw = dict(shield_type="SSM", horizon=3)
    clips = hrg.synthetic_clips(1, seed=1, min_frames=200, max_frames=300)
    desc = hrg.build_model_desc(kw, n_clips=1)
    env = HipGymEnv(env_kwargs=kw, clips=clips, backend=OracleBackend(desc, clips, 1))
    with pytest.raises(ValueError):
        env.step(np.zeros(7))                           # human_env.py:487-488
    obs = env.reset()
    assert obs.shape == (18,)
    for k in range(3):
        obs, r, done, info = env.step(env.action_space.sample())
        assert isinstance(r, float) and isinstance(done, bool) and isinstance(info, dict)
    assert done and info["TimeLimit.truncated"]
    with pytest.raises(ValueError):
        env.step(np.zeros(7))
    d = env.observation_dict(env.reset())
    assert d["object-state"].shape == (12,) and d["goal_difference"].shape == (6,)
    env.close()


def test_unsupported_configurations_fail_loudly():
    from human_robot_gym_amd.env_util import make_vec_env
    with pytest.raises(NotImplementedError):
        HipVecEnv(2, env_id="PickPlaceHumanTeleop", backend=object())
    with pytest.raises(NotImplementedError):
        HipVecEnv(2, obs_keys=["robot0_eef_quat"], backend=object())
    with pytest.raises(NotImplementedError):
        make_vec_env("HumanObjectInspectionCart", type="goal_env", vec_env_kwargs=dict(backend=object()))
    with pytest.raises(AssertionError):
        make_vec_env("ReachHuman", type="bogus")
    assert INFO_KEYS[8] == "failsafe_interventions"


def test_reward_and_success_logic_against_closed_form():
    """reward = scale*([goal? task : -1] + [shaping? 1 - 0.1*||q-goal|| : 0]) (human_env.py:629-664, reach_human_env.py:437-475)."""
    kw = dict(shield_type="OFF", horizon=50, reward_shaping=True, reward_scale=2.0, task_reward=3.0, done_at_success=True, goal_dist=0.1)
    env = _vec(8, kw)
    env.reset()
    B = env._backend.B
    rng = np.random.RandomState(1)
    for _ in range(10):
        obs, rew, done, infos = env.step(rng.uniform(-1, 1, (8, 7)))
        for i in range(8):
            o = infos[i]["terminal_observation"] if done[i] else obs[i]
            dist = float(np.linalg.norm(o[12:].astype(np.float64)))
            reached = infos[i]["n_goal_reached"] > 0
            expect = 2.0 * ((3.0 if reached else -1.0) + 1.0 - 0.1 * dist)
            assert rew[i] == pytest.approx(expect, rel=1e-5, abs=1e-5)
            assert done[i] == reached
    assert B is not None
    env.close()


def test_obs_keys_select_and_order_columns_like_the_gym_wrapper():
    """GymWrapper(env, keys=obs_keys) concatenates obs[key] in key order (utils/env_util.py:40-51); the ICRA "R" configs use
    obs_keys=[goal_difference] (config_icra_2024/.../R-SAC.yaml:160-161)."""
    kw = dict(shield_type="OFF", horizon=20)
    full = _vec(3, kw, obs_keys=["object-state", "goal_difference", "robot0_proprio-state", "desired_goal"])
    icra = _vec(3, kw, obs_keys=["goal_difference"])
    mixed = _vec(3, kw, obs_keys=["robot0_eef_pos", "dist_eef_to_human_head", "goal-state"])
    assert full.observation_space.shape == (39,) and icra.observation_space.shape == (6,) and mixed.observation_space.shape == (16,)
    of, oi, om = full.reset(), icra.reset(), mixed.reset()
    a = np.random.RandomState(0).uniform(-1, 1, (3, 7))
    for _ in range(3):
        of, _, _, _ = full.step(a); oi, _, _, _ = icra.step(a); om, _, _, _ = mixed.step(a)
    np.testing.assert_array_equal(oi, of[:, 12:18])
    np.testing.assert_array_equal(om[:, :3], of[:, 30:33])            # robot0_eef_pos
    np.testing.assert_array_equal(om[:, 3], of[:, 11])                # dist_eef_to_human_head
    np.testing.assert_array_equal(om[:, 4:10], of[:, 33:39])          # goal-state = desired_goal, goal_difference
    np.testing.assert_array_equal(om[:, 10:16], of[:, 12:18])
    np.testing.assert_allclose(of[:, 33:39] - of[:, 18:24], of[:, 12:18], atol=1e-6)   # goal - q = goal_difference
    np.testing.assert_allclose(np.linalg.norm(of[:, 0:3], axis=1), of[:, 3], rtol=1e-6)
    for e in (full, icra, mixed):
        e.close()


def test_expert_observation_side_channel():
    """ExpertObsWrapper (wrappers/expert_obs_wrapper.py:155-184): infos carry previous / current expert observations as dicts;
    after an auto-reset the next step's "previous" is the new episode's first observation."""
    from human_robot_gym_amd.env_util import make_vec_env
    clips = hrg.synthetic_clips(2, seed=0, min_frames=200, max_frames=300)
    kw = dict(shield_type="OFF", horizon=3, seed=4)
    keys = ["object_gripped", "vec_eef_to_object", "vec_eef_to_target", "robot0_gripper_qpos"]   # PickPlaceHumanCartExpertObservation
    desc = hrg.build_model_desc(kw, n_clips=clips.n_clips, env_id="PickPlaceHumanCart")
    back = OracleBackend(desc, clips, 2)
    env = make_vec_env("PickPlaceHumanCart", type="env", expert_obs_keys=keys, n_envs=2, env_kwargs=kw, vec_env_kwargs=dict(clips=clips, backend=back))
    env.reset()
    first = back.B.obs.copy()
    last_cur = None
    for k in range(4):
        obs, rew, done, infos = env.step(np.zeros((2, 7)))
        prev, cur = infos[0]["previous_expert_observation"], infos[0]["current_expert_observation"]
        assert set(prev) == set(keys) and prev["vec_eef_to_object"].shape == (3,) and cur["robot0_gripper_qpos"].shape == (2,)
        if k == 0:
            np.testing.assert_array_equal(prev["robot0_gripper_qpos"], first[0, 53:55])
        elif k < 3:
            np.testing.assert_array_equal(prev["vec_eef_to_object"], last_cur["vec_eef_to_object"])
        if k == 2:
            assert done.all()
            np.testing.assert_array_equal(cur["vec_eef_to_target"], back.B.term_obs[0, 43:46])   # terminal step's own observation
            after_reset = back.B.obs.copy()
        if k == 3:
            np.testing.assert_array_equal(prev["vec_eef_to_object"], after_reset[0, 40:43])
        last_cur = cur


@pytest.mark.parametrize("env_id", ["ReachHuman", "PickPlaceHumanCart"])
@pytest.mark.parametrize("shaping", [False, True])
def test_goal_env_type_serves_her(env_id, shaping):
    """make_vec_env(type="goal_env") = GoalEnvironmentGymWrapper (wrappers/goal_env_wrapper.py): dict observations and the invariant
    reward == compute_reward(achieved_goal, desired_goal, info) (goal_env_wrapper.py:172-190), also after relabelling."""
    from human_robot_gym_amd.env_util import make_vec_env
    clips = hrg.synthetic_clips(2, seed=0, min_frames=200, max_frames=300)
    kw = dict(shield_type="OFF", horizon=12, seed=6, reward_shaping=shaping, collision_reward=-3.0, goal_dist=0.5)
    desc = hrg.build_model_desc(kw, n_clips=clips.n_clips, env_id=env_id)
    env = make_vec_env(env_id, type="goal_env", n_envs=6, env_kwargs=kw, vec_env_kwargs=dict(clips=clips, backend=OracleBackend(desc, clips, 6)))
    gdim, adim = (6, 6) if env_id == "ReachHuman" else (3, 7)
    obs = env.reset()
    assert set(obs) == {"observation", "achieved_goal", "desired_goal"}
    assert obs["desired_goal"].shape == (6, gdim) and obs["achieved_goal"].shape == (6, adim)
    assert obs["observation"].shape == (6, 12 + 15 + gdim)                       # object-state, robot0_proprio-state, desired_goal
    assert env.observation_space["desired_goal"].shape == (gdim,)
    rng = np.random.RandomState(0)
    for k in range(12):
        prev = obs
        obs, rew, done, infos = env.step(rng.uniform(-1, 1, (6, 7)))
        term = {key: np.stack([infos[i]["terminal_observation"][key] if done[i] else obs[key][i] for i in range(6)]) for key in obs}
        r2 = env.compute_reward(term["achieved_goal"], term["desired_goal"], infos)
        np.testing.assert_allclose(r2, rew, rtol=1e-5, atol=1e-6)
        assert env.env_method("compute_reward", term["achieved_goal"][:2], term["desired_goal"][:2], infos[:2], indices=[0])[0].shape == (2,)
    # hindsight relabelling: the achieved goal as the desired one is a success
    ag = term["achieved_goal"]
    relabeled = ag if env_id == "ReachHuman" else ag[:, 3:6]
    r3 = env.compute_reward(ag, relabeled, [dict(collision_type=0)] * 6)
    assert np.allclose(r3, 1.0 + (1.0 + (0.0 if env_id == "ReachHuman" else -0.02 * np.linalg.norm(ag[:, 3:6] - ag[:, :3], axis=1)) if shaping else 0.0), atol=1e-6)
    assert isinstance(env.compute_reward(ag[0], relabeled[0], dict(collision_type=8)), float)
    assert abs(env.compute_reward(ag[0], relabeled[0], dict(collision_type=8)) - (r3[0] - 3.0)) < 1e-6   # static collision penalty


def test_lazy_info_dicts_behave_like_dicts():
    """LazyInfo: SB3's per-step lookups do not materialise a row; any other use does, and copies / pickles are plain dicts."""
    import copy
    import pickle
    from human_robot_gym_amd.vec_env import INFO_KEYS, LazyInfo, _InfoSource
    rows = np.array([[i] * len(INFO_KEYS) for i in range(3)], np.int32)
    src = _InfoSource(rows, np.arange(21.0).reshape(3, 7), None, None, None)

    def mk(i):
        d = LazyInfo.__new__(LazyInfo)
        d._src, d._i = src, i
        return d
    d = mk(1)
    assert d.get("episode") is None and d.get("TimeLimit.truncated", False) is False and "terminal_observation" not in d
    assert d._src is not None                                   # still lazy after the lookups SB3 does on every info
    assert d["n_goal_reached"] == 1 and d["collision"] is True and d._src is None
    assert len(mk(2)) == len(INFO_KEYS) and "TimeLimit.truncated" not in mk(2)   # (+ action, - TimeLimit.truncated)
    for plain in (dict(mk(2)), {**mk(2)}, copy.deepcopy(mk(2)), pickle.loads(pickle.dumps(mk(2))), mk(2).copy()):
        assert type(plain) is dict and plain["n_collisions"] == 2 and plain["sim_crash"] is True and plain["action"][0] == 14.0
    assert set(mk(1).keys()) == (set(INFO_KEYS) - {"TimeLimit.truncated"}) | {"action"}
    d = mk(0)
    d["x"] = 5
    assert d["x"] == 5 and d["timeout"] is False and "collision_type" in repr(mk(0))
    assert [k for k in mk(1)][:2] == INFO_KEYS[:2]


def test_every_observation_key_of_the_reference_configs_is_served():
    """`obs_keys` / `expert_obs_keys` of every yaml under human_robot_gym/training/config (enumerated once with a yaml walk over the reference tree; the list is
    spelled out here because the reference does not travel): each key must map to columns of the observation superset."""
    from human_robot_gym_amd.vec_env import OBS_COLUMNS
    used = ["dist_eef_to_human_head", "vec_eef_to_object", "vec_eef_to_target", "object_gripped", "robot0_gripper_qpos", "gripper_aperture",
            "dist_eef_to_human_rh", "dist_eef_to_human_lh", "goal_difference", "board_quat", "vec_eef_to_human_lh", "vec_eef_to_human_rh",
            "board_gripped", "vec_eef_to_all_objects", "object-state"]
    od = hrg._cstruct.CONST["HRG_OBS_DIM"]
    for k in used:
        cols = list(OBS_COLUMNS[k])
        assert cols and all(0 <= c < od for c in cols) and len(set(cols)) == len(cols), k


def test_monitor_rows_do_not_depend_on_the_info_dicts(tmp_path):
    """SB3's Monitor writes one r,l,t row per finished episode whatever the caller does with the info dicts: with info_dicts=False (the fast path) the rows are the
    same as with them."""
    import os
    rows = {}
    for flag in (True, False):
        d = tmp_path / f"m{int(flag)}"
        env = _vec(6, dict(shield_type="OFF", horizon=5, seed=2), info_dicts=flag, monitor_dir=str(d), monitor_kwargs=dict(info_keywords=("n_goal_reached", "timeout")))
        env.reset()
        rng = np.random.RandomState(0)
        for _ in range(11):
            env.step(rng.uniform(-1, 1, (6, 7)))
        env.close()
        lines = open(os.path.join(str(d), "hip_batch_0.monitor.csv")).read().splitlines()
        assert lines[1] == "r,l,t,n_goal_reached,timeout"
        rows[flag] = [l.split(",")[:2] + l.split(",")[3:] for l in lines[2:]]   # (without the wall-clock column)
    assert len(rows[True]) >= 12 and rows[True] == rows[False]


def test_robot_geometry_reaches_the_model_through_the_vec_env():
    """`HipVecEnv(robot_geometry="hull")` composes the model with the arm links' hull tables (the backend factory sees the description the env built); the default
    stays the capsule model, the object tasks refuse hulls (no hull - box narrowphase)."""
    clips = hrg.synthetic_clips(2, seed=0, min_frames=100, max_frames=120)
    seen = {}

    def factory(desc, clips_, n, env_id0):
        seen["hulls"] = (int(desc.robot_hulls), list(desc.hull_off))
        return OracleBackend(desc, clips_, n)

    env = HipVecEnv(3, env_kwargs=dict(shield_type="OFF", horizon=20), clips=clips, backend=factory, robot_geometry="hull")
    assert seen["hulls"][0] == 1 and seen["hulls"][1][-1] == 4321
    env.reset()
    obs, r, d, info = env.step(np.zeros((3, 7)))
    assert np.isfinite(obs).all()
    env.close()
    env = HipVecEnv(3, env_kwargs=dict(shield_type="OFF", horizon=20), clips=clips, backend=factory)
    assert seen["hulls"][0] == 0
    env.close()
    with pytest.raises(ValueError):
        hrg.build_model_desc(None, robot_geometry="mesh")


# BASELINE configs[0]: the keyword set of demos/demo_reach_human_environment.py:41-62 (suite.make) + its CollisionPreventionWrapper (75-77)
DEMO_KW = dict(robot_base_offset=[0, 0, 0], reward_shaping=True, control_freq=5, hard_reset=False, horizon=1000, shield_type="SSM", base_human_pos_offset=[1.0, 0.0, 0.0],
               goal_dist=0.0001, human_rand=[1.0, 0.5, 0.2], seed=0)
DEMO_CP = dict(replace_type=0, n_resamples=20)


def demo_rollout(env, steps=100):
    """the demo's loop (85-100) with an expert in the spirit of ReachHumanExpert: move along the goal difference, some noise"""
    obs = env.reset()
    rng = np.random.RandomState(0)
    out = []
    for t in range(steps):
        gd = env.observation_dict(obs)["goal_difference"]
        a = np.zeros(7)
        a[:6] = np.clip(5.0 * gd + 0.05 * rng.randn(6), -1, 1)
        obs, r, done, info = env.step(a)
        out.append((obs.copy(), r, done, dict(info)))
        if done:
            break
    return out


def test_reference_demo_configuration_on_the_oracle_backend():
    """One env, the demo's keywords, 100 steps of its loop: the 4-tuple API, the dense reward, the shield's counters, the wrapper's resample counter."""
    clips = hrg.synthetic_clips(3, seed=0, min_frames=600, max_frames=900)
    desc = hrg.build_model_desc(DEMO_KW, n_clips=3, collision_prevention=DEMO_CP)
    assert desc.n_cycles == 50 and desc.horizon == 1000 and desc.goal_dist == 0.0001            # control_freq 5: 50 shield cycles per policy step
    env = HipGymEnv(env_kwargs=DEMO_KW, clips=clips, collision_prevention=DEMO_CP, backend=OracleBackend(desc, clips, 1))
    out = demo_rollout(env)
    assert len(out) == 100 and not out[-1][2]
    rewards = np.array([o[1] for o in out])
    assert np.isfinite(rewards).all() and (rewards < 0).all() and rewards[-1] > rewards[0]       # dense reward = -distance to the goal: the expert closes in
    info = out[-1][3]
    assert info["n_goal_reached"] == 0 and "action_resamples" in info and info["failsafe_interventions"] >= 0
    env.close()
