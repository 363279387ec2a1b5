"""`create_training_vec_env(config, evaluation_mode)` — the one-import boundary (utils/training_utils_SB3.py:45-77) on a fake config object,
oracle backend (host logic; no GPU)."""
import os
from types import SimpleNamespace as NS

import numpy as np
import pytest

import human_robot_gym_amd as hrg
from helpers import OracleBackend


def _config(tmp_path=None, **wrappers):
    clips = hrg.synthetic_clips(2, seed=0, min_frames=200, max_frames=300)
    return NS(
        robot=NS(name="Schunk", controller_config_path="controllers/failsafe_controller/config/failsafe.json", robot_config_path="models/robots/config/schunk.json"),
        environment=NS(env_id="PickPlaceHumanCart", robot_base_offset=[0.0, 0.0, 0.0], env_configuration="default", controller_configs=None, gripper_types="default",
                       initialization_noise="default", use_camera_obs=False, use_object_obs=True, has_renderer=False, has_offscreen_renderer=False,
                       render_camera=None, hard_reset=False, control_freq=10, horizon=6, shield_type="SSM", control_sample_time=0.004, seed=3,
                       human_animation_names=["PickPlace/0", "PickPlace/1"], reward_shaping=False, object_gripped_reward=-0.25, done_at_success=False, verbose=False),
        wrappers=NS(**wrappers),
        run=NS(n_envs=3, seed=7, eval_seed=11, start_index=2, env_type="env", obs_keys=["object_gripped", "vec_eef_to_object", "gripper_aperture"], expert_obs_keys=None,
               monitor_dir=str(tmp_path) if tmp_path is not None else None, monitor_kwargs=dict(info_keywords=("n_goal_reached",)),
               vec_env_kwargs=dict(backend=OracleBackend, clips=clips)),
    )


def test_config_wrappers_become_kernel_front_ends_and_monitor_csv_is_written(tmp_path):
    cfg = _config(tmp_path, collision_prevention=NS(replace_type=0, n_resamples=20),
                  ik_position_delta=NS(urdf_file="models/assets/robots/schunk/robot_pybullet.urdf", action_limit=0.1, x_output_max=1, x_position_limits=None,
                                       residual_threshold=0.001, max_iter=50),
                  action_based_expert_imitation_reward=NS(dataset_name="d", alpha=0.0, rsi_prob=0.0, beta=0.7, iota_m=0.1, iota_g=0.5, m_sim_fn="gaussian", g_sim_fn="gaussian"))
    env = hrg.create_training_vec_env(cfg, wrapper_class=lambda e: e)   # the reference always passes a wrap fn built from the same config: accepted, not called
    assert env.num_envs == 3 and env.action_space.shape == (4,)        # Cartesian action space of the IK front-end (ik_position_delta_wrapper.py:84-88)
    assert np.allclose(env.action_space.high, [0.1, 0.1, 0.1, 1.0]) and env.observation_space.shape == (5,)
    assert env._desc.cp_enabled == 1 and env._desc.cp_n_resamples == 20 and env._desc.ik_enabled == 1 and env._desc.seed == 7
    assert env._env_id0 == 2
    env.reset()
    rng = np.random.RandomState(0)
    for _ in range(6):
        obs, rew, done, infos = env.step(rng.uniform(-0.1, 0.1, (3, 4)))
    assert done.all() and infos[0]["TimeLimit.truncated"] and "action_resamples" in infos[0]
    env.close()
    lines = open(os.path.join(str(tmp_path), "hip_batch_2.monitor.csv")).read().splitlines()
    assert lines[0].startswith('#{"t_start"') and lines[1] == "r,l,t,n_goal_reached" and len(lines) == 2 + 3
    r, l, t, ng = lines[2].split(",")
    assert int(l) == 6 and float(r) == pytest.approx(infos[0]["episode"]["r"], abs=1e-5) and int(ng) == infos[0]["n_goal_reached"]
    ev = hrg.create_training_vec_env(_config(), evaluation_mode=True)
    assert ev._desc.seed == 7   # run.seed re-seeds the env like make_vec_env's env.seed(seed + rank) (env_util_SB3.py:57-58); eval_seed only enters env_kwargs
    assert ev.env_kwargs["seed"] == 7 and ev.action_space.shape == (7,)
    ev.close()


def test_observation_normalisation_from_config():
    cfg = _config(dataset_obs_norm=NS(dataset_name=None, squash_factor=0.5, allow_different_observation_shapes=False, mean=[0.0, 0.1, 0.2, 0.3, 0.5], std=[1.0, 2.0, 0.0, 4.0, 0.25]))
    env, raw = hrg.create_training_vec_env(cfg), hrg.create_training_vec_env(_config())
    o, r = env.reset(), raw.reset()
    want = np.tanh(0.5 * (r - np.array([0.0, 0.1, 0.2, 0.3, 0.5])) / np.array([1.0, 2.0, 1.0, 4.0, 0.25]))   # std == 0 -> 1 (dataset_wrapper.py:241-242)
    np.testing.assert_allclose(o, want, rtol=1e-6, atol=1e-7)
    assert env.observation_space.high.max() == 1.0
    env.close(); raw.close()
    bad = _config(dataset_obs_norm=NS(dataset_name=None, squash_factor=None, allow_different_observation_shapes=False, mean=[0.0], std=[1.0]))
    with pytest.raises(ValueError):
        hrg.create_training_vec_env(bad)


@pytest.mark.parametrize("wrappers", [
    dict(state_based_expert_imitation_reward=NS(alpha=0.5)),
    dict(action_based_expert_imitation_reward=NS(dataset_name="d", alpha=0.3, rsi_prob=0.0)),
    dict(action_based_expert_imitation_reward=NS(dataset_name="d", alpha=0.0, rsi_prob=0.5)),
    dict(dataset_obs_norm=NS(dataset_name="no-such-dataset", squash_factor=None, allow_different_observation_shapes=False, mean=None, std=None)),
    dict(visualization=NS()),
])
def test_unsupported_wrappers_fail_loudly(wrappers):
    with pytest.raises(NotImplementedError):
        hrg.create_training_vec_env(_config(**wrappers))


def test_make_vec_env_refuses_an_opaque_wrapper_class_and_unimplemented_env_kwargs():
    with pytest.raises(NotImplementedError, match="create_training_vec_env"):
        hrg.make_vec_env("ReachHuman", n_envs=2, wrapper_class=lambda e: e)
    for kw in (dict(randomize_initial_pos=True), dict(table_friction=[0.5, 0.005, 0.0001]), dict(initialization_noise=None), dict(no_such_key=1)):
        with pytest.raises(NotImplementedError):
            hrg.build_model_desc(kw)
    hrg.build_model_desc(dict(randomize_initial_pos=False, table_friction=(1.0, 5e-3, 1e-4), initialization_noise="default", has_renderer=False))


def test_hammering_environment_config_of_the_reference_is_accepted():
    """training/config/environment/default/collaborative_hammering_cart.yaml through the one-import boundary, oracle backend."""
    from human_robot_gym_amd.mixed import task_clips
    cfg = _config()
    cfg.environment = NS(env_id="CollaborativeHammeringCart", robot_base_offset=[0.0, 0.0, 0.0], env_configuration="default", controller_configs=None, gripper_types="default",
                         initialization_noise="default", use_camera_obs=False, use_object_obs=True, has_renderer=False, has_offscreen_renderer=False, render_camera=None,
                         hard_reset=False, control_freq=10, horizon=5, shield_type="SSM", control_sample_time=0.004, seed=3, verbose=False,
                         table_full_size=[1.5, 2.0, 0.05], table_friction=[1.0, 5e-3, 1e-4], board_full_size=[1.0, 0.4, 0.03], nail_hammered_in_reward=-1.0,
                         hammer_gripped_reward_bonus=0.0, goal_tolerance=0.05, n_nail_placements_sampled_per_100_steps=1, gripper_controllable=False,
                         human_animation_names=[f"CollaborativeHammering/{i}" for i in range(9)], obstacle_placement_initializer=None, human_animation_freq=100)
    cfg.run.obs_keys = ["hammer_gripped", "vec_eef_to_nail", "nail_hammering_progress", "board_quat", "quat_eef_to_board"]
    cfg.run.vec_env_kwargs = dict(backend=OracleBackend, clips=task_clips("CollaborativeHammeringCart", 2, min_frames=200, max_frames=260))
    env = hrg.create_training_vec_env(cfg)
    assert env.observation_space.shape == (1 + 3 + 1 + 4 + 4,) and env._desc.task == 9 and env._desc.gripper_controllable == 0
    obs = env.reset()
    assert obs.shape == (3, 13) and np.all(obs[:, 9:13] == 0.0)              # quat_eef_to_board: constant zeros in the reference (1276-1282)
    assert np.allclose(np.linalg.norm(obs[:, 5:9], axis=1), 1.0, atol=1e-6)   # board_quat
    for _ in range(5):
        obs, rew, done, infos = env.step(np.zeros((3, 7)))
    assert done.all() and infos[0]["TimeLimit.truncated"]
    env.close()
