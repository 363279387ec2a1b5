"""`make_vec_env` — same signature as the reference factory (utils/env_util_SB3.py:19-87), returning a HipVecEnv.

In the reference, `create_training_vec_env` (utils/training_utils_SB3.py:45-77) calls
`make_vec_env(env_id, type, obs_keys, expert_obs_keys, n_envs, seed, start_index, monitor_dir, wrapper_class,
env_kwargs, vec_env_cls=DummyVecEnv|SubprocVecEnv, vec_env_kwargs, monitor_kwargs, wrapper_kwargs)`.
Pointing that call at this function swaps the per-process CPU envs for the GPU batch.
"""
from typing import Any, Callable, Dict, List, Optional

from .vec_env import HipVecEnv


def make_vec_env(
    env_id: str,
    type: str = "env",  # noqa: A002 - reference keyword
    obs_keys: Optional[List[str]] = None,
    expert_obs_keys: Optional[List[str]] = None,
    n_envs: int = 1,
    seed: Optional[int] = None,
    start_index: int = 0,
    monitor_dir: Optional[str] = None,
    wrapper_class: Optional[Callable] = None,
    env_kwargs: Optional[Dict[str, Any]] = None,
    vec_env_cls=None,
    vec_env_kwargs: Optional[Dict[str, Any]] = None,
    monitor_kwargs: Optional[Dict[str, Any]] = None,
    wrapper_kwargs: Optional[Dict[str, Any]] = None,
) -> HipVecEnv:
    assert type in ["env", "goal_env"], "The type of environment must be either 'env' or 'goal_env'."
    kw = dict(vec_env_kwargs or {})
    translated = kw.pop("_wrappers_from_config", False)
    if wrapper_class is not None and not translated:
        # an opaque per-env closure (get_environment_wrap_fn(config), utils/training_utils.py:350-410) cannot wrap a batch that lives on the GPU and
        # cannot be inspected either: the wrappers it would apply are read from the same config by training_utils.create_training_vec_env
        raise NotImplementedError("wrapper_class: per-env gym wrappers cannot wrap a batched env.  Call human_robot_gym_amd.create_training_vec_env(config) "
                                  "(it reads config.wrappers.collision_prevention / ik_position_delta / dataset_obs_norm itself), or pass "
                                  "vec_env_kwargs=dict(collision_prevention=dict(replace_type=0, n_resamples=20), ik_position_delta=dict(...))")
    if monitor_dir is not None:   # utils/env_util_SB3.py:60-66
        kw["monitor_dir"] = monitor_dir
        kw["monitor_kwargs"] = monitor_kwargs
    # SB3 seeds env rank r with seed + r; here streams are keyed by (seed, global env id), ids start at start_index
    return HipVecEnv(n_envs=n_envs, env_id=env_id, env_kwargs=env_kwargs, obs_keys=obs_keys, seed=seed, env_id0=start_index, expert_obs_keys=expert_obs_keys, goal_env=(type == "goal_env"), **kw)
