"""`create_training_vec_env(config, evaluation_mode)` — the reference's env construction for a training run
(utils/training_utils_SB3.py:45-77), returning the batched HIP VecEnv instead of a SubprocVecEnv.

The reference builds `wrapper_class = get_environment_wrap_fn(config)` (utils/training_utils.py:350-410: per-env gym wrappers chosen by
`config.wrappers.*`) and hands it to `make_vec_env`.  Per-env Python wrappers cannot wrap a batch that lives on the GPU, so the
wrappers the stepper implements in its kernels are read from the same config nodes instead:

  config.wrappers.collision_prevention  -> CollisionPreventionWrapper (wrappers/collision_prevention_wrapper.py) in the step kernel's prologue
  config.wrappers.ik_position_delta     -> IKPositionDeltaWrapper (wrappers/ik_position_delta_wrapper.py) in the step kernel's prologue
  config.wrappers.dataset_obs_norm      -> DatasetObsNormWrapper (wrappers/dataset_wrapper.py:160-300) on the host, when the statistics are
                                           given (mean / std in the config, or datasets/<name>/observations.csv)
  config.wrappers.action_based_expert_imitation_reward with alpha == 0 and rsi_prob == 0 (the *-SAC baselines of config_icra_2024): the
                                           wrapper then returns the environment reward unchanged and never resets to a dataset state; it is skipped
  anything else that is configured (state/action based imitation rewards with a weight, reference-state initialisation, visualisation) raises.

`config` may be the reference's OmegaConf `TrainingConfig`, or any object / dict with the same attribute tree (the tests use a plain namespace).
"""
import csv
import os
from argparse import Namespace
from types import SimpleNamespace
from typing import Any, Dict, Optional

import numpy as np

from .env_util import make_vec_env


def _get(node, name, default=None):
    if node is None:
        return default
    if isinstance(node, dict):
        return node.get(name, default)
    try:
        v = getattr(node, name)
    except Exception:  # noqa: BLE001  (OmegaConf raises its own errors for missing keys)
        try:
            return node[name]
        except Exception:  # noqa: BLE001
            return default
    return v


def _plain(node):
    """OmegaConf node / namespace / dict -> plain dicts and lists (OmegaConf.to_container(resolve=True) when available)."""
    if node is None or isinstance(node, (str, int, float, bool)):
        return node
    try:  # pragma: no cover - omegaconf is not installed in the build container
        from omegaconf import OmegaConf
        if OmegaConf.is_config(node):
            return OmegaConf.to_container(node, resolve=True, throw_on_missing=True)
    except ImportError:
        pass
    if isinstance(node, dict):
        return {k: _plain(v) for k, v in node.items()}
    if isinstance(node, (list, tuple)):
        return [_plain(v) for v in node]
    if isinstance(node, (SimpleNamespace, Namespace)):
        return {k: _plain(v) for k, v in vars(node).items() if not k.startswith("_")}
    return node   # any other object (a ClipSet, a backend factory, ...) is a value, not a config node


def compose_environment_kwargs(config, evaluation_mode: bool = False) -> Dict[str, Any]:
    """utils/training_utils.py:71-88 (`_compose_environment_kwargs`): config.environment without env_id, + robots, + the evaluation seed.
    `controller_configs` is not composed: the stepper's controller is failsafe.json + schunk.json, compiled into the model description."""
    kwargs = dict(_plain(_get(config, "environment")) or {})
    kwargs.pop("env_id", None)
    kwargs["robots"] = _get(_get(config, "robot"), "name", "Schunk")
    if evaluation_mode and _get(_get(config, "run"), "eval_seed") is not None:
        kwargs["seed"] = _get(_get(config, "run"), "eval_seed")
    return kwargs


def _obs_norm_from_config(node) -> Optional[Dict[str, Any]]:
    kw = dict(_plain(node))
    mean, std = kw.get("mean"), kw.get("std")
    if mean is None or std is None:
        path = os.path.join("datasets", str(kw.get("dataset_name")), "observations.csv")   # dataset_wrapper.py:213-216
        if not os.path.exists(path):
            raise NotImplementedError(f"wrappers.dataset_obs_norm: no mean/std in the config and no {path}; computing the statistics from the pickled "
                                      "dataset (dataset_wrapper.py:217-221) is not supported")
        with open(path, newline="") as f:
            rows = list(csv.DictReader(f))
        mean = [float(r["mean"]) for r in rows] if mean is None else mean
        std = [float(r["std"]) for r in rows] if std is None else std
    return dict(mean=np.asarray(mean, np.float64), std=np.asarray(std, np.float64), squash_factor=kw.get("squash_factor"),
                allow_different_observation_shapes=bool(kw.get("allow_different_observation_shapes", False)))


def wrapper_kwargs_from_config(config) -> Dict[str, Any]:
    """`get_environment_wrap_fn(config)` (utils/training_utils.py:350-410), translated into HipVecEnv keyword arguments."""
    w = _get(config, "wrappers")
    out: Dict[str, Any] = {}
    cp = _get(w, "collision_prevention")
    if cp is not None:
        out["collision_prevention"] = dict(_plain(cp))
    ik = _get(w, "ik_position_delta")
    if ik is not None:   # env_has_cartesian_action_space (training_utils.py:204-206); kwargs as _compose_ik_position_delta_wrapper_kwargs (177-201)
        ikw = dict(_plain(ik))
        ikw.pop("urdf_file", None)   # the kernel's chain is the stepper's own model of robot_pybullet.urdf
        out["ik_position_delta"] = ikw
    if _get(w, "state_based_expert_imitation_reward") is not None:
        raise NotImplementedError("wrappers.state_based_expert_imitation_reward: imitation-reward wrappers run per env in Python and are outside the batched stepper")
    ab = _get(w, "action_based_expert_imitation_reward")
    if ab is not None:
        abk = dict(_plain(ab))
        if float(abk.get("alpha") or 0.0) != 0.0 or float(abk.get("rsi_prob") or 0.0) != 0.0:
            raise NotImplementedError("wrappers.action_based_expert_imitation_reward with alpha != 0 or rsi_prob != 0: imitation rewards / reference-state "
                                      "initialisation are outside the batched stepper (alpha = 0, rsi_prob = 0 leaves the environment reward unchanged and is skipped)")
    dn = _get(w, "dataset_obs_norm")
    if dn is not None:
        out["obs_norm"] = _obs_norm_from_config(dn)
    if _get(w, "visualization") is not None:
        raise NotImplementedError("wrappers.visualization: the batched stepper has no renderer")
    return out


def create_training_vec_env(config, evaluation_mode: bool = False, wrapper_class=None):
    """Drop-in for `human_robot_gym.utils.training_utils_SB3.create_training_vec_env` (45-77).  `wrapper_class`, when given (the reference always
    builds one from the same config), is accepted and not called: what it would have wrapped is read from `config.wrappers` here."""
    run = _get(config, "run")
    env_kwargs = compose_environment_kwargs(config, evaluation_mode)
    vec_kw = dict(_plain(_get(run, "vec_env_kwargs")) or {})
    vec_kw.update(wrapper_kwargs_from_config(config))
    vec_kw["_wrappers_from_config"] = True   # tells make_vec_env that `wrapper_class` (if any) has been translated above
    return make_vec_env(
        env_id=_get(_get(config, "environment"), "env_id"),
        type=_get(run, "env_type", "env"),
        obs_keys=_plain(_get(run, "obs_keys")),
        expert_obs_keys=_plain(_get(run, "expert_obs_keys")),
        n_envs=int(_get(run, "n_envs", 1)),
        seed=_get(run, "seed"),
        start_index=int(_get(run, "start_index", 0) or 0),
        monitor_dir=_get(run, "monitor_dir"),
        wrapper_class=wrapper_class,
        env_kwargs=env_kwargs,
        vec_env_cls=None,
        vec_env_kwargs=vec_kw,
        monitor_kwargs=_plain(_get(run, "monitor_kwargs")),
        wrapper_kwargs=None,
    )
