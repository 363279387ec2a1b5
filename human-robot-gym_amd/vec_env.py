"""Drop-in vectorised / single-env facades over the HIP stepper.

`HipVecEnv` keeps the stable-baselines3 `VecEnv` contract that the reference's training utilities consume
(`make_vec_env(..., vec_env_cls=SubprocVecEnv)`: utils/env_util_SB3.py:75-87, utils/training_utils_SB3.py:60-75):
`reset() -> obs[N,18]`, `step_async(actions)`, `step_wait() -> (obs, rewards, dones, infos)`, auto-reset with
`infos[i]["terminal_observation"]`, Monitor-style `infos[i]["episode"] = {"r","l","t"}`, `TimeLimit.truncated`
(wrappers/time_limit.py:40-43) and the `log_info_keys` of training/config/run/default_training.yaml:18-29.
The wrapper stack of the reference (Monitor -> TimeLimit -> GymWrapper -> ReachHuman) is folded into the batch:
flattening to `[object-state, goal_difference]` (human_reach_ppo_parallel.yaml:14-16) happens in the kernel.

`HipGymEnv` is the single-env gym-0.21 facade (4-tuple step) for config 1 (demos/demo_reach_human_environment.py).

If stable-baselines3 / gym are installed the classes subclass their ABCs; otherwise light stand-ins with the same
attributes are used (neither package is available in the build image).
"""
import os
import time
from collections import OrderedDict

import numpy as np

from ._cstruct import CONST
from .animation import synthetic_clips
from .model import build_model_desc, DEFAULT_ENV_KWARGS, ENV_DEFAULTS

try:  # pragma: no cover - not installed in the build image
    from stable_baselines3.common.vec_env import VecEnv as _VecEnvBase
except Exception:  # noqa: BLE001
    class _VecEnvBase:  # minimal stand-in with the attributes SB3 algorithms read
        def __init__(self, num_envs, observation_space, action_space):
            self.num_envs = num_envs
            self.observation_space = observation_space
            self.action_space = action_space

        def step(self, actions):
            self.step_async(actions)
            return self.step_wait()

try:  # pragma: no cover
    from gym import spaces as _spaces
    _Box = _spaces.Box
except Exception:  # noqa: BLE001
    class _Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.shape = tuple(shape if shape is not None else np.shape(low))
            self.low = np.broadcast_to(np.asarray(low, dtype), self.shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype), self.shape).copy()
            self.dtype = np.dtype(dtype)

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1.0)
            hi = np.where(np.isfinite(self.high), self.high, 1.0)
            return np.random.uniform(lo, hi).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

try:  # pragma: no cover
    _DictSpace = _spaces.Dict
except Exception:  # noqa: BLE001
    class _DictSpace:
        def __init__(self, spaces):
            self.spaces = dict(spaces)

        def __getitem__(self, k):
            return self.spaces[k]

INFO_KEYS = [  # column order of the kernel's info block (include/hrgym.h)
    "collision", "collision_type", "n_collisions", "n_collisions_static", "n_collisions_robot", "n_collisions_human",
    "n_collisions_critical", "timeout", "failsafe_interventions", "n_goal_reached", "TimeLimit.truncated", "sim_crash",
    "action_resamples", "n_object_handed_over",
]
INFO_KEY_ALIASES = {"CollaborativeStackingCart": {"n_object_handed_over": "max_stack_height"}}   # the task-specific info column (include/hrgym.h HRG_INFO_MAX_STACK_HEIGHT)
_BOOL_KEYS = {"collision", "timeout", "TimeLimit.truncated", "sim_crash"}
_BOOL_ITEMS = [(j, k) for j, k in enumerate(INFO_KEYS) if k in _BOOL_KEYS]
OBS_KEYS = ["object-state", "goal_difference"]  # default: training/config/human_reach_ppo_parallel.yaml:14-16
# training/config/run/obs_keys of the pick-place experiments (e.g. PP-SAC): the observables the policy sees
PICK_PLACE_OBS_KEYS = ["object_gripped", "vec_eef_to_object", "vec_eef_to_target", "gripper_aperture", "dist_eef_to_human_head",
                       "dist_eef_to_human_lh", "dist_eef_to_human_rh"]
STACKING_OBS_KEYS = ["object_gripped", "vec_eef_to_all_objects", "gripper_aperture", "dist_eef_to_human_head", "dist_eef_to_human_lh", "dist_eef_to_human_rh"]   # CS-SAC.yaml run.obs_keys
LIFTING_OBS_KEYS = ["board_quat", "dist_eef_to_human_head", "vec_eef_to_human_lh", "vec_eef_to_human_rh"]   # CL-SAC.yaml run.obs_keys
HAMMERING_OBS_KEYS = ["hammer_gripped", "vec_eef_to_nail", "nail_hammering_progress", "vec_eef_to_board", "board_quat", "dist_eef_to_human_head", "dist_eef_to_human_lh",
                      "dist_eef_to_human_rh"]   # no run config ships for this task; the expert reads vec_eef_to_nail (collaborative_hammering_cart_expert.py:24-25)
DEFAULT_OBS_KEYS = {k: (HAMMERING_OBS_KEYS if k == "CollaborativeHammeringCart" else OBS_KEYS if k == "ReachHuman" else (LIFTING_OBS_KEYS if k == "CollaborativeLiftingCart" else (STACKING_OBS_KEYS if k == "CollaborativeStackingCart" else PICK_PLACE_OBS_KEYS)))
                    for k in ENV_DEFAULTS}
# columns of the kernel's observation superset (include/hrgym.h HRG_OBS_DIM) per robosuite observable / modality key
OBS_COLUMNS = {
    "object-state": range(0, 12), "goal_difference": range(12, 18), "robot0_joint_pos": range(18, 24),
    "robot0_joint_vel": range(24, 30), "robot0_eef_pos": range(30, 33), "desired_goal": range(33, 39),
    "robot0_proprio-state": range(18, 33), "goal-state": list(range(33, 39)) + list(range(12, 18)),
    "vec_eef_to_human_lh": range(0, 3), "dist_eef_to_human_lh": range(3, 4), "vec_eef_to_human_rh": range(4, 7),
    "dist_eef_to_human_rh": range(7, 8), "vec_eef_to_human_head": range(8, 11), "dist_eef_to_human_head": range(11, 12),
    # PickPlaceHumanCart (pick_place_human_cartesian_env.py:726-841); zero columns for ReachHuman
    "object_gripped": range(39, 40), "vec_eef_to_object": range(40, 43), "vec_eef_to_target": range(43, 46),
    "gripper_aperture": range(46, 47), "object_pos": range(47, 50), "target_pos": range(50, 53),
    "robot0_gripper_qpos": range(53, 55), "robot0_gripper_qvel": range(55, 57),
    # CollaborativeLiftingCart (collaborative_lifting_cartesian_env.py:982-1085): the board sits in the object columns, its balance in the
    # first target column
    "board_pos": range(47, 50), "vec_eef_to_board": range(40, 43), "board_gripped": range(39, 40), "board_balance": range(50, 51),
    "board_quat": [43, 44, 45, 51],
    "object_quat": range(12, 16),   # the cube tasks: orientation of the manipulation object, (x, y, z, w)
    # the handover tasks' quat_eef_to_object / CollaborativeLiftingCart's quat_eef_to_board, computed like the reference does (see oracle compute_obs_e)
    "quat_eef_to_object": range(57, 61),
    # CollaborativeStackingCart (collaborative_stacking_cartesian_env.py:1306-1524): the vectors to the four cubes (a, b, l, r) take the 12 joint-space columns the cube
    # tasks leave empty; vec_eef_to_object / object_pos follow the cube the robot has to place next; next_target_pos sits in the target columns
    "vec_eef_to_all_objects": list(range(12, 18)) + list(range(33, 39)), "vec_eef_to_object_a": range(12, 15), "vec_eef_to_object_b": range(15, 18),
    "vec_eef_to_object_l": range(33, 36), "vec_eef_to_object_r": range(36, 39), "next_target_pos": range(50, 53),
}


# per-task column overrides: the same observable name sits in other columns of the superset (oracle: compute_obs_hammer)
OBS_COLUMNS_TASK = {
    "CollaborativeHammeringCart": {   # collaborative_hammering_cartesian_env.py:1151-1323
        "hammer_quat": range(12, 16), "board_pos": range(33, 36), "vec_eef_to_board": range(36, 39), "hammer_gripped": range(39, 40),
        "vec_eef_to_hammer": range(40, 43), "vec_eef_to_nail": range(43, 46), "hammer_pos": range(47, 50), "nail_pos": range(50, 53),
        "board_quat": range(57, 61), "nail_hammering_progress": range(61, 62),
        # "object_quat" is never in this env's observation cache, so the two relative quaternions are constant zeros in the reference (1217-1225, 1259-1267)
        "quat_eef_to_hammer": [62, 63, 62, 63], "quat_eef_to_board": [62, 63, 62, 63],
        "desired_goal": range(50, 53),   # _get_desired_goal_from_obs: nail_pos (606-621)
    },
}
for _k in ("hammer_quat", "hammer_gripped", "vec_eef_to_hammer", "vec_eef_to_nail", "hammer_pos", "nail_pos", "nail_hammering_progress", "quat_eef_to_hammer"):
    OBS_COLUMNS.setdefault(_k, OBS_COLUMNS_TASK["CollaborativeHammeringCart"][_k])
OBS_COLUMNS["quat_eef_to_board"] = range(57, 61)   # CollaborativeLiftingCart (hammering overrides it with its constant zeros)


_EAGER_KEYS = frozenset(("terminal_observation", "episode", "TimeLimit.truncated"))   # what SB3's rollout loops look up on every info


class LazyInfo(dict):
    """Per-env info dict whose kernel-derived entries (INFO_KEYS, "action", the expert observations) are filled in on first use.

    SB3's collect loops only `.get()` "episode" / "terminal_observation" / "TimeLimit.truncated" on every info of every step; those
    are stored eagerly, so a 4096-env step does not pay for 4096 x 15 dict entries nobody reads.  Any other access (indexing, `in`,
    iteration, `len`, `==`, `dict(info)`, copy / pickle) materialises the row first; afterwards the object is an ordinary dict."""
    __slots__ = ("_src", "_i")

    def _fill(self):
        src = self._src
        if src is not None:
            self._src = None
            src.fill(self, self._i)

    def get(self, key, default=None):
        if self._src is not None and key not in _EAGER_KEYS:
            self._fill()
        return dict.get(self, key, default)

    def __getitem__(self, key):
        if self._src is not None and key not in _EAGER_KEYS:
            self._fill()
        return dict.__getitem__(self, key)

    def __contains__(self, key):
        if self._src is not None and key not in _EAGER_KEYS:
            self._fill()
        return dict.__contains__(self, key)

    def _filled(name):  # noqa: N805
        def method(self, *a, **k):
            self._fill()
            return getattr(dict, name)(self, *a, **k)
        method.__name__ = name
        return method

    for _n in ("__iter__", "__len__", "__eq__", "__ne__", "__repr__", "__setitem__", "__delitem__", "__or__", "__ror__", "__ior__", "__reversed__",
               "keys", "values", "items", "copy", "pop", "popitem", "setdefault", "update", "clear"):
        locals()[_n] = _filled(_n)
    del _n, _filled
    __hash__ = None

    def __reduce__(self):   # pickle / deepcopy: a plain dict
        self._fill()
        return (dict, (dict(self),))


class _InfoSource:
    """What the infos of one step are filled from: a copy of the info block, the executed actions, the expert views."""

    def __init__(self, rows, acts, expert, prev_full, term_obs, keys=None):
        self.rows, self.acts, self.expert, self.prev_full, self.term_obs = rows, acts, expert, prev_full, term_obs
        self.keys = keys or INFO_KEYS

    def fill(self, d, i):
        row = self.rows[i].tolist()
        set_ = dict.__setitem__
        for k, v in zip(self.keys, row):
            if k != "TimeLimit.truncated":
                set_(d, k, v)
        for j, k in _BOOL_ITEMS:
            if k != "TimeLimit.truncated":
                set_(d, k, row[j] != 0)
        set_(d, "action", self.acts[i])  # collision_prevention_wrapper.py:42-43: the executed action
        if self.expert is not None:  # expert_obs_wrapper.py:171-175 (the step's own observation: pre-reset where done)
            prev, cur = self.prev_full[i], self.term_obs[i]
            set_(d, "previous_expert_observation", {k: np.array(prev[list(OBS_COLUMNS[k])]) for k in self.expert})
            set_(d, "current_expert_observation", {k: np.array(cur[list(OBS_COLUMNS[k])]) for k in self.expert})


class _TorchBackend:
    """numpy <-> HipBatch adapter: one H2D copy of the actions, one D2H copy of the packed output block per step."""

    def __init__(self, desc, clips, n_envs, env_id0, device):
        import torch
        from ._lib import HipBatch
        self.torch = torch
        self.batch = HipBatch(desc, clips, n_envs, env_id0=env_id0, device=device)
        self.n = n_envs
        lay = self.batch.packed_layout
        self._host = torch.empty(self.batch.packed.numel(), dtype=torch.uint8, pin_memory=True)
        o, s = lay["offsets"], lay["sizes"]
        hb = self._host.numpy()
        od, idim = CONST["HRG_OBS_DIM"], CONST["HRG_INFO_DIM"]
        self.obs = hb[o[0]:o[0] + s[0]].view(np.float32).reshape(n_envs, od)
        self.term_obs = hb[o[1]:o[1] + s[1]].view(np.float32).reshape(n_envs, od)
        self.reward = hb[o[2]:o[2] + s[2]].view(np.float32)
        self.info = hb[o[3]:o[3] + s[3]].view(np.int32).reshape(n_envs, idim)
        self.done = hb[o[4]:o[4] + s[4]]

    def _fetch(self):
        self._host.copy_(self.batch.packed, non_blocking=False)

    def reset(self):
        self.batch.reset()
        self._fetch()
        return self.obs

    def step_async(self, actions):
        self._act = self.torch.from_numpy(np.ascontiguousarray(actions, np.float64)).to(self.batch.device, non_blocking=True)
        self.batch.step(self._act)

    def step_wait(self):
        self._fetch()
        return self.obs, self.term_obs, self.reward, self.done, self.info

    def executed_actions(self):
        """Actions after CollisionPreventionWrapper screening (the kernel rewrites the action rows in place)."""
        return self._act.cpu().numpy()

    def close(self):
        self.batch.close()


class HipVecEnv(_VecEnvBase):
    """Batched ReachHuman environments stepped by the HIP library (one wavefront per env).

    Args mirror the reference factory (`utils/env_util_SB3.py:19-87`): `env_kwargs` is the dict composed at
    `utils/training_utils.py:71-88`; `seed` plays the role of `seed + rank` (per-env streams are keyed by the
    global env id, so sharding does not change results)."""
    _norm = None      # (mean, std, squash_factor) of DatasetObsNormWrapper, when configured
    _monitor = None   # open Monitor csv, when monitor_dir is given
    _monitor_keys = ()
    _info_keys = INFO_KEYS   # names of the info columns (a task may rename its task-specific column: INFO_KEY_ALIASES)

    def __init__(self, n_envs=1, env_id="ReachHuman", env_kwargs=None, obs_keys=None, seed=None, clips=None,
                 device=0, env_id0=0, backend=None, info_dicts=True, collision_prevention=None, goal_check=True, ik_position_delta=None,
                 expert_obs_keys=None, goal_env=False, obs_norm=None, monitor_dir=None, monitor_kwargs=None, reach_box=False, robot_geometry="capsule"):
        if env_id not in ENV_DEFAULTS:
            raise NotImplementedError(f"env_id {env_id!r}: the HIP stepper covers {sorted(ENV_DEFAULTS)} (DESIGN.md §6)")
        self.env_id = env_id
        self._info_keys = [INFO_KEY_ALIASES.get(env_id, {}).get(k, k) for k in INFO_KEYS]
        # GoalEnvironmentGymWrapper (wrappers/goal_env_wrapper.py): dict observations {observation, achieved_goal, desired_goal} and an
        # externalised reward for hindsight relabelling.  Goals per task: ReachHuman joint angles (reach_human_env.py:477-507),
        # the cube tasks [eef_pos, object_pos, object_gripped] vs target_pos (pick_place_human_cartesian_env.py:574-611)
        self.goal_env = bool(goal_env)
        if self.goal_env:
            if env_id in ("HumanObjectInspectionCart", "CollaborativeLiftingCart", "CollaborativeStackingCart", "CollaborativeHammeringCart"):
                raise NotImplementedError("goal_env: this task's success is a task phase, not a function of the goals")
            if obs_keys is None:  # goal_env_wrapper.py:62-71
                obs_keys = ["object-state", "robot0_proprio-state", "desired_goal"]
            self._ag_cols = np.array(list(range(18, 24)) if env_id == "ReachHuman" else [30, 31, 32, 47, 48, 49, 39], dtype=np.int64)
            self._dg_cols = np.array(list(range(33, 39)) if env_id == "ReachHuman" else [50, 51, 52], dtype=np.int64)
        keys = list(obs_keys) if obs_keys is not None else DEFAULT_OBS_KEYS[env_id]
        unknown = [k for k in keys if k not in OBS_COLUMNS]
        if unknown:
            raise NotImplementedError(f"obs_keys {unknown!r}: available {sorted(OBS_COLUMNS)}")
        self.obs_keys = keys
        # ExpertObsWrapper (wrappers/expert_obs_wrapper.py:155-184): infos carry the expert's view of the state before and after the step
        bad = [k for k in (expert_obs_keys or []) if k not in OBS_COLUMNS]
        if bad:
            raise NotImplementedError(f"expert_obs_keys {bad!r}: available {sorted(OBS_COLUMNS)}")
        self.expert_obs_keys = list(expert_obs_keys) if expert_obs_keys is not None else None
        self._expert_cur = None
        cols_of = dict(OBS_COLUMNS)
        if env_id != "ReachHuman":
            cols_of["desired_goal"] = OBS_COLUMNS["target_pos"]   # _get_desired_goal_from_obs of the cube tasks
        cols_of.update(OBS_COLUMNS_TASK.get(env_id, {}))
        self._cols_of = cols_of
        self._cols = np.array([c for k in keys for c in cols_of[k]], dtype=np.int64)  # GymWrapper: concatenate in key order
        kw = dict(env_kwargs or {})
        if seed is not None:
            kw["seed"] = int(seed)
        self.env_kwargs = kw
        self._clips = clips if clips is not None else synthetic_clips()
        # collision_prevention: dict(replace_type=0|1|2, n_resamples=20) = config/wrappers/collision_prevention/*.yaml
        # ik_position_delta: dict(action_limit=0.15, x_output_max=1, ...) = config/wrappers/ik_position_delta/*.yaml: actions become
        # [dx, dy, dz, gripper] (IKPositionDeltaWrapper, wrappers/ik_position_delta_wrapper.py), converted in the kernel
        self._cp, self._goal_check, self._ik = collision_prevention, goal_check, ik_position_delta
        self._reach_box = bool(reach_box)   # ReachHuman with its free smallBox object (stepped by the cube kernel); default: the lean model
        self._robot_geometry = robot_geometry   # "capsule" (default) | "hull": the arm links collide as the convex hulls of their meshes (ReachHuman only)
        self._desc = build_model_desc(kw, n_clips=self._clips.n_clips, collision_prevention=collision_prevention, goal_check=goal_check, env_id=env_id,
                                      ik_position_delta=ik_position_delta, reach_box=self._reach_box, robot_geometry=robot_geometry)
        self._device, self._env_id0 = device, env_id0
        if backend is not None and (isinstance(backend, type) or not hasattr(backend, "step_async")):   # a factory (desc, clips, n_envs, env_id0) -> backend: the caller cannot build the
            backend = backend(self._desc, self._clips, n_envs, env_id0)    # backend itself when the model description is composed here (create_training_vec_env)
        self._backend = backend if backend is not None else _TorchBackend(self._desc, self._clips, n_envs, env_id0, device)
        obs_space = _Box(-np.inf, np.inf, shape=(len(self._cols),), dtype=np.float32)
        if self.goal_env:
            goal_space = _Box(-np.inf, np.inf, shape=(len(self._dg_cols),), dtype=np.float32)
            ag_space = _Box(-np.inf, np.inf, shape=(len(self._ag_cols),), dtype=np.float32)
            obs_space = _DictSpace({"observation": obs_space, "desired_goal": goal_space, "achieved_goal": ag_space})
        if ik_position_delta is None:
            act_space = _Box(-1.0, 1.0, shape=(CONST["HRG_ACT_DIM"],), dtype=np.float32)
        else:  # ik_position_delta_wrapper.py:84-88: position delta limits + one gripper dof
            lim = float(self._desc.ik_action_limit)
            act_space = _Box(np.array([-lim] * 3 + [-1.0], np.float32), np.array([lim] * 3 + [1.0], np.float32), dtype=np.float32)
        super().__init__(n_envs, obs_space, act_space)
        self.info_dicts = info_dicts
        self._ep_ret = np.zeros(n_envs, np.float64)
        self._ep_len = np.zeros(n_envs, np.int64)
        self._t_start = time.time()
        self._actions = None
        self._last_full = None
        self.horizon = int(self._desc.horizon)
        # DatasetObsNormWrapper (wrappers/dataset_wrapper.py:160-300): (obs - mean) / std, optionally tanh(squash_factor * .), applied to the policy's
        # flat observation (and to terminal observations); std == 0 -> 1; shorter / longer statistics are padded / cut when allowed (223-239)
        self._norm = None
        if obs_norm is not None:
            if self.goal_env:
                raise NotImplementedError("obs_norm with goal_env: the reference normalises flat observations only")
            mean, std = np.array(obs_norm["mean"], np.float64), np.array(obs_norm["std"], np.float64)
            k = len(self._cols)
            if mean.shape != (k,) or std.shape != (k,):
                if not obs_norm.get("allow_different_observation_shapes", False):
                    raise ValueError(f"obs_norm: statistics of length {mean.shape[0]} for an observation of length {k} (Environment and dataset observation space do not match!)")
                mean = np.concatenate([mean, np.zeros(max(0, k - len(mean)))])[:k]
                std = np.concatenate([std, np.ones(max(0, k - len(std)))])[:k]
            std[std == 0] = 1
            self._norm = (mean, std, obs_norm.get("squash_factor"))
            if self._norm[2] is not None:
                self.observation_space = _Box(-1.0, 1.0, shape=(k,), dtype=np.float32)
        # Monitor (SB3 [UPSTREAM]; utils/env_util_SB3.py:60-66 gives every worker <monitor_dir>/<rank>.monitor.csv): ONE csv for the batch, same header and
        # r,l,t rows (+ info_keywords columns), which stable_baselines3.common.monitor.load_results() reads like any other *.monitor.csv
        self._monitor = None
        if monitor_dir is not None:
            import json
            os.makedirs(monitor_dir, exist_ok=True)
            self._monitor_keys = tuple((monitor_kwargs or {}).get("info_keywords", ()))
            path = os.path.join(monitor_dir, f"hip_batch_{int(env_id0)}.monitor.csv")
            self._monitor = open(path, "w", newline="")
            self._monitor.write("#" + json.dumps({"t_start": self._t_start, "env_id": env_id, "n_envs": int(n_envs)}) + "\n")
            self._monitor.write(",".join(("r", "l", "t") + self._monitor_keys) + "\n")
            self._monitor.flush()

    # ---- VecEnv API -------------------------------------------------------------------------------------
    def reset(self):
        self._ep_ret[:] = 0
        self._ep_len[:] = 0
        full = np.asarray(self._backend.reset())
        self._last_full = full
        if self.expert_obs_keys is not None:
            self._expert_cur = np.array(full, copy=True)
        return self._view(full)

    def step_async(self, actions):
        if self._ik is not None:  # [dx, dy, dz, gripper] in the first four columns of the 7-wide action rows
            a4 = np.asarray(actions, np.float64).reshape(self.num_envs, 4)
            actions = np.zeros((self.num_envs, CONST["HRG_ACT_DIM"]))
            actions[:, :4] = a4
        actions = np.asarray(actions, np.float64).reshape(self.num_envs, CONST["HRG_ACT_DIM"])
        self._actions = actions
        self._backend.step_async(actions)

    def step_wait(self):
        obs, term_obs, reward, done, info = self._backend.step_wait()
        full = np.asarray(obs)
        self._last_full = full
        obs, reward = self._view(full), np.array(reward, copy=True)
        dones = np.asarray(done).astype(bool)
        self._ep_ret += reward
        self._ep_len += 1
        if (self._cp is not None or self._ik is not None) and self.info_dicts:
            self._actions = np.array(self._backend.executed_actions(), copy=True)
        if self.info_dicts:
            infos = self._make_infos(info, dones, term_obs)
        else:
            infos = [{} for _ in range(self.num_envs)]
            if self._monitor is not None:   # the Monitor csv does not depend on the per-env dicts: episode rows from the done mask and the info block
                self._monitor_rows(np.asarray(info), np.nonzero(dones)[0])
        if self.expert_obs_keys is not None:
            self._expert_cur = np.array(full, copy=True)   # after an auto-reset: the new episode's first observation (wrapper reset())
        self._ep_ret[dones] = 0
        self._ep_len[dones] = 0
        return obs, reward, dones, infos

    def _make_infos(self, info, dones, term_obs):
        # the per-env dicts are filled from a copy of the info block on first use (LazyInfo)
        info = np.array(info, copy=True)
        src = _InfoSource(info, self._actions, self.expert_obs_keys, self._expert_cur, np.array(term_obs, copy=True) if self.expert_obs_keys is not None else None,
                          keys=self._info_keys)
        n = self.num_envs
        new = LazyInfo.__new__
        infos = [new(LazyInfo) for _ in range(n)]
        for i, d in enumerate(infos):
            d._src = src
            d._i = i
        idx = np.nonzero(dones)[0]
        if len(idx):
            now = round(time.time() - self._t_start, 6)
            trunc = info[idx, INFO_KEYS.index("TimeLimit.truncated")] != 0
            set_ = dict.__setitem__
            for i, tr in zip(idx.tolist(), trunc.tolist()):
                d = infos[i]
                set_(d, "TimeLimit.truncated", tr)
                set_(d, "terminal_observation", self._view(np.array(term_obs[i])))
                set_(d, "episode", {"r": float(self._ep_ret[i]), "l": int(self._ep_len[i]), "t": now})
            if self._monitor is not None:
                self._monitor_rows(info, idx, now)
        return infos

    def _monitor_rows(self, info, idx, now=None):
        """One r,l,t(+info_keywords) row per finished episode (SB3 Monitor [UPSTREAM]); info_keywords are columns of the kernel's info block."""
        if not len(idx):
            return
        now = round(time.time() - self._t_start, 6) if now is None else now
        for i in idx.tolist():
            extra = []
            keys = list(self._info_keys)
            for k in self._monitor_keys:
                if k not in keys:
                    raise KeyError(f"Monitor info_keywords: {k!r} is not a column of the info block ({keys})")
                v = int(info[i, keys.index(k)])
                extra.append(str(bool(v)) if k in _BOOL_KEYS else str(v))
            self._monitor.write(",".join([f"{round(float(self._ep_ret[i]), 6)}", str(int(self._ep_len[i])), str(now)] + extra) + "\n")
        self._monitor.flush()

    def close(self):
        if getattr(self, "_monitor", None) is not None:
            self._monitor.close()
            self._monitor = None
        self._backend.close()

    def seed(self, seed=None):
        """Re-key the per-env random streams (takes effect at the next reset by rebuilding the batch)."""
        if seed is None:
            return [None] * self.num_envs
        self.env_kwargs["seed"] = int(seed)
        self._desc = build_model_desc(self.env_kwargs, n_clips=self._clips.n_clips, collision_prevention=self._cp, goal_check=self._goal_check,
                                      env_id=self.env_id, ik_position_delta=self._ik, reach_box=self._reach_box, robot_geometry=self._robot_geometry)
        if isinstance(self._backend, _TorchBackend):
            self._backend.close()
            self._backend = _TorchBackend(self._desc, self._clips, self.num_envs, self._env_id0, self._device)
        else:
            self._backend.reseed(self._desc)
        return [int(seed) + i for i in range(self.num_envs)]

    def get_attr(self, attr_name, indices=None):
        idx = self._indices(indices)
        if attr_name in ("horizon", "env_kwargs"):
            return [getattr(self, attr_name)] * len(idx)
        if attr_name == "joint_pos":   # env.robots[0].controller.joint_pos (ik_position_delta_wrapper.py:107): the arm's joint angles after the last step / reset
            if self._last_full is None:
                raise RuntimeError("joint_pos: call reset() first")
            return [np.array(self._last_full[i, 18:24], np.float64) for i in idx]
        raise AttributeError(f"HipVecEnv has no per-env attribute {attr_name!r}")

    # HumanEnv.get_environment_state / set_environment_state (human_env.py:1845-1900; used by the dataset / reference-state-initialisation wrappers):
    # the stepper's state blocks, batched.  Each entry is (hrg_env_state, hrg_box_state or None); a restored episode keeps its own random streams.
    def get_environment_state(self, indices=None):
        batch = getattr(self._backend, "batch", None)
        if batch is None or not hasattr(batch, "get_states"):
            raise NotImplementedError("get_environment_state needs the HIP batch backend")
        idx = self._indices(indices)
        states, boxes = batch.get_states(np.asarray(idx, np.int32))
        if self.env_id == "CollaborativeStackingCart":   # CollaborativeStackingEnvState (collaborative_stacking_cartesian_env.py:63-97): the four cubes + stack bookkeeping
            return [(st, batch.get_stack(i)) for st, i in zip(states, idx)]
        if self.env_id == "CollaborativeHammeringCart":  # CollaborativeHammeringEnvState (collaborative_hammering_cartesian_env.py:56-89): board, hammer, nail + bookkeeping
            return [(st, batch.get_hammer(i)) for st, i in zip(states, idx)]
        has_box = self._reach_box or self.env_id != "ReachHuman"   # ReachHuman with its smallBox carries the object block too
        return [(st, boxes[k] if has_box else None) for k, st in enumerate(states)]

    def set_environment_state(self, states, indices=None):
        batch = getattr(self._backend, "batch", None)
        if batch is None or not hasattr(batch, "set_states"):
            raise NotImplementedError("set_environment_state needs the HIP batch backend")
        idx = self._indices(indices)
        if len(states) != len(idx):
            raise ValueError(f"{len(states)} states for {len(idx)} envs")
        from ._cstruct import BoxState, EnvState
        st_arr = (EnvState * len(idx))(*[st for st, _ in states])
        if self.env_id in ("CollaborativeStackingCart", "CollaborativeHammeringCart"):
            batch.set_states(np.asarray(idx, np.int32), st_arr, None)
            for i, (_, sk) in zip(idx, states):
                (batch.set_stack if self.env_id == "CollaborativeStackingCart" else batch.set_hammer)(i, sk)
            return
        boxes = [b for _, b in states]
        bx_arr = (BoxState * len(idx))(*boxes) if all(b is not None for b in boxes) and (self._reach_box or self.env_id != "ReachHuman") else None
        batch.set_states(np.asarray(idx, np.int32), st_arr, bx_arr)

    def check_collision_action(self, actions):
        """HumanEnv.check_collision_action (human_env.py:588-627) for every env: bool array, True where the joint-space action would drive the robot
        into the static scene or itself.  Nothing is stepped."""
        batch = getattr(self._backend, "batch", None)
        if batch is None or not hasattr(batch, "check_actions"):
            raise NotImplementedError("check_collision_action needs the HIP batch backend")
        a = np.asarray(actions, np.float64).reshape(self.num_envs, CONST["HRG_ACT_DIM"])
        import torch
        return batch.check_actions(torch.from_numpy(np.ascontiguousarray(a))).cpu().numpy().astype(bool)

    def set_attr(self, attr_name, value, indices=None):
        raise NotImplementedError("per-env attributes are fixed at construction (hrg_model_desc)")

    def _view(self, full):
        """Policy view of rows of the observation superset: flat array, or the goal-env dict."""
        full = np.asarray(full)
        if not self.goal_env:
            v = full[..., self._cols]
            if getattr(self, "_norm", None) is not None:
                mean, std, squash = self._norm
                v = (v - mean) / std
                if squash is not None:
                    v = np.tanh(squash * v)
                v = v.astype(np.float32)
            return v
        return {"observation": full[..., self._cols], "achieved_goal": full[..., self._ag_cols], "desired_goal": full[..., self._dg_cols]}

    def compute_reward(self, achieved_goal, desired_goal, info):
        """GoalEnvironmentGymWrapper.compute_reward -> HumanEnv._compute_reward (human_env.py:629-664, 766-792), vectorised: sparse task
        reward (+ 1 + dense reward when shaping), collision penalty from info["collision_type"], reward scale.  `info` is one dict or a
        sequence of dicts (as SB3's HerReplayBuffer passes them)."""
        if not self.goal_env:
            raise NotImplementedError("compute_reward: construct with goal_env=True / make_vec_env(type='goal_env')")
        d = self._desc
        ag, dg = np.atleast_2d(np.asarray(achieved_goal, np.float64)), np.atleast_2d(np.asarray(desired_goal, np.float64))
        infos = [info] if isinstance(info, dict) else list(info)
        ctype = np.array([int(i.get("collision_type", 0)) for i in infos])
        if self.env_id == "ReachHuman":
            dist = np.linalg.norm(ag - dg, axis=-1)
            r = np.where(dist <= d.goal_dist, d.task_reward, -1.0)
            dense = -0.1 * dist
        else:
            e2o, o2t = np.linalg.norm(ag[:, 3:6] - ag[:, 0:3], axis=-1), np.linalg.norm(dg - ag[:, 3:6], axis=-1)
            r = np.where(o2t <= d.goal_dist, d.task_reward, np.where(ag[:, 6] != 0, d.object_gripped_reward, -1.0))
            dense = -(e2o * 0.2 + o2t) * 0.1
        if d.reward_shaping:
            r = r + 1.0 + dense
        illegal = (ctype & (CONST["HRG_COL_STATIC"] | CONST["HRG_COL_ROBOT"] | CONST["HRG_COL_HUMAN_CRIT"])) != 0
        r = (r + np.where(illegal, d.collision_reward, 0.0)) * d.reward_scale
        return float(r[0]) if isinstance(info, dict) and np.ndim(achieved_goal) == 1 else r.astype(np.float32)

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        if method_name == "check_collision_action":   # per-env call of the reference: env_method("check_collision_action", action, indices=[i])
            idx = self._indices(indices)
            acts = np.zeros((self.num_envs, CONST["HRG_ACT_DIM"]))
            acts[idx] = np.asarray(method_args[0], np.float64)
            return [bool(x) for x in self.check_collision_action(acts)[idx]]
        if method_name == "compute_reward":   # SB3 HerReplayBuffer: env_method("compute_reward", achieved, desired, infos, indices=[0])
            return [self.compute_reward(*method_args, **method_kwargs) for _ in self._indices(indices)]
        raise NotImplementedError(f"env_method({method_name!r}) is not available on the batched stepper")

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * len(self._indices(indices))

    def get_images(self):
        raise NotImplementedError("rendering is out of scope")

    def render(self, mode="human"):
        raise NotImplementedError("rendering is out of scope")

    def _indices(self, indices):
        if indices is None:
            return list(range(self.num_envs))
        if isinstance(indices, int):
            return [indices]
        return list(indices)


class HipGymEnv:
    """Single ReachHuman env with the gym-0.21 API: `reset() -> obs`, `step(a) -> (obs, reward, done, info)`.

    Stepping a finished episode raises ValueError like HumanEnv.step (human_env.py:487-488)."""

    def __init__(self, env_kwargs=None, seed=None, clips=None, device=0, backend=None, obs_keys=None, collision_prevention=None,
                 env_id="ReachHuman"):
        self._vec = HipVecEnv(1, env_id=env_id, env_kwargs=env_kwargs, seed=seed, clips=clips, device=device, backend=backend, obs_keys=obs_keys,
                              collision_prevention=collision_prevention)
        self.observation_space = self._vec.observation_space
        self.action_space = self._vec.action_space
        self._done = True
        self._next_obs = None

    def reset(self):
        if self._next_obs is not None:  # the kernel already reset the env when the episode ended
            obs, self._next_obs = self._next_obs, None
        else:
            obs = self._vec.reset()[0]
        self._done = False
        return obs

    def step(self, action):
        if self._done:
            raise ValueError("executing action in terminated episode")
        obs, rew, done, infos = self._vec.step(np.asarray(action, np.float64)[None])
        info = infos[0]
        if done[0]:
            self._done = True
            self._next_obs = obs[0]
            return info["terminal_observation"], float(rew[0]), True, info
        return obs[0], float(rew[0]), False, info

    def observation_dict(self, obs):
        """Split a flat observation back into the reference's observable / modality keys."""
        out, k0 = OrderedDict(), 0
        for key in self._vec.obs_keys:
            n = len(OBS_COLUMNS[key])
            out[key] = obs[k0:k0 + n]
            k0 += n
        return out

    def close(self):
        self._vec.close()
