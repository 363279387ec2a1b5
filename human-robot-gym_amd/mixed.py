"""Mixed-task batches: several tasks of the reference's experiment suite stepped side by side on one GPU.

The reference runs its ICRA-2024 task suite as separate trainings, one SubprocVecEnv per task
(`training/icra_2024_run_experiments.sh:4-9`); BASELINE.json's last configuration batches the tasks together.  A task is a
compile-time kernel variant here (hrg_step_kernel / _box / _ho), so a mixed batch is one `HipBatch` per task writing row slices of
ONE packed output block (one all-gather for N > 1 GPUs, one D2H copy for the VecEnv), each launched on its own HIP stream so that
the variants fill the chip together instead of one after the other."""
import numpy as np

from ._cstruct import CONST
from .animation import synthetic_clips
from .model import build_model_desc

# training/icra_2024_run_experiments.sh:4-9 (horizons as there) with the environment blocks of training/config_icra_2024/environment_evaluation/training/<task>-SAC.yaml: where
# those differ from the task's general training config (ENV_DEFAULTS) the experiment's value is listed here (tests/test_assets.py compares every common keyword)
ICRA_TASKS = (
    ("ReachHuman", dict(horizon=100, shield_type="SSM", reward_shaping=True)),
    ("PickPlaceHumanCart", dict(horizon=1000, shield_type="SSM", done_at_success=True)),
    ("CollaborativeLiftingCart", dict(horizon=5000, shield_type="SSM")),
    ("RobotHumanHandoverCart", dict(horizon=1000, shield_type="PFL", collision_reward=0)),
    ("HumanRobotHandoverCart", dict(horizon=1000, shield_type="PFL", collision_reward=0)),
    ("CollaborativeStackingCart", dict(horizon=3000, shield_type="SSM")),
)


# ... and with the one task of the reference that is not part of that suite (CollaborativeHammeringCart has no experiment config): every task the stepper covers
ALL_TASKS = ICRA_TASKS + (("CollaborativeHammeringCart", dict(horizon=1000, shield_type="SSM")),)
# ms per 4096-env step of each task's kernel alone (profiles/r03t_tasks_summary.md): the mixed batch launches its kernels longest first.  The batch is bound by LDS residency
# (the tasks' LDS images exceed the chip's 41 MB), so the order matters by 5 - 10 %, but which order wins did not reproduce from one GPU box to the next (round 2: a search over
# all 720 orders and an A/B/A/B run, "short kernels first" 5.33 vs 5.60 ms on one box and 5.95 vs 5.71 ms on another).  Longest first is the order that was never the worst.
_STEP_MS = {"CollaborativeHammeringCart": 11.5, "CollaborativeStackingCart": 6.27, "RobotHumanHandoverCart": 4.28, "HumanRobotHandoverCart": 3.93, "CollaborativeLiftingCart": 2.66,
            "HumanObjectInspectionCart": 2.43, "PickPlaceHumanCart": 1.67, "ReachHuman": 1.00}


def task_clips(env_id, n_clips=13, seed=0, **kw):
    """Synthetic clip set carrying the animation info `env_id` reads."""
    extra = {"HumanObjectInspectionCart": dict(inspection=True), "HumanRobotHandoverCart": dict(handover=True),
             "RobotHumanHandoverCart": dict(handover="r2h"), "CollaborativeStackingCart": dict(stacking=True),
             }.get(env_id, {})
    if env_id == "CollaborativeHammeringCart":
        from .animation import HAMMERING_HANDS
        extra = dict(hammering=HAMMERING_HANDS)
    if env_id == "CollaborativeLiftingCart":
        from .animation import lifting_hands_nominal
        extra = dict(lifting=lifting_hands_nominal(build_model_desc(None, env_id=env_id)), fps=20.0)
    return synthetic_clips(n_clips, seed=seed, **extra, **kw)


def task_env_kwargs(env_id):
    """Environment keyword arguments that go with the synthetic clips of `task_clips` (the reference's defaults belong to its recorded clips)."""
    if env_id == "CollaborativeHammeringCart":
        from .animation import hammering_relquat
        return dict(hammer_weld_relquat=hammering_relquat())
    return {}


def split_evenly(n_envs, n_tasks):
    """Env counts per task: as even as possible, the remainder to the first tasks."""
    q, r = divmod(int(n_envs), int(n_tasks))
    return [q + (1 if i < r else 0) for i in range(n_tasks)]


class MixedBatch:
    """`parts`: list of (env_id, desc, clips, n_envs).  Rows of every output are ordered part by part; `slices[i]` is part i's row range."""

    def __init__(self, parts, env_id0=0, device=0, concurrent=True):
        import torch
        from ._lib import HipBatch
        from .dist import packed_layout
        if not torch.cuda.is_available():
            raise RuntimeError("MixedBatch needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.n = sum(int(p[3]) for p in parts)
        n, od, idim = self.n, CONST["HRG_OBS_DIM"], CONST["HRG_INFO_DIM"]
        lay = packed_layout(n)
        offs, sizes = lay["offsets"], lay["sizes"]
        self.packed = torch.zeros(lay["total"], dtype=torch.uint8, device=self.device)
        self.packed_head = self.packed[:lay["head"]]
        self.packed_layout = dict(offsets=offs, sizes=sizes)
        self.obs = self.packed[offs[0]:offs[0] + sizes[0]].view(torch.float32).view(n, od)
        self.term_obs = self.packed[offs[1]:offs[1] + sizes[1]].view(torch.float32).view(n, od)
        self.reward = self.packed[offs[2]:offs[2] + sizes[2]].view(torch.float32)
        self.info = self.packed[offs[3]:offs[3] + sizes[3]].view(torch.int32).view(n, idim)
        self.done = self.packed[offs[4]:offs[4] + sizes[4]]
        self.env_ids, self.slices, self.batches = [], [], []
        r0 = 0
        for env_id, desc, clips, k in parts:
            sl = slice(r0, r0 + int(k))
            out = (self.obs[sl], self.term_obs[sl], self.reward[sl], self.info[sl], self.done[sl])
            self.batches.append(HipBatch(desc, clips, int(k), env_id0=env_id0 + r0, device=device, out=out))
            self.env_ids.append(env_id)
            self.slices.append(sl)
            r0 += int(k)
        self.concurrent = bool(concurrent) and len(parts) > 1
        # launch the long kernels first: the step ends with the last kernel to finish, and a kernel launched late queues behind the others for wave slots and LDS (see _STEP_MS)
        self._launch_order = sorted(range(len(parts)), key=lambda i: -_STEP_MS.get(parts[i][0], 2.1) * int(parts[i][3]))
        with torch.cuda.device(self.device):
            self.streams = [torch.cuda.Stream() for _ in parts] if self.concurrent else None   # (high-priority streams for the long kernels: no effect, 8.46 vs 8.47 ms)

    def _each(self, fn):
        t = self.torch
        if not self.concurrent:
            for i, b in enumerate(self.batches):
                fn(i, b)
            return
        cur = t.cuda.current_stream(self.device)
        for i in self._launch_order:
            b, s = self.batches[i], self.streams[i]
            s.wait_stream(cur)          # inputs written on the caller's stream
            with t.cuda.stream(s):
                fn(i, b)
        for s in self.streams:
            cur.wait_stream(s)          # outputs are consumed on the caller's stream

    def reset(self, mask=None):
        self._each(lambda i, b: b.reset(None if mask is None else mask[self.slices[i]]))
        return self.obs

    def step(self, actions):
        """actions: float64 [n, 7] on the device (rows in part order).  Returns (obs, reward, done, info) views of the shared block."""
        t = self.torch
        if actions.dtype != t.float64 or actions.device != self.device or not actions.is_contiguous():
            actions = actions.to(device=self.device, dtype=t.float64).contiguous()
        if tuple(actions.shape) != (self.n, CONST["HRG_ACT_DIM"]):
            raise ValueError(f"actions must be [{self.n}, {CONST['HRG_ACT_DIM']}]")
        self._each(lambda i, b: b.step(actions[self.slices[i]]))
        self._keep = actions
        return self.obs, self.reward, self.done, self.info

    def kernel_time(self):
        """(sum over the parts of their average step-kernel ms, launches per part) since the previous call; the first call arms the timers.
        With `concurrent` the kernels overlap, so the sum exceeds the step's wall time."""
        per = [b.kernel_time() for b in self.batches]
        return sum(ms for ms, _ in per), (per[0][1] if per else 0)

    def part_of(self, env):
        """(part index, row within the part) of global row `env`."""
        for i, sl in enumerate(self.slices):
            if sl.start <= env < sl.stop:
                return i, env - sl.start
        raise IndexError(env)

    def get_state(self, e):
        i, r = self.part_of(e)
        return self.batches[i].get_state(r)

    def close(self):
        for b in self.batches:
            b.close()
        self.batches = []


def make_mixed_batch(n_envs, tasks=ICRA_TASKS, env_kwargs=None, clips=None, n_clips=13, seed=None, env_id0=0, device=0, concurrent=True):
    """`n_envs` environments split evenly over `tasks` = [(env_id, kwargs), ...]; `env_kwargs` are applied to every task first."""
    counts = split_evenly(n_envs, len(tasks))
    parts = []
    for (env_id, kw), k in zip(tasks, counts):
        if k == 0:
            continue
        kw = dict(task_env_kwargs(env_id), **dict(env_kwargs or {}, **kw))
        if seed is not None:
            kw["seed"] = int(seed)
        c = (clips or {}).get(env_id) if isinstance(clips, dict) else None
        c = c if c is not None else task_clips(env_id, n_clips)
        parts.append((env_id, build_model_desc(kw, n_clips=c.n_clips, env_id=env_id), c, k))
    return MixedBatch(parts, env_id0=env_id0, device=device, concurrent=concurrent)


class _MixedBackend:
    """numpy <-> MixedBatch adapter with the interface of vec_env._TorchBackend (one H2D action copy, one D2H copy of the block)."""

    def __init__(self, batch):
        import torch
        self.torch = torch
        self.batch = batch
        n = self.n = batch.n
        lay = batch.packed_layout
        self._host = torch.empty(batch.packed.numel(), dtype=torch.uint8, pin_memory=True)
        o, s = lay["offsets"], lay["sizes"]
        hb = self._host.numpy()
        od, idim = CONST["HRG_OBS_DIM"], CONST["HRG_INFO_DIM"]
        self.obs = hb[o[0]:o[0] + s[0]].view(np.float32).reshape(n, od)
        self.term_obs = hb[o[1]:o[1] + s[1]].view(np.float32).reshape(n, od)
        self.reward = hb[o[2]:o[2] + s[2]].view(np.float32)
        self.info = hb[o[3]:o[3] + s[3]].view(np.int32).reshape(n, idim)
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
        return self._act.cpu().numpy()

    def close(self):
        self.batch.close()


def make_mixed_vec_env(n_envs, tasks=ICRA_TASKS, obs_keys=None, env_kwargs=None, seed=None, start_index=0, clips=None, n_clips=13,
                       device=0, info_dicts=True, concurrent=True):
    """A `HipVecEnv`-shaped VecEnv over a mixed batch.  One policy sees every task, so the observation is the same columns for all
    of them: `obs_keys` (names valid for every task) or, by default, the whole 64-column observation superset (columns a task does
    not fill are zero).  `infos[i]["task"]` names the task of row i; `env.task_slices` maps env ids to row ranges."""
    batch = make_mixed_batch(n_envs, tasks, env_kwargs=env_kwargs, clips=clips, n_clips=n_clips, seed=seed, env_id0=start_index,
                             device=device, concurrent=concurrent)
    return __getattr__("MixedHipVecEnv")(batch, obs_keys=obs_keys, info_dicts=info_dicts)


def _mixed_cls():
    from .vec_env import OBS_COLUMNS, HipVecEnv, _Box, _VecEnvBase
    import time

    class MixedHipVecEnv(HipVecEnv):
        """VecEnv over a `MixedBatch` (see `make_mixed_vec_env`).  Joint-space actions [n, 7] for every task."""

        def __init__(self, batch, obs_keys=None, info_dicts=True):
            self.env_id = "mixed(" + ",".join(batch.env_ids) + ")"
            self.task_slices = dict(zip(batch.env_ids, batch.slices))
            self.goal_env = False
            self.expert_obs_keys, self._expert_cur = None, None
            self._cp = self._ik = None
            if obs_keys is None:
                self.obs_keys = None
                self._cols = np.arange(CONST["HRG_OBS_DIM"], dtype=np.int64)
            else:
                unknown = [k for k in obs_keys if k not in OBS_COLUMNS]
                if unknown:
                    raise NotImplementedError(f"obs_keys {unknown!r}: available {sorted(OBS_COLUMNS)}")
                self.obs_keys = list(obs_keys)
                self._cols = np.array([c for k in obs_keys for c in OBS_COLUMNS[k]], dtype=np.int64)
            self._backend = _MixedBackend(batch)
            _VecEnvBase.__init__(self, batch.n, _Box(-np.inf, np.inf, shape=(len(self._cols),), dtype=np.float32),
                                 _Box(-1.0, 1.0, shape=(CONST["HRG_ACT_DIM"],), dtype=np.float32))
            self.info_dicts = info_dicts
            self._ep_ret = np.zeros(batch.n, np.float64)
            self._ep_len = np.zeros(batch.n, np.int64)
            self._t_start = time.time()
            self._actions = None
            self._last_full = None
            self._task_of_row = [eid for eid, sl in zip(batch.env_ids, batch.slices) for _ in range(sl.stop - sl.start)]

        def _make_infos(self, info, dones, term_obs):
            infos = super()._make_infos(info, dones, term_obs)
            for d, task in zip(infos, self._task_of_row):
                dict.__setitem__(d, "task", task)   # (not d[...] = ...: that would fill the lazy rows)
            return infos

        def seed(self, seed=None):
            raise NotImplementedError("re-seeding a mixed batch: build a new one with make_mixed_vec_env(seed=...)")

        def get_attr(self, attr_name, indices=None):
            if attr_name == "task":
                return [self._task_of_row[i] for i in self._indices(indices)]
            if attr_name == "joint_pos":
                return HipVecEnv.get_attr(self, attr_name, indices)
            raise AttributeError(f"MixedHipVecEnv has no per-env attribute {attr_name!r}")

        def compute_reward(self, *a, **k):
            raise NotImplementedError("goal-env relabelling is per task; use one HipVecEnv(goal_env=True) per task")

    return MixedHipVecEnv


def __getattr__(name):
    if name == "MixedHipVecEnv":
        cls = _mixed_cls()
        globals()["MixedHipVecEnv"] = cls
        return cls
    raise AttributeError(name)
