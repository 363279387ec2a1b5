"""Shared helpers for the parity tests (oracle = checker, HIP library = unit under test)."""
import ctypes

import numpy as np

import human_robot_gym_amd as hrg
from human_robot_gym_amd._cstruct import CONST, struct_to_dict

# tolerance stated by BASELINE.json north_star: qpos/obs within 1e-5 relative
RTOL = 1e-5
ATOL = 1e-7


def make_pair(n_envs, env_kwargs=None, n_clips=3, clip_seed=0, env_id0=0, clips=None, **desc_kw):
    """(OracleBatch, HipBatch) on identical model / clips / seeds."""
    from oracle.oracle import OracleBatch
    from human_robot_gym_amd._lib import HipBatch
    clips = clips or hrg.synthetic_clips(n_clips, seed=clip_seed, min_frames=300, max_frames=600)
    d1 = hrg.build_model_desc(env_kwargs, n_clips=clips.n_clips, **desc_kw)
    d2 = hrg.build_model_desc(env_kwargs, n_clips=clips.n_clips, **desc_kw)
    return OracleBatch(d1, clips, n_envs, env_id0), HipBatch(d2, clips, n_envs, env_id0)


def flat_state(s, names=None):
    """hrg_env_state -> (float vector, int vector) for comparisons."""
    d = struct_to_dict(s)
    fl, it = [], []

    def walk(x, path):
        if isinstance(x, dict):
            for k, v in x.items():
                walk(v, path + "." + k)
        elif isinstance(x, list):
            for i, v in enumerate(x):
                walk(v, f"{path}[{i}]")
        elif isinstance(x, float):
            fl.append(x)
            if names is not None:
                names[0].append(path)
        else:
            it.append(int(x))
            if names is not None:
                names[1].append(path)
    walk(d, "st")
    return np.array(fl), np.array(it)


def assert_state_close(so, sg, what=""):
    names = ([], [])
    fo, io = flat_state(so, names)
    fg, ig = flat_state(sg)
    bad_i = [f"{names[1][k]}: oracle {io[k]} hip {ig[k]}" for k in np.nonzero(io != ig)[0][:12]]
    assert not bad_i, f"integer state differs {what}: {bad_i}"
    bad = np.nonzero(~np.isclose(fg, fo, rtol=RTOL, atol=ATOL))[0]
    bad_f = [f"{names[0][k]}: oracle {fo[k]!r} hip {fg[k]!r}" for k in bad[:12]]
    assert not bad_f, f"float state differs {what}: {bad_f}"


class OracleBackend:
    """HipVecEnv backend interface served by the CPU oracle (host-logic tests without a GPU)."""

    def __init__(self, desc, clips, n_envs, env_id0=0):
        from oracle.oracle import OracleBatch
        self.B = OracleBatch(desc, clips, n_envs, env_id0)

    def reset(self):
        return self.B.reset()

    def step_async(self, actions):
        self._out = self.B.step(actions)

    def step_wait(self):
        obs, rew, done, info = self._out
        return obs, self.B.term_obs.copy(), rew, done, info

    def executed_actions(self):
        return self.B.last_actions

    def close(self):
        self.B.close()
