"""Shared helpers for the parity tests (oracle = checker, HIP library = unit under test)."""
import ctypes

import numpy as np

import human_robot_gym_amd as hrg
from human_robot_gym_amd._cstruct import CONST, struct_to_dict

# tolerance stated by BASELINE.json north_star: qpos/obs within 1e-5 relative
RTOL = 1e-5
ATOL = 1e-7


def record_live(test, live, floor):
    """Free-running comparisons drop an env once its oracle trajectory turns violent (|qvel| > 5 rad/s or a crash: chaotic from then on).  The fraction
    that stayed in is printed, appended to gpurun_out/parity_live.jsonl (so that the floors below are the levels the runs actually achieve) and
    asserted against `floor`."""
    import json
    import os
    frac = float(np.mean(live))
    print(f"[parity] {test}: {int(np.sum(live))}/{len(live)} envs compared to the end (live fraction {frac:.3f}, floor {floor})")
    try:
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_live.jsonl"), "a") as f:
            f.write(json.dumps(dict(test=test, live=frac, n=int(len(live)), floor=floor)) + "\n")
    except OSError:
        pass
    assert frac >= floor, f"{test}: too many envs dropped as chaotic: live fraction {frac:.3f} < {floor}"


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


def _ltt_samples(ltt):
    """A long-term trajectory as what it commands: (q, q', q'') of every joint at fractions of its duration.  The segment
    table itself has knife edges (a start-acceleration ramp exists only for |a0| > 1e-9, zero-length segments move
    between slots), so two tables that command the same motion need not be equal entry by entry."""
    from oracle.oracle import load
    lib = load()
    out = (ctypes.c_double * 3)()
    res = []
    for j in range(CONST["HRG_NARM"]):
        for f in (0.0, 0.15, 0.3, 0.5, 0.7, 0.85, 1.0, 1.2):
            lib.hrgo_test_ltt_eval(ctypes.byref(ltt), ctypes.c_int(j), ctypes.c_double(f * ltt.T), out)
            res += list(out)
    return np.array(res)


def assert_state_close(so, sg, what=""):
    names = ([], [])
    fo, io = flat_state(so, names)
    fg, ig = flat_state(sg)
    if hasattr(so, "ltt"):
        keep = np.array([not (nm.startswith("st.ltt.dur") or nm.startswith("st.ltt.jerk")) for nm in names[0]])
        fo, fg, names = fo[keep], fg[keep], ([nm for nm, k in zip(names[0], keep) if k], names[1])
        # sampled q'' values sit on piecewise-linear ramps of slope j_max = 15 rad/s^3: a 1e-6 s shift of a segment boundary moves them by 1.5e-5
        np.testing.assert_allclose(_ltt_samples(sg.ltt), _ltt_samples(so.ltt), rtol=1e-4, atol=2e-5, err_msg=f"long-term trajectory differs {what}")
    bad_i = [f"{names[1][k]}: oracle {io[k]} hip {ig[k]}" for k in np.nonzero(io != ig)[0][:12]]
    assert not bad_i, f"integer state differs {what}: {bad_i}"
    # segment durations of a planned profile are square roots of velocity differences: a ramp of ~1e-10 rad/s has a
    # duration of microseconds that moves by its own size under rounding-level state differences (and carries no motion)
    atol = np.array([2e-5 if (".dur[" in nm or nm == "st.ltt.T") else ATOL for nm in names[0]])
    bad = np.nonzero(~(np.abs(fg - fo) <= atol + RTOL * np.abs(fo)))[0]
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
