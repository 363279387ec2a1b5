"""Shared helpers for the parity tests (oracle = checker, HIP library = unit under test)."""
import ctypes

import numpy as np

import human_robot_gym_amd as hrg
from human_robot_gym_amd._cstruct import CONST, struct_to_dict

# tolerance stated by BASELINE.json north_star: qpos/obs within 1e-5 relative
RTOL = 1e-5
ATOL = 1e-7


def record_live(test, live, floor, **extra):
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
            f.write(json.dumps(dict(test=test, live=frac, n=int(len(live)), floor=floor, **extra)) + "\n")
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


def _ltt_samples(ltt, T):
    """A long-term trajectory as what it commands: (q, q', q'') of every joint at fixed times (fractions of the ORACLE's duration T, the same absolute times
    for both sides).  The segment table itself has knife edges (a start-acceleration ramp exists only for |a0| > 1e-9, zero-length segments move between
    slots), so two tables that command the same motion need not be equal entry by entry."""
    from oracle.oracle import load
    lib = load()
    out = (ctypes.c_double * 3)()
    res = []
    for j in range(CONST["HRG_NARM"]):
        for f in (0.0, 0.15, 0.3, 0.5, 0.7, 0.85, 1.0, 1.2):
            lib.hrgo_test_ltt_eval(ctypes.byref(ltt), ctypes.c_int(j), ctypes.c_double(f * T), out)
            res += list(out)
    return np.array(res).reshape(-1, 3)


def _path_samples(path, T):
    """A fail-safe / recovery profile of the path parameter as what it commands: (s, s', s'') at fixed times within the ORACLE's duration T.  Its three
    (duration, jerk) phases are not unique either: a single ramp can sit in the first or in the last slot with an empty phase of the opposite sign beside it."""
    from oracle.oracle import load
    lib = load()
    out = (ctypes.c_double * 3)()
    res = []
    for f in (0.0, 0.2, 0.4, 0.6, 0.8, 0.98):   # (beyond T the profile continues at the model's fail-safe speed, which the state block does not carry)
        lib.hrgo_test_path_eval(ctypes.byref(path), ctypes.c_double(f * T), ctypes.c_double(0.0), out)
        res.append(list(out))
    return np.array(res)


def assert_state_close(so, sg, what=""):
    """Integers bit-exact; floats within the north-star tolerance (1e-5 relative, ATOL absolute).  Exceptions, each a knife edge of a REPRESENTATION, not of
    the motion it stands for (the motion itself is compared at 1e-5):
      * ltt.dur / ltt.jerk / safe_path.dur / safe_path.jerk are compared through the trajectory they define, sampled at fixed times (see above);
      * the sampled second derivatives sit on ramps of slope j_max (15 rad/s^3 for the joints, path_jmax for s): a segment boundary that moves by dt moves
        them by j dt.  Boundaries are square roots of velocity differences, so rounding-level state differences (1e-16) at |dv| ~ 1e-10 move a boundary by
        ~1e-8 s: atol 1e-6 on q'' (5e-6 on s'', whose jerk is ~20 x larger);
      * ltt.T (the sum of the durations) changes by the length of a ramp that exists on one side only (|a0| just above / below 1e-9: 2 sqrt(1e-9 / j) ~ 1.6e-5 s)."""
    names = ([], [])
    fo, io = flat_state(so, names)
    fg, ig = flat_state(sg)
    if hasattr(so, "ltt"):
        rep = (".ltt.dur", ".ltt.jerk", ".safe_path.dur", ".safe_path.jerk")
        keep = np.array([not any(r in nm for r in rep) for nm in names[0]])
        fo, fg, names = fo[keep], fg[keep], ([nm for nm, k in zip(names[0], keep) if k], names[1])
        lo, lg = _ltt_samples(so.ltt, so.ltt.T), _ltt_samples(sg.ltt, so.ltt.T)
        np.testing.assert_allclose(lg[:, :2], lo[:, :2], rtol=RTOL, atol=ATOL, err_msg=f"long-term trajectory (q, q') differs {what}")
        np.testing.assert_allclose(lg[:, 2], lo[:, 2], rtol=RTOL, atol=1e-6, err_msg=f"long-term trajectory (q'') differs {what}")
        Tp = sum(so.safe_path.dur)
        po, pg = _path_samples(so.safe_path, Tp), _path_samples(sg.safe_path, Tp)
        np.testing.assert_allclose(pg[:, :2], po[:, :2], rtol=RTOL, atol=ATOL, err_msg=f"fail-safe path (s, s') differs {what}")
        np.testing.assert_allclose(pg[:, 2], po[:, 2], rtol=RTOL, atol=5e-6, err_msg=f"fail-safe path (s'') differs {what}")
    bad_i = [f"{names[1][k]}: oracle {io[k]} hip {ig[k]}" for k in np.nonzero(io != ig)[0][:12]]
    assert not bad_i, f"integer state differs {what}: {bad_i}"
    atol = np.array([2e-5 if nm == "st.ltt.T" else ATOL for nm in names[0]])
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


# ---------------------------------------------------------------------------------------------- bulk comparisons (benchmark-size batches)
def struct_leaves(t, base=0, path=""):
    """[(path, byte offset, numpy dtype, count)] of the scalar leaves of a ctypes struct type (arrays of scalars are one leaf)."""
    import ctypes
    out = []
    for name, ft in t._fields_:
        off = base + getattr(t, name).offset
        p = f"{path}.{name}" if path else name
        dims = 1
        et = ft
        while issubclass(et, ctypes.Array):
            dims *= et._length_
            et = et._type_
        if issubclass(et, ctypes.Structure):
            for k in range(dims):
                out += struct_leaves(et, off + k * ctypes.sizeof(et), f"{p}[{k}]" if dims > 1 or issubclass(ft, ctypes.Array) else p)
        else:
            dt = {ctypes.c_double: np.float64, ctypes.c_float: np.float32, ctypes.c_int32: np.int32, ctypes.c_uint32: np.uint32, ctypes.c_int64: np.int64,
                  ctypes.c_uint64: np.uint64, ctypes.c_uint8: np.uint8}[et]
            out.append((p, off, dt, dims))
    return out


def states_as_bytes(arr):
    """ctypes array of n structs -> uint8 [n, sizeof] view."""
    import ctypes
    n = len(arr)
    return np.frombuffer(arr, dtype=np.uint8).reshape(n, ctypes.sizeof(arr._type_))


def compare_states_bulk(so, sg, skip=(), only=None):
    """Two ctypes arrays of the same struct type, env by env: integers bit-exact, floats within RTOL / ATOL (ltt.T: atol 2e-5, see assert_state_close).
    Leaves whose path contains one of `skip` are left out; with `only`, just the leaves whose path contains one of its entries are compared.  Returns (ok [n] bool, first mismatch message or None)."""
    t = so._type_
    bo, bg = states_as_bytes(so), states_as_bytes(sg)
    n = len(so)
    ok = np.ones(n, bool)
    msg = None
    for path, off, dt, cnt in struct_leaves(t):
        if any(s in path for s in skip) or (only is not None and not any(s in path for s in only)):
            continue
        w = np.dtype(dt).itemsize * cnt
        a = np.ascontiguousarray(bo[:, off:off + w]).view(dt).reshape(n, cnt)
        b = np.ascontiguousarray(bg[:, off:off + w]).view(dt).reshape(n, cnt)
        if np.issubdtype(dt, np.floating):
            atol = 2e-5 if path.endswith("ltt.T") else ATOL
            good = np.all(np.abs(b - a) <= atol + RTOL * np.abs(a), axis=1)
        else:
            good = np.all(a == b, axis=1)
        if msg is None and not good.all():
            e = int(np.nonzero(~good)[0][0])
            msg = f"{path}: env {e}: oracle {a[e].tolist()} hip {b[e].tolist()}"
        ok &= good
    return ok, msg
