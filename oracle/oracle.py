"""ctypes wrapper of the CPU oracle (oracle/hrg_oracle.c).  TEST INFRASTRUCTURE ONLY — imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product package."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libhrg_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "hrg_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def load():
    if not os.path.exists(_SO):
        build()
    lib = ctypes.CDLL(_SO)
    lib.hrgo_state_bytes.restype = ctypes.c_size_t
    lib.hrgo_desc_bytes.restype = ctypes.c_size_t
    lib.hrgo_test_segseg.restype = ctypes.c_double
    lib.hrgo_test_u01.restype = ctypes.c_double
    lib.hrgo_test_u01.argtypes = [ctypes.c_uint64] * 5
    lib.hrgo_test_path.argtypes = [ctypes.c_double] * 7 + [ctypes.c_void_p]
    lib.hrgo_test_ltt_eval.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
    lib.hrgo_test_path_eval.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]
    return lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class OracleBatch:
    """Same surface as human_robot_gym_amd._lib.HipBatch, on numpy arrays."""

    def __init__(self, desc, clips, n_envs, env_id0=0):
        from human_robot_gym_amd._cstruct import CONST, EnvState
        self.C = CONST
        self.EnvState = EnvState
        self.lib = load()
        assert self.lib.hrgo_state_bytes() == ctypes.sizeof(EnvState)
        assert self.lib.hrgo_desc_bytes() == ctypes.sizeof(type(desc))
        self.n = n_envs
        self._clips = clips
        self._table = clips.table()
        self.h = ctypes.c_void_p()
        rc = self.lib.hrgo_create(ctypes.byref(desc), ctypes.byref(self._table), ctypes.c_int32(n_envs), ctypes.c_int64(env_id0), ctypes.byref(self.h))
        assert rc == 0
        C = CONST
        self.obs = np.zeros((n_envs, C["HRG_OBS_DIM"]), np.float32)
        self.term_obs = np.zeros((n_envs, C["HRG_OBS_DIM"]), np.float32)
        self.reward = np.zeros(n_envs, np.float32)
        self.done = np.zeros(n_envs, np.uint8)
        self.info = np.zeros((n_envs, C["HRG_INFO_DIM"]), np.int32)

    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self.lib.hrgo_reset(self.h, None if m is None else _p(m), _p(self.obs))
        return self.obs.copy()

    def step(self, actions):
        a = np.ascontiguousarray(actions, np.float64)
        assert a.shape == (self.n, self.C["HRG_ACT_DIM"])
        self.last_actions = a  # rewritten in place by the collision-prevention screening
        self.lib.hrgo_step(self.h, _p(a), _p(self.obs), _p(self.term_obs), _p(self.reward), _p(self.done), _p(self.info))
        return self.obs.copy(), self.reward.copy(), self.done.copy(), self.info.copy()

    def step_range(self, e0, e1, actions):
        a = np.ascontiguousarray(actions, np.float64)
        self.lib.hrgo_step_range(self.h, ctypes.c_int(e0), ctypes.c_int(e1), _p(a), _p(self.obs), _p(self.reward), _p(self.done), _p(self.info))

    def rollout_parallel(self, pool, n_steps, n_workers, cpus=None):
        """n_steps vec-steps on n_workers pthreads (barrier per vec-step), actions cycled from pool [n_pool, n, 7]: the CPU-baseline harness."""
        pool = np.ascontiguousarray(pool, np.float64)
        assert pool.ndim == 3 and pool.shape[1:] == (self.n, self.C["HRG_ACT_DIM"])
        cp = None if cpus is None else np.ascontiguousarray(cpus, np.int32)
        assert cp is None or len(cp) >= n_workers
        rc = self.lib.hrgo_rollout_parallel(self.h, ctypes.c_int(n_workers), None if cp is None else _p(cp), ctypes.c_int(n_steps), _p(pool), ctypes.c_int(len(pool)),
                                            _p(self.obs), _p(self.reward), _p(self.done), _p(self.info))
        assert rc == 0, rc

    def rollout_parallel2(self, pool, n_steps, n_workers, cpus=None, chunk=8):
        """rollout_parallel with the envs of a vec-step drawn in runs of `chunk` from a shared counter (0: fixed ranges); returns the seconds each worker
        spent stepping envs (the rest of the wall time it waited at the per-step barrier)."""
        pool = np.ascontiguousarray(pool, np.float64)
        assert pool.ndim == 3 and pool.shape[1:] == (self.n, self.C["HRG_ACT_DIM"])
        cp = None if cpus is None else np.ascontiguousarray(cpus, np.int32)
        assert cp is None or len(cp) >= n_workers
        busy = np.zeros(n_workers, np.float64)
        rc = self.lib.hrgo_rollout_parallel2(self.h, ctypes.c_int(n_workers), None if cp is None else _p(cp), ctypes.c_int(n_steps), _p(pool), ctypes.c_int(len(pool)),
                                             ctypes.c_int(chunk), _p(self.obs), _p(self.reward), _p(self.done), _p(self.info), _p(busy))
        assert rc == 0, rc
        return busy

    def step_parallel(self, actions, n_workers=None):
        """`step` on n_workers threads (default: the CPUs of the affinity mask, at most 32): parity tests at the benchmark's batch sizes."""
        import os
        a = np.ascontiguousarray(actions, np.float64)
        assert a.shape == (self.n, self.C["HRG_ACT_DIM"])
        self.last_actions = a
        nw = n_workers or max(1, min(32, len(os.sched_getaffinity(0))))
        rc = self.lib.hrgo_step_parallel(self.h, ctypes.c_int(nw), _p(a), _p(self.obs), _p(self.term_obs), _p(self.reward), _p(self.done), _p(self.info))
        assert rc == 0, rc
        return self.obs.copy(), self.reward.copy(), self.done.copy(), self.info.copy()

    def get_states_all(self, box=False, stack=False, hammer=False):
        """(EnvState[n], BoxState[n] | None, StackState[n] | None, HammerState[n] | None): every state block of the batch in one call."""
        from human_robot_gym_amd._cstruct import BoxState, StackState, HammerState
        st = (self.EnvState * self.n)()
        bx = (BoxState * self.n)() if box else None
        sk = (StackState * self.n)() if stack else None
        hm = (HammerState * self.n)() if hammer else None
        ref = lambda x: ctypes.byref(x) if x is not None else None
        assert self.lib.hrgo_get_states(self.h, ctypes.byref(st), ref(bx), ref(sk), ref(hm)) == 0
        return st, bx, sk, hm

    def set_states_all(self, st=None, bx=None, sk=None, hm=None):
        for x in (st, bx, sk, hm):
            assert x is None or len(x) == self.n
        ref = lambda x: ctypes.byref(x) if x is not None else None
        assert self.lib.hrgo_set_states(self.h, ref(st), ref(bx), ref(sk), ref(hm)) == 0

    def get_state(self, e):
        s = self.EnvState()
        assert self.lib.hrgo_get_state(self.h, ctypes.c_int(e), ctypes.byref(s), ctypes.c_size_t(ctypes.sizeof(s))) == 0
        return s

    def set_state(self, e, s):
        assert self.lib.hrgo_set_state(self.h, ctypes.c_int(e), ctypes.byref(s), ctypes.c_size_t(ctypes.sizeof(s))) == 0

    def get_box(self, e):
        from human_robot_gym_amd._cstruct import BoxState
        s = BoxState()
        assert self.lib.hrgo_get_box(self.h, ctypes.c_int(e), ctypes.byref(s), ctypes.c_size_t(ctypes.sizeof(s))) == 0
        return s

    def set_box(self, e, s):
        assert self.lib.hrgo_set_box(self.h, ctypes.c_int(e), ctypes.byref(s), ctypes.c_size_t(ctypes.sizeof(s))) == 0

    def get_hammer(self, e):
        from human_robot_gym_amd._cstruct import HammerState
        s = HammerState()
        assert self.lib.hrgo_get_hammer(self.h, ctypes.c_int(e), ctypes.byref(s), ctypes.c_size_t(ctypes.sizeof(s))) == 0
        return s

    def set_hammer(self, e, s):
        assert self.lib.hrgo_set_hammer(self.h, ctypes.c_int(e), ctypes.byref(s), ctypes.c_size_t(ctypes.sizeof(s))) == 0

    def get_stack(self, e):
        from human_robot_gym_amd._cstruct import StackState
        s = StackState()
        assert self.lib.hrgo_get_stack(self.h, ctypes.c_int(e), ctypes.byref(s), ctypes.c_size_t(ctypes.sizeof(s))) == 0
        return s

    def set_stack(self, e, s):
        assert self.lib.hrgo_set_stack(self.h, ctypes.c_int(e), ctypes.byref(s), ctypes.c_size_t(ctypes.sizeof(s))) == 0

    def check_actions(self, actions):
        a = np.ascontiguousarray(actions, np.float64)
        out = np.zeros(self.n, np.uint8)
        self.lib.hrgo_check_actions(self.h, _p(a), _p(out))
        return out

    def contacts(self):
        pairs = np.zeros((self.n, self.C["HRG_NCON_MAX"], 2), np.int32)
        ncon = np.zeros(self.n, np.int32)
        self.lib.hrgo_contacts(self.h, _p(pairs), _p(ncon))
        return pairs, ncon

    def capsules(self):
        r = np.zeros((self.n, self.C["HRG_NSHIELD_RCAP"], 7))
        h = np.zeros((self.n, self.C["HRG_NHCAP_MAX"], 7))
        nh = np.zeros(self.n, np.int32)
        self.lib.hrgo_capsules(self.h, _p(r), _p(h), _p(nh))
        return r, h, nh

    def close(self):
        if self.h:
            self.lib.hrgo_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
