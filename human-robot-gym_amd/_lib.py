"""Thin ctypes binding of libhrgym_hip.so (include/hrgym.h) + `HipBatch`, a tensor-level handle.

There is NO CPU fallback: if the HIP library is missing or the GPU is unavailable this module raises.
PyTorch-ROCm is used only to own device buffers / streams (plumbing); all compute is in the HIP library.
"""
import ctypes
import os
import subprocess

from ._cstruct import CONST, EnvState, BoxState, StackState, HammerState, ModelDesc, ClipTable

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhrgym_hip.so")   # the one shipping library; no environment variable redirects it
_variant = None   # tuning experiments only: set through use_variant_library(), reported by bench.py as "variant_lib"
SRC = os.path.join(_HERE, "csrc", "hrgym_hip.hip")
SRC_BOX = os.path.join(_HERE, "csrc", "hrgym_box.hip")   # the same sources compiled with the manipulation object (PickPlaceHumanCart)
SRC_HO = os.path.join(_HERE, "csrc", "hrgym_handover.hip")   # ... and once more with the object <-> hand weld of the handover tasks
SRC_LIFT = os.path.join(_HERE, "csrc", "hrgym_lift.hip")     # ... and with the connect equalities / task logic of CollaborativeLiftingCart
SRC_STACK = os.path.join(_HERE, "csrc", "hrgym_stack.hip")   # ... and the four-cube system of CollaborativeStackingCart
SRC_HAMMER = os.path.join(_HERE, "csrc", "hrgym_hammer.hip")  # ... and board + nail + hammer of CollaborativeHammeringCart
SRC_HULLS = os.path.join(_HERE, "csrc", "hrgym_hulls.hip")    # ... and the ReachHuman kernels with the arm links' convex hulls as collision geometry

EXPORTS = [
    "hrg_last_error", "hrg_version", "hrg_state_bytes", "hrg_batch_create", "hrg_batch_destroy", "hrg_batch_reset",
    "hrg_batch_step", "hrg_batch_contacts", "hrg_batch_capsules", "hrg_batch_get_state", "hrg_batch_set_state",
    "hrg_batch_kernel_time", "hrg_batch_enable_taps", "hrg_box_bytes", "hrg_batch_get_box", "hrg_batch_set_box", "hrg_batch_get_states", "hrg_batch_set_states",
    "hrg_batch_check_actions", "hrg_stack_bytes", "hrg_batch_get_stack", "hrg_batch_set_stack", "hrg_batch_launch_order",
    "hrg_hammer_bytes", "hrg_batch_get_hammer", "hrg_batch_set_hammer", "hrg_test_hull_queries",
]


def build_library(force=False, verbose=False):
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    deps = [SRC, SRC_BOX, SRC_HO, SRC_LIFT, SRC_STACK, SRC_HAMMER, SRC_HULLS] + [os.path.join(_HERE, "csrc", f) for f in ("hrgym_device.h", "hrgym_kernels.h", "hrgym_hull.h")] + [
        os.path.join(os.path.dirname(_HERE), "include", f) for f in ("hrgym.h", "hrgym_state.h")]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    # -fapprox-func, device code only: FP64 divisions become v_rcp_f64 + two Newton steps + one residual correction (8 instructions, within an ulp) instead of the
    # IEEE-exact sequence with scaling and fix-up (12+): 140 division sites in the ReachHuman kernel alone, 5 % of its vector instructions.  The host side and the
    # oracle keep IEEE arithmetic; the parity tolerance (1e-5 relative) is eleven orders of magnitude above the difference.
    # -mllvm -disable-machine-licm: the machine-level loop-invariant code motion pulled the materialisation of ~25 FP64 literals (polynomial coefficients of the
    # in-loop sincos / atan / exp code) out of the 25-cycle loop into VGPR pairs that then stayed live across EVERY phase -- a fifth of the 128-register budget, one
    # pair spilled.  Without it the ReachHuman kernel allocates 120 VGPRs with no VGPR spill (was 128 + 5 spilled; SGPR spills 93 -> 64) and every variant is
    # 3 - 6 % faster (profiles/r03_*).
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-Xarch_device", "-fapprox-func", "-mllvm", "-disable-machine-licm", "-o", LIB_PATH,
           SRC, SRC_BOX, SRC_HO, SRC_LIFT, SRC_STACK, SRC_HAMMER, SRC_HULLS]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def use_variant_library(path):
    """Tuning experiments only (tools/): load another build of the library instead of the shipping one.  Must be called before the first
    batch is created; `variant_library()` tells callers (bench.py echoes it in its JSON line) that results are not the shipping build's."""
    global _variant
    if _lib is not None:
        raise RuntimeError("use_variant_library() must be called before the library is loaded")
    _variant = os.path.abspath(path)


def variant_library():
    return _variant


def load_library():
    """dlopen the HIP library and declare signatures. Raises if it is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    for var in ("HRG_LIB_PATH", "HRG_PHASE_MASK"):   # round-1 tuning switches: a stale variable must not change what runs
        if os.environ.get(var):
            raise RuntimeError(f"{var} is set: the library no longer honours it (use _lib.use_variant_library() / a -DHRG_STAMPS diagnostic build); unset it")
    path = _variant or LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                           "There is no CPU fallback for the stepper.")
    lib = ctypes.CDLL(path)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    lib.hrg_last_error.restype = ctypes.c_char_p
    lib.hrg_version.restype = ctypes.c_char_p
    lib.hrg_state_bytes.restype = ctypes.c_size_t
    lib.hrg_batch_create.argtypes = [ctypes.POINTER(ModelDesc), ctypes.POINTER(ClipTable), i32, i64, i32, ctypes.POINTER(vp)]
    lib.hrg_batch_destroy.argtypes = [vp]
    lib.hrg_batch_destroy.restype = None
    lib.hrg_batch_reset.argtypes = [vp, vp, vp, vp]
    lib.hrg_batch_step.argtypes = [vp] * 8
    lib.hrg_batch_contacts.argtypes = [vp, vp, vp]
    lib.hrg_batch_launch_order.argtypes = [vp, vp, vp]
    lib.hrg_batch_capsules.argtypes = [vp, vp, vp, vp]
    lib.hrg_batch_get_state.argtypes = [vp, i32, vp, ctypes.c_size_t]
    lib.hrg_batch_set_state.argtypes = [vp, i32, vp, ctypes.c_size_t]
    lib.hrg_batch_get_states.argtypes = [vp, vp, i32, vp, vp]
    lib.hrg_batch_set_states.argtypes = [vp, vp, i32, vp, vp]
    lib.hrg_box_bytes.restype = ctypes.c_size_t
    lib.hrg_batch_get_box.argtypes = [vp, i32, vp, ctypes.c_size_t]
    lib.hrg_batch_set_box.argtypes = [vp, i32, vp, ctypes.c_size_t]
    lib.hrg_hammer_bytes.restype = ctypes.c_size_t
    lib.hrg_batch_get_hammer.argtypes = [vp, i32, vp, ctypes.c_size_t]
    lib.hrg_batch_set_hammer.argtypes = [vp, i32, vp, ctypes.c_size_t]
    lib.hrg_test_hull_queries.argtypes = [vp, vp, vp, i32, vp]
    lib.hrg_stack_bytes.restype = ctypes.c_size_t
    lib.hrg_batch_get_stack.argtypes = [vp, i32, vp, ctypes.c_size_t]
    lib.hrg_batch_set_stack.argtypes = [vp, i32, vp, ctypes.c_size_t]
    lib.hrg_batch_enable_taps.argtypes = [vp, i32]
    lib.hrg_batch_check_actions.argtypes = [vp, vp, vp, vp]
    lib.hrg_batch_kernel_time.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(i64)]
    if lib.hrg_state_bytes() != ctypes.sizeof(EnvState):
        raise RuntimeError("hrg_env_state layout mismatch between header mirror and library: rebuild")
    _lib = lib
    return lib


class HrgError(RuntimeError):
    pass


def _check(lib, rc):
    if rc != 0:
        raise HrgError(f"hrgym error {rc}: {lib.hrg_last_error().decode()}")


class HipBatch:
    """n_envs ReachHuman environments resident on one MI355X.

    All I/O buffers are torch tensors on the batch's device; `step` is asynchronous (ordered on torch's
    current stream of that device)."""

    def __init__(self, desc, clips, n_envs, env_id0=0, device=0, out=None):
        """`out`: optional (obs, term_obs, reward, info, done) device tensors to write into — row slices of a larger block that several
        batches share (mixed.MixedBatch); by default the batch allocates its own packed block."""
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("HipBatch needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.torch = torch
        self.lib = load_library()
        self.n = int(n_envs)
        self.device = torch.device("cuda", device)
        self._clips = clips  # keep host frame table alive during create
        table = clips.table()
        if desc.n_clips != clips.n_clips:
            desc.n_clips = clips.n_clips
        self.h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.hrg_batch_create(ctypes.byref(desc), ctypes.byref(table), self.n, int(env_id0), device, ctypes.byref(self.h)))
            C = CONST
            # one contiguous SoA output block so that multi-GPU runs need ONE all-gather per step (dist.packed_layout):
            # [obs f32 n*57 | reward f32 n | info i32 n*13 | done u8 n | term_obs f32 n*57]; `packed_head` (everything but the terminal
            # observations) is the part a step publishes to the other ranks
            from .dist import packed_layout
            n, od, idim = self.n, C["HRG_OBS_DIM"], C["HRG_INFO_DIM"]
            if out is not None:
                self.obs, self.term_obs, self.reward, self.info, self.done = out
                want = [((n, od), torch.float32), ((n, od), torch.float32), ((n,), torch.float32), ((n, idim), torch.int32), ((n,), torch.uint8)]
                for t, (shape, dt) in zip(out, want):
                    if tuple(t.shape) != shape or t.dtype != dt or t.device != self.device or not t.is_contiguous():
                        raise ValueError(f"out tensor {tuple(t.shape)} {t.dtype} on {t.device}: expected contiguous {shape} {dt} on {self.device}")
                self.packed = self.packed_head = self.packed_layout = None
                return
            lay = packed_layout(n)
            offs, sizes, tot = lay["offsets"], lay["sizes"], lay["total"]
            self.packed = torch.zeros(tot, dtype=torch.uint8, device=self.device)
            self.packed_head = self.packed[:lay["head"]]
            self.packed_layout = dict(offsets=offs, sizes=sizes)
            self.obs = self.packed[offs[0]:offs[0] + sizes[0]].view(torch.float32).view(n, od)
            self.term_obs = self.packed[offs[1]:offs[1] + sizes[1]].view(torch.float32).view(n, od)
            self.reward = self.packed[offs[2]:offs[2] + sizes[2]].view(torch.float32)
            self.info = self.packed[offs[3]:offs[3] + sizes[3]].view(torch.int32).view(n, idim)
            self.done = self.packed[offs[4]:offs[4] + sizes[4]]

    def _stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def reset(self, mask=None):
        """Reset all envs (mask None) or those with mask != 0 (uint8 tensor on device). Returns obs tensor."""
        mp = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=self.torch.uint8).contiguous()
            mp = ctypes.c_void_p(mask.data_ptr())
        with self.torch.cuda.device(self.device):
            _check(self.lib, self.lib.hrg_batch_reset(self.h, mp, ctypes.c_void_p(self.obs.data_ptr()), self._stream()))
        return self.obs

    def step(self, actions):
        """actions: float64 tensor [n, 7] on device. Returns (obs, reward, done, info) device tensors (views)."""
        t = self.torch
        if actions.dtype != t.float64 or actions.device != self.device or not actions.is_contiguous():
            actions = actions.to(device=self.device, dtype=t.float64).contiguous()
        if tuple(actions.shape) != (self.n, CONST["HRG_ACT_DIM"]):
            raise ValueError(f"actions must be [{self.n}, {CONST['HRG_ACT_DIM']}]")
        vp = ctypes.c_void_p
        with t.cuda.device(self.device):
            _check(self.lib, self.lib.hrg_batch_step(self.h, vp(actions.data_ptr()), vp(self.obs.data_ptr()), vp(self.term_obs.data_ptr()),
                                                   vp(self.reward.data_ptr()), vp(self.done.data_ptr()), vp(self.info.data_ptr()), self._stream()))
        self._keep = actions
        return self.obs, self.reward, self.done, self.info

    def check_actions(self, actions):
        """HumanEnv.check_collision_action for the whole batch: uint8 tensor [n], 1 where the joint-space action's goal configuration collides
        with the static scene or the robot itself (nothing is stepped)."""
        t = self.torch
        if actions.dtype != t.float64 or actions.device != self.device or not actions.is_contiguous():
            actions = actions.to(device=self.device, dtype=t.float64).contiguous()
        if tuple(actions.shape) != (self.n, CONST["HRG_ACT_DIM"]):
            raise ValueError(f"actions must be [{self.n}, {CONST['HRG_ACT_DIM']}]")
        out = t.empty(self.n, dtype=t.uint8, device=self.device)
        with t.cuda.device(self.device):
            _check(self.lib, self.lib.hrg_batch_check_actions(self.h, ctypes.c_void_p(actions.data_ptr()), ctypes.c_void_p(out.data_ptr()), self._stream()))
        self._keep_chk = actions
        return out

    def get_state(self, e):
        s = EnvState()
        _check(self.lib, self.lib.hrg_batch_get_state(self.h, int(e), ctypes.byref(s), ctypes.sizeof(s)))
        return s

    def set_state(self, e, s):
        _check(self.lib, self.lib.hrg_batch_set_state(self.h, int(e), ctypes.byref(s), ctypes.sizeof(s)))

    def get_states(self, envs):
        """Environment states (+ manipulation objects) of several envs in one call: (EnvState[n], BoxState[n])."""
        import numpy as np
        idx = np.ascontiguousarray(envs, np.int32)
        st, bx = (EnvState * len(idx))(), (BoxState * len(idx))()
        _check(self.lib, self.lib.hrg_batch_get_states(self.h, idx.ctypes.data_as(ctypes.c_void_p), len(idx), ctypes.byref(st), ctypes.byref(bx)))
        return st, bx

    def set_states(self, envs, states, boxes=None):
        """Reference-state initialisation: overwrite the listed envs' states (wrappers/dataset_wrapper.py:88-160)."""
        import numpy as np
        idx = np.ascontiguousarray(envs, np.int32)
        assert len(states) == len(idx) and (boxes is None or len(boxes) == len(idx))
        _check(self.lib, self.lib.hrg_batch_set_states(self.h, idx.ctypes.data_as(ctypes.c_void_p), len(idx), ctypes.byref(states),
                                                       ctypes.byref(boxes) if boxes is not None else None))

    def stagger_episode_phases(self, horizon):
        """Spread the TimeLimit phase over the batch: env e continues as if it were (e * horizon) // n policy steps into its episode.  A freshly reset batch
        has every env at step 0, so that all of them time out in the same step, every `horizon` steps; a batch that has been training for a while has its
        episode ends spread evenly (early successes and failures shift each env's phase).  Benchmarks call this once after reset."""
        import numpy as np
        idx = np.arange(self.n, dtype=np.int32)
        st, bx = self.get_states(idx)
        for e in range(self.n):
            st[e].timestep = (e * int(horizon)) // self.n
        self.set_states(idx, st, bx)

    def get_box(self, e):
        """The manipulation object of env e (PickPlaceHumanCart; zeros for ReachHuman)."""
        s = BoxState()
        _check(self.lib, self.lib.hrg_batch_get_box(self.h, int(e), ctypes.byref(s), ctypes.sizeof(s)))
        return s

    def set_box(self, e, s):
        _check(self.lib, self.lib.hrg_batch_set_box(self.h, int(e), ctypes.byref(s), ctypes.sizeof(s)))

    def get_hammer(self, e):
        """Board, hammer, nail + task bookkeeping of env e (CollaborativeHammeringCart)."""
        s = HammerState()
        _check(self.lib, self.lib.hrg_batch_get_hammer(self.h, int(e), ctypes.byref(s), ctypes.sizeof(s)))
        return s

    def set_hammer(self, e, s):
        _check(self.lib, self.lib.hrg_batch_set_hammer(self.h, int(e), ctypes.byref(s), ctypes.sizeof(s)))

    def get_stack(self, e):
        """The four cubes + task bookkeeping of env e (CollaborativeStackingCart)."""
        s = StackState()
        _check(self.lib, self.lib.hrg_batch_get_stack(self.h, int(e), ctypes.byref(s), ctypes.sizeof(s)))
        return s

    def set_stack(self, e, s):
        _check(self.lib, self.lib.hrg_batch_set_stack(self.h, int(e), ctypes.byref(s), ctypes.sizeof(s)))

    def contacts(self):
        import numpy as np
        pairs = np.zeros((self.n, CONST["HRG_NCON_MAX"], 2), np.int32)
        ncon = np.zeros(self.n, np.int32)
        _check(self.lib, self.lib.hrg_batch_contacts(self.h, pairs.ctypes.data_as(ctypes.c_void_p), ncon.ctypes.data_as(ctypes.c_void_p)))
        return pairs, ncon

    def launch_order(self):
        """(order, n_busy): the env each workgroup of the next step launch will step, and how many of them -- from the front -- were busy in the last step."""
        import numpy as np
        order = np.zeros(self.n, np.int32)
        nb = ctypes.c_int32(0)
        _check(self.lib, self.lib.hrg_batch_launch_order(self.h, order.ctypes.data_as(ctypes.c_void_p), ctypes.byref(nb)))
        return order, int(nb.value)

    def enable_taps(self, on=True):
        _check(self.lib, self.lib.hrg_batch_enable_taps(self.h, int(bool(on))))

    def capsules(self):
        import numpy as np
        r = np.zeros((self.n, CONST["HRG_NSHIELD_RCAP"], 7))
        h = np.zeros((self.n, CONST["HRG_NHCAP_MAX"], 7))
        nh = np.zeros(self.n, np.int32)
        vp = ctypes.c_void_p
        _check(self.lib, self.lib.hrg_batch_capsules(self.h, r.ctypes.data_as(vp), h.ctypes.data_as(vp), nh.ctypes.data_as(vp)))
        return r, h, nh

    def kernel_time(self):
        """(avg kernel ms, launches) of the step kernels since the previous call (HIP events on the launch stream).
        The first call arms the timer."""
        ms, n = ctypes.c_double(), ctypes.c_int64()
        _check(self.lib, self.lib.hrg_batch_kernel_time(self.h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.lib.hrg_batch_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
