"""Diagnostic (-DHRG_STAMPS build): the slowest envs of a step launch, with their phase cycles and solver counters."""
import os, sys, ctypes, numpy as np
sys.path.insert(0, '.')
import torch
import human_robot_gym_amd as hrg
from human_robot_gym_amd import _lib
from human_robot_gym_amd._lib import HipBatch, load_library
_lib.use_variant_library("human-robot-gym_amd/variant_stamps.so")
lib = load_library()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
clips = hrg.synthetic_clips(13, seed=0)
kw = dict(shield_type="SSM", control_freq=10, horizon=100, done_at_success=True, reward_shaping=True, seed=1234)
G = HipBatch(hrg.build_model_desc(kw, n_clips=13), clips, n); G.reset()
G.stagger_episode_phases(100)
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
acts = [torch.rand((n, 7), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1 for _ in range(16)]
for k in range(150): G.step(acts[k % 16])
torch.cuda.synchronize()
names = {0: "prolog", 1: "sh-tail", 2: "rdyn", 3: "ctrl", 4: "human", 5: "collide", 6: "classify", 7: "dynstep", 8: "epi", 9: "reset", 10: "s-plan", 11: "s-paths", 12: "s-qe", 13: "s-fk", 14: "s-verify", 15: "s-upd",
         20: "d-M", 21: "d-rows", 22: "d-warm", 25: "d-solve", 26: "d-ls"}
buf = np.zeros((n, 3), np.uint64); acc = np.zeros((n, 32), np.uint64)
for rep in range(6):
    obs, r, d, info = G.step(acts[rep]); torch.cuda.synchronize()
    lib.hrg_debug_envcyc(buf.ctypes.data_as(ctypes.c_void_p), n); lib.hrg_debug_envacc(acc.ctypes.data_as(ctypes.c_void_p), n)
    dur = (buf[:, 1] - buf[:, 0]).astype(np.float64); a = acc.astype(np.float64)
    order = np.argsort(-dur)
    med = np.median(a, axis=0)
    print("launch %d: dur p50 %.0f p99 %.0f max %.0f" % (rep, np.median(dur), np.percentile(dur, 99), dur.max()))
    print("  median env : " + " ".join("%s %.0fk" % (names[k], med[k] / 1e3) for k in names) + " | it %.0f ls %.0f fact %.0f rows %.0f" % tuple(med[16:20]))
    for e in order[:6]:
        st, _ = G.get_states(np.array([e]))
        print("  env %5d dur %.2fM: " % (e, dur[e] / 1e6) + " ".join("%s %.0fk" % (names[k], a[e, k] / 1e3) for k in names) + " | it %.0f ls %.0f fact %.0f rows %.0f | ncon %d safe %d done %d" % (tuple(a[e, 16:20]) + (st[0].ncon, st[0].is_safe, int(d[e]))))
