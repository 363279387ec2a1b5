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
env_id = sys.argv[2] if len(sys.argv) > 2 else "ReachHuman"
from human_robot_gym_amd.mixed import task_clips
clips = task_clips(env_id, 13)
kw = dict(shield_type=sys.argv[3] if len(sys.argv) > 3 else "SSM", control_freq=10, seed=1234)
if env_id == "ReachHuman":
    kw.update(horizon=100, done_at_success=True, reward_shaping=True)
G = HipBatch(hrg.build_model_desc(kw, n_clips=13, env_id=env_id), clips, n); G.reset()
G.stagger_episode_phases(100 if env_id == "ReachHuman" else 1000)
sfx = {"ReachHuman": "", "CollaborativeStackingCart": "_stack", "HumanRobotHandoverCart": "_ho", "RobotHumanHandoverCart": "_ho", "CollaborativeLiftingCart": "_lift"}.get(env_id, "_box")
envcyc, envacc = getattr(lib, "hrg_debug_envcyc" + sfx), getattr(lib, "hrg_debug_envacc" + sfx)
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
acts = [torch.rand((n, 7), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1 for _ in range(16)]
for k in range(int(os.environ.get('PREROLL', 150))): G.step(acts[k % 16])
torch.cuda.synchronize()
names = {0: "prolog", 1: "sh-tail", 2: "rdyn", 3: "ctrl", 4: "human", 5: "collide", 6: "classify", 7: "dynstep", 8: "epi", 9: "reset", 10: "s-plan", 11: "s-paths", 12: "s-qe", 13: "s-fk", 14: "s-verify", 15: "s-upd",
         20: "d-M", 21: "d-rows", 22: "d-warm", 23: "d-gradH", 24: "d-chol", 25: "d-solve", 26: "d-ls", 27: "d-grad/ho-tail"}
buf = np.zeros((n, 3), np.uint64); acc = np.zeros((n, 32), np.uint64)
for rep in range(6):
    order, _nb = G.launch_order()          # workgroup slot -> env of the launch about to run (the stamps are per slot)
    obs, r, d, info = G.step(acts[rep]); torch.cuda.synchronize()
    envcyc(buf.ctypes.data_as(ctypes.c_void_p), n); envacc(acc.ctypes.data_as(ctypes.c_void_p), n)
    dur = (buf[:, 1] - buf[:, 0]).astype(np.float64); a = acc.astype(np.float64)
    med = np.median(a, axis=0)
    print("launch %d: dur p50 %.0f p99 %.0f max %.0f" % (rep, np.median(dur), np.percentile(dur, 99), dur.max()))
    print("  median env : " + " ".join("%s %.0fk" % (names[k], med[k] / 1e3) for k in names) + " | it %.0f ls %.0f fact %.0f rows %.0f" % tuple(med[16:20]))
    slots = order_by_dur = np.argsort(-dur)
    for sl in slots[:6]:
        e = int(order[sl])
        st, _ = G.get_states(np.array([e]))
        print("  env %5d dur %.2fM: " % (e, dur[sl] / 1e6) + " ".join("%s %.0fk" % (names[k], a[sl, k] / 1e3) for k in names) + " | it %.0f ls %.0f fact %.0f rows %.0f | ncon %d safe %d done %d" % (tuple(a[sl, 16:20]) + (st[0].ncon, st[0].is_safe, int(d[e]))))
