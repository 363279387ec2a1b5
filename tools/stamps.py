"""Diagnostic: per-phase share of wave cycles from the -DHRG_STAMPS build (human-robot-gym_amd/variant_stamps.so)."""
import os, sys, ctypes, numpy as np
sys.path.insert(0, '.')
import torch
import human_robot_gym_amd as hrg
from human_robot_gym_amd import _lib
from human_robot_gym_amd._lib import HipBatch, load_library
_lib.use_variant_library("human-robot-gym_amd/variant_stamps.so")   # the -DHRG_STAMPS diagnostic build
lib = load_library()
env_id = sys.argv[2] if len(sys.argv) > 2 else "ReachHuman"
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
SLOW = int(float(sys.argv[4])) if len(sys.argv) > 4 else 0     # > 0: also profile the waves that live longer than this many cycles (ReachHuman kernel only)
GEOM = os.environ.get("HRG_GEOMETRY", "capsule")                # "hull": the hull variant of the ReachHuman kernel
from human_robot_gym_amd.mixed import task_clips
clips = task_clips(env_id, 13)
from human_robot_gym_amd.mixed import task_env_kwargs
kw = dict(shield_type=sys.argv[1] if len(sys.argv) > 1 else "SSM", control_freq=10, seed=1234, **task_env_kwargs(env_id))
if env_id == "ReachHuman":
    kw.update(horizon=100, done_at_success=True, reward_shaping=True)
G = HipBatch(hrg.build_model_desc(kw, n_clips=13, env_id=env_id, robot_geometry=GEOM), clips, N); G.reset()
if os.environ.get("HRG_STAGGER", "1") != "0": G.stagger_episode_phases(100)
stamps = (lib.hrg_debug_stamps_hull if GEOM == "hull" else lib.hrg_debug_stamps) if env_id == "ReachHuman" else (lib.hrg_debug_stamps_hammer if "Hammering" in env_id else lib.hrg_debug_stamps_stack if "Stacking" in env_id else (lib.hrg_debug_stamps_ho if "Handover" in env_id else lib.hrg_debug_stamps_box))
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
acts = [torch.rand((N, 7), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1 for _ in range(16)]
out = np.zeros(32)
for k in range(150): G.step(acts[k % 16])
torch.cuda.synchronize(); stamps(out.ctypes.data_as(ctypes.c_void_p), 1)
slow = np.zeros(34)
if SLOW: lib.hrg_debug_stamps_slow(slow.ctypes.data_as(ctypes.c_void_p), ctypes.c_ulonglong(SLOW), 1)
for k in range(50): G.step(acts[k % 16])
torch.cuda.synchronize(); stamps(out.ctypes.data_as(ctypes.c_void_p), 1)
names = {0: "cycle prologue (set_goal)", 1: "shield: tail (after des)", 2: "robot_dynamics_terms", 3: "controller", 4: "human_control", 5: "collide", 6: "classify", 7: "dynamics_step", 8: "epilogue", 9: "reset/obs",
         10: " shield: cur+plan", 11: " shield: paths", 12: " shield: qe eval", 13: " shield: chain fk", 14: " shield: reach+verify", 15: " shield: update+des"}
names.update({20: " dyn: M chol + a0", 21: " dyn: row setup", 22: " dyn: warm start", 23: " dyn: grad + Hessian", 24: " dyn: chol H", 25: " dyn: solve + p", 26: " dyn: line search", 27: " handover: first-pass tail / stacking: gradient", 28: " dyn: end of the Newton loop", 29: " dyn: noslip sweeps + write-back (hammering)", 31: " dyn: noslip forces + Gram matrix (hammering)"})
tot = out[:16].sum() + out[20:30].sum() + out[31]
for k in list(range(16)) + list(range(20, 30)) + [31]:
    print("%-26s %6.2f %%  (%.0f cycles/env-step)" % (names.get(k, k), 100 * out[k] / tot, out[k] / (50 * N)))
sub = 50 * N * 25
print("per substep: Newton iterations %.2f, line-search evaluations %.2f, Hessian factorizations %.2f, active rows %.2f, noslip sweeps %.2f" % (out[16] / sub, out[17] / sub, out[18] / sub, out[19] / sub, out[30] / sub))

if SLOW:
    lib.hrg_debug_stamps_slow(slow.ctypes.data_as(ctypes.c_void_p), ctypes.c_ulonglong(SLOW), 1)
    ns = max(slow[32], 1)
    print("slow waves (> %d cycles): %d of %d, mean lifetime %.0f" % (SLOW, slow[32], 50 * N, slow[33] / ns))
    tots = slow[:16].sum() + slow[20:28].sum()
    for k in list(range(16)) + list(range(20, 28)):
        print("  %-26s slow %9.0f cycles/env-step (%5.1f %%)   all %9.0f" % (names.get(k, k), slow[k] / ns, 100 * slow[k] / tots, out[k] / (50 * N)))
    print("  slow waves per substep: Newton iterations %.2f, line-search evaluations %.2f, factorizations %.2f, active rows %.2f" % tuple(slow[k] / (ns * 25) for k in (16, 17, 18, 19)))
