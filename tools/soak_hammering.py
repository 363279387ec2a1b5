"""Long run of CollaborativeHammeringCart on the HIP stepper (random actions, full-length episodes): finite values, crash rate, phase histogram -- and who drives
the nail.  Round 2's model (soft friction row only) let gravity push the nail in: 19 553 tasks "completed" in this run without the robot's doing.  With MuJoCo's
noslip post-pass restated (collaborative_hammering_cartesian_env.py:1161) the nail only yields to a force above its friction loss: every `hammered in` event is
counted together with whether a hammer geom touched the nail head at any substep end since the nail was last reset.
    python tools/soak_hammering.py [steps] [random|still] [nail_frictionloss]
`still`: zero actions -- the robot holds its posture with the hammer in the gripper while the human presents the board: nobody works on the nail, so it has to stay out."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import human_robot_gym_amd as hrg
from human_robot_gym_amd import mixed
from human_robot_gym_amd._cstruct import CONST as C
from human_robot_gym_amd._lib import HipBatch
n = 4096
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
mode = sys.argv[2] if len(sys.argv) > 2 else "random"
extra = dict(nail_frictionloss=float(sys.argv[3])) if len(sys.argv) > 3 else {}
env_id = "CollaborativeHammeringCart"
clips = mixed.task_clips(env_id, 13)
d = hrg.build_model_desc(dict(shield_type="SSM", horizon=1000, seed=5, **mixed.task_env_kwargs(env_id), **extra), n_clips=clips.n_clips, env_id=env_id)
G = HipBatch(d, clips, n); G.reset(); G.stagger_episode_phases(1000)
g = torch.Generator(device="cpu").manual_seed(4)
crashes = dones = wins = bad = 0
GEOM_BOX = C["HRG_NRCAP"] + C["HRG_NHB"] + 2
g_nail, g_hammer = GEOM_BOX + C["HRG_HG_NAIL"], (GEOM_BOX + C["HRG_HG_HANDLE"], GEOM_BOX + C["HRG_HG_HEAD"])
struck = np.zeros(n, bool)          # a hammer geom has touched the nail head since the nail was last pulled out
was_in = np.zeros(n, bool)
in_events = in_unstruck = 0
touch_hammer = touch_other_only = touch_none = touch_unknown = 0   # hrg_hammer_state.nail_touch of a newly driven-in nail: who touched the nail head at ANY substep end
goals_prev = np.zeros(n, np.int64)
completed = completed_unstruck = 0
max_progress = 0.0
ncon_hist = np.zeros(C["HRG_NCON_MAX"] + 1, np.int64)   # contacts at the substep that ends a policy step (the solve takes the first HRG_NCON_DYN_HAMMER)
for k in range(steps):
    a = (torch.rand((n, 7), generator=g, dtype=torch.float64) * 2 - 1).cuda()
    if mode == "still":
        a.zero_()
    obs, r, dn, info = G.step(a)
    crashes += int(info[:, 11].sum().item()); dones += int(dn.sum().item()); wins += int((r > 0).sum().item())
    pairs, ncon = G.contacts()
    ncon_hist += np.bincount(np.minimum(ncon, C["HRG_NCON_MAX"]), minlength=C["HRG_NCON_MAX"] + 1)
    hit = np.zeros(n, bool)
    for gh in g_hammer:
        hit |= np.any(((pairs[:, :, 0] == gh) & (pairs[:, :, 1] == g_nail)) | ((pairs[:, :, 1] == gh) & (pairs[:, :, 0] == g_nail)), axis=1)
    struck |= hit
    prog = G.term_obs[:, 61].cpu().numpy()                    # nail_hammering_progress of the step's own observation (before an auto-reset)
    max_progress = max(max_progress, float(prog.max()))
    now_in = 1.0 - prog < d.hm_goal_tolerance
    new_in = now_in & ~was_in
    in_events += int(new_in.sum()); in_unstruck += int((new_in & ~struck).sum())
    for e in np.nonzero(new_in)[0]:
        if bool(dn[e].item()):
            touch_unknown += 1          # the episode ended in this very step: the reset has pulled the nail out and cleared the record
            continue
        t = G.get_hammer(int(e)).nail_touch
        touch_hammer += int(t & 1 != 0); touch_other_only += int(t == 2); touch_none += int(t == 0)
    was_in = now_in
    goals = info[:, 9].cpu().numpy().astype(np.int64)
    done_np = dn.cpu().numpy().astype(bool)
    fresh = (goals > goals_prev) & ~done_np
    completed += int(fresh.sum()); completed_unstruck += int((fresh & ~struck).sum())
    goals_prev = np.where(done_np, 0, goals)
    renew = done_np | fresh                                   # the nail is pulled out again at a reset and at _on_goal_reached
    struck[renew] = False; was_in[renew] = False
    if k % 250 == 249 or k == steps - 1:
        o = obs.cpu().numpy(); bad += int((~np.isfinite(o)).sum())
        hms = [G.get_hammer(e) for e in range(0, n, 16)]
        ph = np.bincount([h.task_phase for h in hms], minlength=5)
        print(f"step {k}: phases {ph.tolist()} progress mean {np.mean(o[:, 61]):.4f} max so far {max_progress:.4f} gripped {np.mean(o[:, 39]):.3f} board z [{o[:, 35].min():.2f}, {o[:, 35].max():.2f}] "
              f"hammer z min {o[:, 49].min():.2f} crashes {crashes} dones {dones} successes {wins} non-finite {bad} | nails hammered in {in_events} (without a hammer contact {in_unstruck}) "
              f"tasks completed {completed} (without a hammer contact {completed_unstruck})", flush=True)
order, nb = G.launch_order()
print("launch order a permutation", bool(np.array_equal(np.sort(order), np.arange(n))))
print(f"RESULT mode {mode} steps {steps} envs {n} nail_frictionloss {d.hm_nail_frictionloss}: nails hammered in {in_events}, of them without a hammer contact {in_unstruck}; tasks completed {completed}, "
      f"of them without a hammer contact {completed_unstruck}; largest nail progress {max_progress:.4f}; crashes {crashes}")
print(f'nail_touch of the {in_events} driven-in nails (every substep end since the nail was pulled out): a hammer geom touched it {touch_hammer}, only other geoms {touch_other_only}, nothing {touch_none}, episode ended in the same step {touch_unknown}')
print('contacts per env at the end of a policy step (histogram 0 ..):', ncon_hist.tolist(), ' share above', C['HRG_NCON_DYN_HAMMER'], ':', float(ncon_hist[C['HRG_NCON_DYN_HAMMER'] + 1:].sum()) / ncon_hist.sum())
