"""Long run of CollaborativeHammeringCart on the HIP stepper (full-length episodes: the nail is driven in, RETREAT, COMPLETE, next animation): finite values, crash rate, phase histogram."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import human_robot_gym_amd as hrg
from human_robot_gym_amd import mixed
from human_robot_gym_amd._lib import HipBatch
n, steps = 4096, 1500
env_id = "CollaborativeHammeringCart"
clips = mixed.task_clips(env_id, 13)
d = hrg.build_model_desc(dict(shield_type="SSM", horizon=1000, seed=5, **mixed.task_env_kwargs(env_id)), n_clips=clips.n_clips, env_id=env_id)
G = HipBatch(d, clips, n); G.reset(); G.stagger_episode_phases(1000)
g = torch.Generator(device="cpu").manual_seed(4)
crashes = dones = wins = bad = 0
ngoal = 0
for k in range(steps):
    a = (torch.rand((n, 7), generator=g, dtype=torch.float64) * 2 - 1).cuda()
    obs, r, dn, info = G.step(a)
    crashes += int(info[:, 11].sum().item()); dones += int(dn.sum().item()); wins += int((r > 0).sum().item())
    if k % 250 == 249:
        o = obs.cpu().numpy(); bad += int((~np.isfinite(o)).sum())
        hms = [G.get_hammer(e) for e in range(0, n, 16)]
        ph = np.bincount([h.task_phase for h in hms], minlength=5)
        print(f"step {k}: phases {ph.tolist()} progress mean {np.mean(o[:, 61]):.3f} gripped {np.mean(o[:, 39]):.3f} board z [{o[:, 35].min():.2f}, {o[:, 35].max():.2f}] hammer z min {o[:, 49].min():.2f} "
              f"crashes {crashes} dones {dones} successes {wins} non-finite {bad}", flush=True)
order, nb = G.launch_order()
print("launch order a permutation", bool(np.array_equal(np.sort(order), np.arange(n))), "n_goal_reached max", int(info[:, 9].max().item()))
