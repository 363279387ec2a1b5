"""Statistics of long random-action runs of the collaboration tasks on the HIP stepper (finite values, crash rate, phase histogram)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import human_robot_gym_amd as hrg  # noqa: E402
from human_robot_gym_amd import mixed  # noqa: E402
from human_robot_gym_amd._lib import HipBatch  # noqa: E402

n, steps = 4096, 600
for env_id, shield in (("ReachHuman", "SSM"), ("PickPlaceHumanCart", "SSM"), ("HumanObjectInspectionCart", "SSM"), ("HumanRobotHandoverCart", "PFL"), ("RobotHumanHandoverCart", "PFL"),
                       ("CollaborativeLiftingCart", "SSM"), ("CollaborativeStackingCart", "SSM"), ("CollaborativeHammeringCart", "SSM")):
    clips = mixed.task_clips(env_id, 5, min_frames=300, max_frames=600) if env_id != "ReachHuman" else hrg.synthetic_clips(5, seed=0)
    d = hrg.build_model_desc(dict(shield_type=shield, horizon=150, seed=31, **mixed.task_env_kwargs(env_id)), n_clips=clips.n_clips, env_id=env_id)
    G = HipBatch(d, clips, n)
    G.reset()
    g = torch.Generator(device="cpu").manual_seed(4)
    crashes = dones = wins = 0
    bad = 0
    for k in range(steps):
        a = (torch.rand((n, 7), generator=g, dtype=torch.float64) * 2 - 1).cuda()
        obs, r, dn, info = G.step(a)
        crashes += int(info[:, 11].sum().item()); dones += int(dn.sum().item()); wins += int((r > 0).sum().item())
        if k % 100 == 99:
            o = obs.cpu().numpy()
            bad += int((~np.isfinite(o)).sum())
            z = o[:, 49]
            print(f"  {env_id} step {k}: object z min {z.min():.3f} max {z.max():.3f}; below table-5mm {(z < 0.8 + 0.02 - 5e-3).mean():.3f}; gripped {o[:, 39].mean():.3f}", flush=True)
    order, nb = G.launch_order()
    perm_ok = bool(np.array_equal(np.sort(order), np.arange(n)))
    if env_id == "CollaborativeHammeringCart":
        hms = [G.get_hammer(e) for e in range(0, n, 8)]
        ph = np.bincount([h.task_phase for h in hms], minlength=6)
        qn = max(abs(np.linalg.norm(list(h.quat[b])) - 1) for h in hms for b in range(2))
        print(f"{env_id}: crashes {crashes} dones {dones} positive rewards {wins} non-finite {bad} phases {ph.tolist()} gripped {np.mean([h.gripped for h in hms]):.3f} "
              f"nail progress mean {np.mean([min(max(h.nail_q / 0.06, 0), 1) for h in hms]):.3f} |quat|-1 {qn:.2e} launch order a permutation {perm_ok} busy {nb}", flush=True)
    elif env_id == "CollaborativeStackingCart":
        sks = [G.get_stack(e) for e in range(0, n, 8)]
        ph = np.bincount([s_.task_phase for s_ in sks], minlength=6)
        qn = max(abs(np.linalg.norm(list(s_.quat[c])) - 1) for s_ in sks for c in range(4))
        print(f"{env_id}: crashes {crashes} dones {dones} positive rewards {wins} non-finite {bad} phases {ph.tolist()} max stack height {max(s_.max_stack_height for s_ in sks)} "
              f"|quat|-1 {qn:.2e} launch order a permutation {perm_ok} busy {nb}", flush=True)
    else:
        st, bx = G.get_states(np.arange(0, n, 8))
        ph = np.bincount([b.task_phase for b in bx], minlength=6)
        qn = max(abs(np.linalg.norm(list(b.quat)) - 1) for b in bx) if env_id != "ReachHuman" else 0.0
        print(f"{env_id}: crashes {crashes} dones {dones} positive rewards {wins} non-finite {bad} phases {ph.tolist()} weld_active {np.mean([b.weld_active for b in bx]):.3f} "
              f"handed over {sum(b.n_handed_over for b in bx)} |quat|-1 {qn:.2e} launch order a permutation {perm_ok} busy {nb}", flush=True)
    G.close()
