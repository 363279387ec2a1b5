"""Tuning aid: alternate two launch orders of the mixed batch (A/B/A/B...) so that the drift of the simulation state cancels."""
import sys, time
sys.path.insert(0, '.')
import torch
from human_robot_gym_amd import mixed
n = 4096
M = mixed.make_mixed_batch(n, seed=1234)
M.reset()
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
acts = [torch.rand((n, 7), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1 for _ in range(8)]
for k in range(1000): M.step(acts[k % 8])
torch.cuda.synchronize()
idx = {e: i for i, e in enumerate(M.env_ids)}
S, R2H, H2R, L, P, R = (idx[k] for k in ("CollaborativeStackingCart", "RobotHumanHandoverCart", "HumanRobotHandoverCart", "CollaborativeLiftingCart", "PickPlaceHumanCart", "ReachHuman"))
orders = {"longest first": [S, R2H, H2R, L, P, R], "stack, small, handovers": [S, L, P, R, R2H, H2R], "lift r2h reach h2r stack pp": [L, R2H, R, H2R, S, P],
          "small, stack, handovers": [R, P, L, S, R2H, H2R], "stack, lift, r2h, pp, h2r, reach": [S, L, R2H, P, H2R, R]}
tot = {k: 0.0 for k in orders}
for rep in range(6):
    for name, order in orders.items():
        M._launch_order = list(order)
        for k in range(5): M.step(acts[k % 8])
        torch.cuda.synchronize(); t = time.perf_counter()
        for k in range(40): M.step(acts[k % 8])
        torch.cuda.synchronize(); tot[name] += 1e3 * (time.perf_counter() - t) / 40
for name in orders:
    print("%-34s %.3f ms" % (name, tot[name] / 6), flush=True)
