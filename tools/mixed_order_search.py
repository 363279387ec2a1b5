"""Tuning aid: step time of the six-task mixed batch for different launch orders of its per-task kernels (each on its own stream)."""
import itertools, sys, time
sys.path.insert(0, '.')
import torch
from human_robot_gym_amd import mixed
n = 4096
M = mixed.make_mixed_batch(n, seed=1234)
M.reset()
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
acts = [torch.rand((n, 7), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1 for _ in range(8)]
for k in range(300): M.step(acts[k % 8])
torch.cuda.synchronize()
names = [e.replace("Collaborative", "").replace("Cart", "")[:10] for e in M.env_ids]
base = list(M._launch_order)
cands = {"longest first (shipping)": base, "shortest first": base[::-1]}
idx = {e: i for i, e in enumerate(M.env_ids)}
S, R2H, H2R, L, P, R = (idx[k] for k in ("CollaborativeStackingCart", "RobotHumanHandoverCart", "HumanRobotHandoverCart", "CollaborativeLiftingCart", "PickPlaceHumanCart", "ReachHuman"))
cands.update({"stack, small three, handovers": [S, L, P, R, R2H, H2R], "handovers, small three, stack": [R2H, H2R, L, P, R, S], "stack, r2h, small three, h2r": [S, R2H, L, P, R, H2R],
              "small three, stack, handovers": [R, P, L, S, R2H, H2R], "stack, reach, r2h, pp, h2r, lift": [S, R, R2H, P, H2R, L], "r2h, stack, lift, h2r, pp, reach": [R2H, S, L, H2R, P, R]})
for name, order in cands.items():
    M._launch_order = list(order)
    for k in range(10): M.step(acts[k % 8])
    torch.cuda.synchronize(); t = time.perf_counter()
    for k in range(60): M.step(acts[k % 8])
    torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t) / 60
    print("%-36s %s  %.3f ms" % (name, [names[i] for i in order], ms), flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "all":
    res = []
    for order in itertools.permutations(range(len(M.env_ids))):
        M._launch_order = list(order)
        for k in range(4): M.step(acts[k % 8])
        torch.cuda.synchronize(); t = time.perf_counter()
        for k in range(24): M.step(acts[k % 8])
        torch.cuda.synchronize(); res.append((1e3 * (time.perf_counter() - t) / 24, order))
    res.sort()
    for ms, order in res[:12] + res[-3:]:
        print("%.3f ms  %s" % (ms, [names[i] for i in order]), flush=True)
    # re-time the best five with more steps
    for ms, order in res[:5]:
        M._launch_order = list(order)
        for k in range(10): M.step(acts[k % 8])
        torch.cuda.synchronize(); t = time.perf_counter()
        for k in range(100): M.step(acts[k % 8])
        torch.cuda.synchronize()
        print("retimed %.3f ms  %s" % (1e3 * (time.perf_counter() - t) / 100, [names[i] for i in order]), flush=True)
