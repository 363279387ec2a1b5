"""Tuning aid: kernel time of the step for OFF / SSM with the human near (default) and far (always safe)."""
import sys, torch
sys.path.insert(0, '.')
import human_robot_gym_amd as hrg
from human_robot_gym_amd._lib import HipBatch
clips = hrg.synthetic_clips(13, seed=0)
for name, kw in [("OFF", dict(shield_type="OFF")), ("SSM near", dict(shield_type="SSM")), ("SSM far", dict(shield_type="SSM", base_human_pos_offset=[6.0, 0, 0]))]:
    kw.update(control_freq=10, horizon=100, done_at_success=True, reward_shaping=True, seed=1234)
    G = HipBatch(hrg.build_model_desc(kw, n_clips=13), clips, 4096)
    G.reset()
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    acts = [torch.rand((4096, 7), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1 for _ in range(8)]
    for k in range(5): G.step(acts[k % 8])
    G.kernel_time()
    fs = 0
    for k in range(30):
        o, r, d, i = G.step(acts[k % 8])
    torch.cuda.synchronize()
    ms, n = G.kernel_time()
    print(name, "kernel ms %.3f" % ms, "failsafe interventions/env (cumulative mean)", float(i[:, 8].float().mean()), "n_collisions mean", float(i[:, 2].float().mean()), "done rate", float(d.float().mean()))
    G.close()
