"""PCIe-inclusive rate of the numpy HipVecEnv path (H2D actions, D2H packed outputs, 4096 info dicts per step)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from human_robot_gym_amd.vec_env import HipVecEnv
kw = dict(shield_type="SSM", control_freq=10, horizon=100, done_at_success=True, reward_shaping=True, seed=1234)
for dicts in (False, True):
    env = HipVecEnv(4096, env_kwargs=kw, info_dicts=dicts)
    env.reset()
    rng = np.random.RandomState(0)
    acts = [rng.uniform(-1, 1, (4096, 7)) for _ in range(8)]
    for k in range(20): env.step(acts[k % 8])
    t = time.perf_counter()
    for k in range(60): env.step(acts[k % 8])
    el = time.perf_counter() - t
    print("HipVecEnv numpy path, info dicts %s: %.0f env steps/s (%.2f ms per 4096-env step)" % (dicts, 4096 * 60 / el, 1e3 * el / 60))
    env.close()
