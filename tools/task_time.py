"""Kernel time of one task's step at a batch size: python tools/task_time.py ENV_ID N_ENVS [SHIELD] (HRG_TT_LIB=path times a variant build through _lib.use_variant_library)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import human_robot_gym_amd as hrg  # noqa: E402
from human_robot_gym_amd import mixed  # noqa: E402
from human_robot_gym_amd import _lib  # noqa: E402
from human_robot_gym_amd._lib import HipBatch  # noqa: E402

if os.environ.get("HRG_TT_LIB"):
    _lib.use_variant_library(os.environ["HRG_TT_LIB"])

env_id = sys.argv[1] if len(sys.argv) > 1 else "HumanRobotHandoverCart"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
shield = sys.argv[3] if len(sys.argv) > 3 else ("PFL" if "Handover" in env_id else "SSM")
clips = mixed.task_clips(env_id, 13)
d = hrg.build_model_desc(dict(shield_type=shield, horizon=1000, seed=9), n_clips=clips.n_clips, env_id=env_id)
G = HipBatch(d, clips, n)
G.reset()
torch.cuda.synchronize()
print("reset done", flush=True)
g = torch.Generator(device="cpu").manual_seed(0)
acts = [(torch.rand((n, 7), generator=g, dtype=torch.float64) * 2 - 1).cuda() for _ in range(8)]
for k in range(10):
    G.step(acts[k % 8])
    if os.environ.get("HRG_TT_SYNC"):
        torch.cuda.synchronize()
        print("warm-up step", k, flush=True)
torch.cuda.synchronize()
G.kernel_time()
t0 = time.time()
for k in range(40):
    G.step(acts[k % 8])
torch.cuda.synchronize()
dt = time.time() - t0
ms, nl = G.kernel_time()
print(f"{env_id} {shield} n={n} lib={os.path.basename(_lib.variant_library() or 'default')}: {n * 40 / dt / 1e6:.3f} M env steps/s, kernel {ms:.3f} ms x{nl}", flush=True)
G.close()
