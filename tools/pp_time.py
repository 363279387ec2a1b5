import sys, time, torch
sys.path.insert(0, '/root/repo')
import human_robot_gym_amd as hrg
from human_robot_gym_amd._lib import HipBatch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
clips = hrg.synthetic_clips(3, seed=0, min_frames=300, max_frames=600)
d = hrg.build_model_desc(dict(shield_type="SSM", horizon=1000, seed=9), n_clips=clips.n_clips, env_id="PickPlaceHumanCart")
G = HipBatch(d, clips, n); G.reset()
g = torch.Generator(device="cpu").manual_seed(0)
acts = [(torch.rand((n, 7), generator=g, dtype=torch.float64) * 2 - 1).cuda() for _ in range(8)]
for k in range(10): G.step(acts[k % 8])
torch.cuda.synchronize(); G.kernel_time()
t0 = time.time()
for k in range(50): G.step(acts[k % 8])
torch.cuda.synchronize(); dt = time.time() - t0
ms, nl = G.kernel_time()
print(f"pick-place n={n}: {n*50/dt/1e6:.3f} M env steps/s, kernel {ms:.3f} ms x{nl}")
