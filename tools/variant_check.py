"""Tuning aid: a variant build must compute bit-identical results.  python tools/variant_check.py [variant.so] -> prints a digest of 30 steps of three tasks."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import human_robot_gym_amd as hrg
from human_robot_gym_amd import _lib, mixed
if len(sys.argv) > 1:
    _lib.use_variant_library(sys.argv[1])
from human_robot_gym_amd._lib import HipBatch
h = hashlib.sha256()
for env_id, n in (("ReachHuman", 1027), ("PickPlaceHumanCart", 515), ("HumanRobotHandoverCart", 130), ("CollaborativeLiftingCart", 66)):
    clips = mixed.task_clips(env_id, 5, min_frames=300, max_frames=600)
    G = HipBatch(hrg.build_model_desc(dict(seed=3, horizon=20), n_clips=5, env_id=env_id), clips, n)
    h.update(G.reset().cpu().numpy().tobytes())
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    for k in range(30):
        o, r, d, i = G.step(torch.rand((n, 7), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1)
        torch.cuda.synchronize()
        for x in (o, r, d, i, G.term_obs):
            h.update(x.cpu().numpy().tobytes())
    G.close()
print("digest", h.hexdigest()[:16], "lib", os.path.basename(_lib.variant_library() or "default"))
