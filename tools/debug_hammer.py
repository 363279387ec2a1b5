"""Diagnostic: CollaborativeHammeringCart, HIP vs oracle -- where do reset / the first steps differ?  python tools/debug_hammer.py [n_steps]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import human_robot_gym_amd as hrg
from human_robot_gym_amd.mixed import task_clips, task_env_kwargs
from human_robot_gym_amd._lib import HipBatch
from human_robot_gym_amd._cstruct import struct_to_dict
from oracle.oracle import OracleBatch
from helpers import flat_state
ENV = "CollaborativeHammeringCart"
n = 4
clips = task_clips(ENV, 3, min_frames=300, max_frames=420)
kw = dict(task_env_kwargs(ENV), shield_type="OFF", horizon=60, seed=2)
mk = lambda: hrg.build_model_desc(kw, n_clips=clips.n_clips, env_id=ENV)
O, G = OracleBatch(mk(), clips, n), HipBatch(mk(), clips, n)
oo, og = O.reset(), G.reset().cpu().numpy()


def cmp(tag):
    for e in range(1):
        for nm, a, b in (("hammer", O.get_hammer(e), G.get_hammer(e)), ("state", O.get_state(e), G.get_state(e))):
            names = ([], [])
            fo, io = flat_state(a, names)
            fg, ig = flat_state(b)
            bad = np.nonzero(~(np.abs(fo - fg) <= 1e-7 + 1e-5 * np.abs(fo)))[0]
            for k in bad[:16]:
                print(f"  {tag} {nm} {names[0][k]}: oracle {fo[k]!r} hip {fg[k]!r}")
            for k in np.nonzero(io != ig)[0][:8]:
                print(f"  {tag} {nm} {names[1][k]}: oracle {io[k]} hip {ig[k]}")


d = np.abs(oo - og)
print("reset obs: columns that differ", np.nonzero(d.max(0) > 1e-6)[0].tolist())
cmp("reset")
rng = np.random.RandomState(1)
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    a = rng.uniform(-1, 1, (n, 7)) * 0.3
    o_o, r_o, d_o, i_o = O.step(a)
    o_g, r_g, d_g, i_g = G.step(torch.from_numpy(a).cuda())
    torch.cuda.synchronize()
    dd = np.abs(o_o - o_g.cpu().numpy())
    po, no = O.contacts(); pg, ng = G.contacts()
    print(f"step {k}: obs cols differ {np.nonzero(dd.max(0) > 1e-6)[0].tolist()} max {dd.max():.3e} ncon oracle {no.tolist()} hip {ng.tolist()} rew {r_o.tolist()} {r_g.cpu().numpy().tolist()}")
    if (no != ng).any() or (po != pg).any():
        print("   pairs oracle", po[0][:no[0]].tolist(), "hip", pg[0][:ng[0]].tolist())
    cmp(f"step {k}")
    for e in range(n):
        G.set_state(e, O.get_state(e)); G.set_hammer(e, O.get_hammer(e))
