"""Debug aid: step oracle and HIP stacking batches side by side and print the largest state differences per step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import human_robot_gym_amd as hrg
from human_robot_gym_amd.mixed import task_clips
from helpers import flat_state
from oracle.oracle import OracleBatch
from human_robot_gym_amd._lib import HipBatch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
shield = sys.argv[3] if len(sys.argv) > 3 else "OFF"
clips = task_clips("CollaborativeStackingCart", 3, min_frames=400, max_frames=700)
FREE = os.environ.get("FREE") == "1"
kw = dict(shield_type=shield, horizon=int(os.environ.get("HORIZON", 60)), seed=int(os.environ.get("SEED", 2)), control_freq=float(sys.argv[4]) if len(sys.argv) > 4 else 10)
mk = lambda: hrg.build_model_desc(kw, n_clips=3, env_id="CollaborativeStackingCart")
O, G = OracleBatch(mk(), clips, n), HipBatch(mk(), clips, n)
O.reset(); G.reset()
rng = np.random.RandomState(int(os.environ.get("RSEED", 1)))
for k in range(steps):
    a = rng.uniform(-1, 1, (n, 7))
    oo = O.step(a)[0]; og = G.step(torch.from_numpy(a).cuda())[0].cpu().numpy()
    torch.cuda.synchronize()
    d = np.abs(og - oo)
    e, c = np.unravel_index(d.argmax(), d.shape)
    print(f"step {k}: max obs diff {d.max():.3e} at env {e} col {c} (oracle {oo[e, c]:.6f} hip {og[e, c]:.6f}); cols with diff > 1e-6: {sorted(set(np.nonzero(d > 1e-6)[1].tolist()))}")
    for e in range(n):
        names = ([], [])
        fo, io = flat_state(O.get_stack(e), names); fg, ig = flat_state(G.get_stack(e))
        bad = np.nonzero(np.abs(fo - fg) > 1e-7 + 1e-5 * np.abs(fo))[0]
        if (len(bad) or (io != ig).any()) and not FREE:
            print("  env", e, "ints differ" if (io != ig).any() else "", [(names[0][b], fo[b], fg[b]) for b in bad[:6]])
        fo, io = flat_state(O.get_state(e), names := ([], [])); fg, ig = flat_state(G.get_state(e))
        bad = np.nonzero(np.abs(fo - fg) > 1e-7 + 1e-5 * np.abs(fo))[0]
        if (len(bad) or (io != ig).any()) and not FREE:
            print("  env", e, "STATE", "ints differ" if (io != ig).any() else "", [(names[0][b], fo[b], fg[b]) for b in bad[:6]])
        if not FREE:
            G.set_state(e, O.get_state(e)); G.set_stack(e, O.get_stack(e))
    if FREE:
        dmax = []
        for e in range(n):
            fo, _ = flat_state(O.get_stack(e)); fg, _ = flat_state(G.get_stack(e))
            dmax.append(float(np.abs(fo - fg).max()))
        po, no = O.contacts(); pg, ng = G.contacts()
        print("   max |d stack| per env:", " ".join("%.1e" % x for x in dmax), "| ncon", no.tolist(), ng.tolist() if (no != ng).any() else "")
