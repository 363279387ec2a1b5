import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import make_pair, flat_state
import human_robot_gym_amd as hrg
from make_golden import CASES
name = sys.argv[1]
g = np.load(f"tests/golden/{name}.npz")
clips = hrg.synthetic_clips(3, seed=0, min_frames=300, max_frames=600)
O, G = make_pair(8, CASES[name], clips=clips)
O.reset(); G.reset()
for k in range(g["actions"].shape[0]):
    a = g["actions"][k]
    O.step(a); G.step(torch.from_numpy(a).cuda())
    worst = []
    for e in range(8):
        names = ([], [])
        fo, io = flat_state(O.get_state(e), names); fg, ig = flat_state(G.get_state(e))
        dd = np.abs(fo - fg) / (1e-3 + np.abs(fo))
        j = int(dd.argmax()); worst.append((dd[j], e, names[0][j], fo[j], fg[j]))
        if (io != ig).any(): print("  int diff", k, e, [(names[1][q], io[q], ig[q]) for q in np.nonzero(io != ig)[0][:6]])
    w = max(worst)
    print(k, "worst %.2e env %d %s %r %r" % w)
