import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import make_pair, flat_state
shield = sys.argv[1] if len(sys.argv) > 1 else "OFF"
resync = (sys.argv[2] == "resync") if len(sys.argv) > 2 else True
if resync:
    kw = dict(shield_type=shield, reward_shaping=True, base_human_pos_offset=[1.0, 0.0, 0.0], horizon=20); n = 16; seed = 1
else:
    kw = dict(shield_type=shield, reward_shaping=True, base_human_pos_offset=[0.9, 0.1, 0.0], human_rand=[0.3, 0.3, 0.5], horizon=15); n = 32; seed = 2
O, G = make_pair(n, kw)
O.reset(); G.reset()
rng = np.random.RandomState(seed)
for k in range(45):
    a = rng.uniform(-1, 1, (n, 7))
    o_o, r_o, d_o, i_o = O.step(a)
    o_g, r_g, d_g, i_g = G.step(torch.from_numpy(a).cuda())
    i_g = i_g.cpu().numpy()
    mx = 0
    for e in range(n):
        names = ([], [])
        fo, io = flat_state(O.get_state(e), names); fg, ig = flat_state(G.get_state(e))
        dd = np.abs(fo - fg) / (1e-7 + np.abs(fo))
        if dd.max() > mx: mx = dd.max(); arg = (e, names[0][int(dd.argmax())], fo[int(dd.argmax())], fg[int(dd.argmax())])
        if (io != ig).any():
            print("step", k, "env", e, [(names[1][q], io[q], ig[q]) for q in np.nonzero(io != ig)[0][:10]])
        if resync: G.set_state(e, O.get_state(e))
    print("step", k, "max rel state diff %.3e" % mx, arg)
    if not (i_g == i_o).all():
        idx = np.argwhere(i_g != i_o)
        print("step", k, "info mismatch at", idx.tolist())
        for e in set(idx[:, 0].tolist()):
            print(" env", e, "oracle", i_o[e], "hip", i_g[e])
        break
print("done")
