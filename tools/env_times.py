"""Diagnostic (-DHRG_STAMPS build): distribution of per-env wave lifetimes within one step launch."""
import os, sys, ctypes, numpy as np
sys.path.insert(0, '.')
import torch
import human_robot_gym_amd as hrg
from human_robot_gym_amd import _lib
from human_robot_gym_amd._lib import HipBatch, load_library
_lib.use_variant_library("human-robot-gym_amd/variant_stamps.so")   # the -DHRG_STAMPS diagnostic build
lib = load_library()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
clips = hrg.synthetic_clips(13, seed=0)
kw = dict(shield_type="SSM", control_freq=10, horizon=100, done_at_success=True, reward_shaping=True, seed=1234)
G = HipBatch(hrg.build_model_desc(kw, n_clips=13), clips, n); G.reset()
if os.environ.get("HRG_STAGGER", "1") != "0": G.stagger_episode_phases(100 if "ReachHuman" in str(getattr(G, "n", "")) or True else 100)
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
acts = [torch.rand((n, 7), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1 for _ in range(16)]
for k in range(150): G.step(acts[k % 16])
torch.cuda.synchronize()
buf = np.zeros((n, 3), np.uint64)
for rep in range(3):
    obs, r, d, info = G.step(acts[rep]); torch.cuda.synchronize()
    lib.hrg_debug_envcyc(buf.ctypes.data_as(ctypes.c_void_p), n)
    t0 = buf[:, 0].min(); beg = (buf[:, 0] - t0).astype(np.float64); end = (buf[:, 1] - t0).astype(np.float64)
    dur = end - beg
    done = d.cpu().numpy().astype(bool); fs = info[:, 0].cpu().numpy() != 0
    print("launch span %.0f | start spread p50 %.0f max %.0f | dur mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f | end p50 %.0f p99 %.0f max %.0f | resets %d (dur mean %.0f) collisions %d" % (
        end.max(), np.median(beg), beg.max(), dur.mean(), np.median(dur), np.percentile(dur, 90), np.percentile(dur, 99), dur.max(),
        np.median(end), np.percentile(end, 99), end.max(), done.sum(), dur[done].mean() if done.any() else 0, fs.sum()))
st, _ = G.get_states(np.arange(n))
is_safe = np.array([s.is_safe for s in st]); new_goal = np.array([s.new_goal for s in st]); ncon = np.array([s.ncon for s in st])
pv = np.array([s.path_v for s in st]); ts = np.array([s.timestep for s in st])
slow = dur > np.percentile(dur, 88)
for name, m in (("slow", slow), ("rest", ~slow)):
    print(name, "n", m.sum(), "dur", dur[m].mean().round(0), "unsafe %.2f" % (1 - is_safe[m].mean()), "goal pending %.2f" % new_goal[m].mean(), "ncon>0 %.2f" % (ncon[m] > 0).mean(),
          "path_v<0.99 %.2f" % (pv[m] < 0.99).mean(), "timestep==0(reset) %.2f" % (ts[m] == 0).mean())
for lo, hi in ((0, 50), (50, 80), (80, 90), (90, 99), (99, 100)):
    m = (dur >= np.percentile(dur, lo)) & (dur <= np.percentile(dur, hi))
    print("pct %d-%d: dur %.0f unsafe %.2f pending %.2f slowpath %.2f ncon %.2f reset %.2f" % (lo, hi, dur[m].mean(), 1 - is_safe[m].mean(), new_goal[m].mean(), (pv[m] < 0.99).mean(), (ncon[m] > 0).mean(), (ts[m] == 0).mean()))

hw = buf[:, 2]
xcc = ((hw >> np.uint64(32)) & np.uint64(0xf)).astype(int); h = (hw & np.uint64(0xffffffff)).astype(np.int64)
simd = (h >> 4) & 3; cu = (h >> 8) & 15; sh = (h >> 12) & 1; se = (h >> 13) & 7; wave = h & 15
print("xcc ids", np.unique(xcc), "se", np.unique(se), "sh", np.unique(sh), "cu", np.unique(cu), "simd", np.unique(simd))
for x in np.unique(xcc):
    m = xcc == x
    print("xcc", x, "n", m.sum(), "dur mean %.0f p50 %.0f max %.0f" % (dur[m].mean(), np.median(dur[m]), dur[m].max()))
key = xcc * 100000 + se * 10000 + sh * 1000 + cu * 10
ks, cnt = np.unique(key, return_counts=True)
print("CUs used", len(ks), "WGs per CU: min", cnt.min(), "max", cnt.max(), "hist", np.bincount(cnt))
per_cu = np.array([dur[key == k].mean() for k in ks])
print("per-CU mean dur: min %.0f p50 %.0f max %.0f" % (per_cu.min(), np.median(per_cu), per_cu.max()))
for c in np.unique(cnt):
    sel = np.isin(key, ks[cnt == c])
    print("CUs with", c, "WGs: mean dur %.0f" % dur[sel].mean())
k2 = key + simd
k2s, c2 = np.unique(k2, return_counts=True)
print("waves per SIMD hist", np.bincount(c2))
for c in np.unique(c2):
    sel = np.isin(k2, k2s[c2 == c])
    print("SIMDs with", c, "waves: mean dur %.0f" % dur[sel].mean())
