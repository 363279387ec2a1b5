"""Diagnostic (-DHRG_STAMPS build): which SIMD does workgroup i of a step launch land on?  Prints how many distinct SIMDs the first 256 / 512 / 1024 / ... workgroups cover."""
import os, sys, ctypes, numpy as np
sys.path.insert(0, '.')
import torch
import human_robot_gym_amd as hrg
from human_robot_gym_amd import _lib
from human_robot_gym_amd._lib import HipBatch, load_library
_lib.use_variant_library("human-robot-gym_amd/variant_stamps.so")
lib = load_library()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
clips = hrg.synthetic_clips(13, seed=0)
kw = dict(shield_type="SSM", control_freq=10, horizon=100, done_at_success=True, reward_shaping=True, seed=1234)
G = HipBatch(hrg.build_model_desc(kw, n_clips=13), clips, n); G.reset()
G.stagger_episode_phases(100)
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
acts = [torch.rand((n, 7), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1 for _ in range(16)]
for k in range(20): G.step(acts[k % 16])
torch.cuda.synchronize()
buf = np.zeros((n, 3), np.uint64)
prev = None
for rep in range(4):
    G.step(acts[rep]); torch.cuda.synchronize()
    lib.hrg_debug_envcyc(buf.ctypes.data_as(ctypes.c_void_p), n)
    hw = buf[:, 2]
    xcc = ((hw >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64); h = (hw & np.uint64(0xffffffff)).astype(np.int64)
    simd = (h >> 4) & 3; cu = (h >> 8) & 15; sh = (h >> 12) & 1; se = (h >> 13) & 7; wave = h & 15
    sid = ((xcc * 8 + se) * 2 + sh) * 16 * 4 + cu * 4 + simd
    print("launch", rep, "distinct SIMDs used", len(np.unique(sid)), "| same placement as previous launch: %s" % (None if prev is None else float((prev == sid).mean())))
    for w in (256, 512, 1024, 2048):
        for off in (0, w):
            if off + w <= n: print("   workgroups [%d, %d): distinct SIMDs %d, distinct CUs %d" % (off, off + w, len(np.unique(sid[off:off + w])), len(np.unique(sid[off:off + w] // 4))))
    print("   first 24 workgroups (xcc,se,cu,simd,slot):", [(int(xcc[i]), int(se[i]), int(cu[i]), int(simd[i]), int(wave[i])) for i in range(24)])
    for st in (8, 32, 256, 1024):
        i0 = 5
        print("   stride", st, "from wg", i0, ":", [(int(xcc[i]), int(se[i]), int(cu[i]), int(simd[i])) for i in range(i0, min(n, i0 + 6 * st), st)])
    prev = sid
