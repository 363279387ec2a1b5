"""Multi-process (N > 1) path on CPU: gloo, world_size 2.  Covers sharding, the packed all-gather and the property that
results do not depend on how the global batch is sharded (per-env streams are keyed by the global env id)."""
import os
import socket

import numpy as np
import pytest

import human_robot_gym_amd as hrg
from human_robot_gym_amd import dist as hdist
from human_robot_gym_amd._cstruct import CONST

N_GLOBAL = 12
KW = dict(shield_type="SSM", horizon=6, reward_shaping=True, seed=5, human_rand=[0.2, 0.2, 0.3])


def test_shard_range_partitions_the_batch():
    for n, w in [(4096 * 8, 8), (10, 4), (7, 2), (3, 5)]:
        seen = []
        for r in range(w):
            lo, hi = hdist.shard_range(n, r, w)
            seen += list(range(lo, hi))
        assert seen == list(range(n))


def test_packed_layout_matches_hip_batch_layout():
    lay = hdist.packed_layout(4096)
    od = CONST["HRG_OBS_DIM"]
    assert lay["sizes"] == [4 * 4096 * od, 4 * 4096 * od, 4 * 4096, 4 * 4096 * CONST["HRG_INFO_DIM"], 4096] and all(o % 256 == 0 for o in lay["offsets"])
    blk = np.arange(lay["total"], dtype=np.uint32).astype(np.uint8)
    u = hdist.unpack(blk, 4096)
    assert u["obs"].shape == (4096, od) and u["info"].shape == (4096, CONST["HRG_INFO_DIM"]) and u["done"].shape == (4096,)
    # the terminal observations come last: the head (obs, reward, info, done) is what a step publishes to the other ranks
    assert lay["head"] == lay["offsets"][1] and lay["offsets"][1] > max(lay["offsets"][i] for i in (0, 2, 3, 4))
    head = hdist.unpack(blk[:lay["head"]], 4096)
    assert "term_obs" not in head and (head["info"] == u["info"]).all() and (head["done"] == u["done"]).all() and u["term_obs"].shape == (4096, od)


def _rollout(lo, hi, steps, env_id="ReachHuman"):
    from oracle.oracle import OracleBatch
    clips = hrg.synthetic_clips(2, seed=0, min_frames=200, max_frames=300, inspection=True)
    B = OracleBatch(hrg.build_model_desc(KW, n_clips=clips.n_clips, env_id=env_id), clips, hi - lo, env_id0=lo)
    B.reset()
    out = []
    for k in range(steps):
        a = np.random.RandomState(100 + k).uniform(-1, 1, (N_GLOBAL, 7))[lo:hi]
        out.append(B.step(a) + (B.term_obs.copy(),))
    return out


def _worker(rank, world, port, steps, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = hdist.shard_range(N_GLOBAL, rank, world)
    n = hi - lo
    lay = hdist.packed_layout(n)
    og = hdist.OverlappedGather(torch.zeros(lay["total"], dtype=torch.uint8), world)
    res = []
    for obs, rew, done, info, tobs in _rollout(lo, hi, steps):
        blk = np.zeros(lay["total"], np.uint8)
        u = hdist.unpack(blk, n)
        u["obs"][:], u["term_obs"][:], u["reward"][:], u["info"][:], u["done"][:] = obs, tobs, rew, info, done
        g = hdist.gather_global(torch.from_numpy(blk), n)
        res.append(g)
        og.publish(torch.from_numpy(blk), len(res) - 1)     # the overlapped publisher (synchronous on CPU) must deliver the same blocks
        blocks = og.result(len(res) - 1).numpy()
        for r_ in range(world):
            u2 = hdist.unpack(blocks[r_], n)
            lo2, hi2 = hdist.shard_range(N_GLOBAL, r_, world)
            np.testing.assert_array_equal(u2["obs"], g["obs"][lo2:hi2])
            np.testing.assert_array_equal(u2["done"], g["done"][lo2:hi2])
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        q.put(res)


def test_two_rank_gloo_gather_equals_single_process_batch():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    steps = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, steps, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _rollout(0, N_GLOBAL, steps)
    for k in range(steps):
        obs, rew, done, info, tobs = ref[k]
        np.testing.assert_array_equal(res[k]["obs"], obs)
        np.testing.assert_array_equal(res[k]["term_obs"], tobs)
        np.testing.assert_array_equal(res[k]["reward"], rew)
        np.testing.assert_array_equal(res[k]["info"], info)
        np.testing.assert_array_equal(res[k]["done"], done)
    assert sum(int(r[2].sum()) for r in ref) > 0  # auto-resets happened inside the compared window


@pytest.mark.parametrize("env_id", ["PickPlaceHumanCart", "HumanObjectInspectionCart"])
def test_sharding_does_not_change_results_for_the_cube_tasks(env_id):
    """Object placements, targets, loop properties and animation ids are keyed by the global env id like everything else:
    two shards stepped separately reproduce the single batch bit for bit."""
    steps = 8
    ref = _rollout(0, N_GLOBAL, steps, env_id)
    parts = [_rollout(lo, hi, steps, env_id) for lo, hi in (hdist.shard_range(N_GLOBAL, r, 2) for r in range(2))]
    for k in range(steps):
        for j in range(5):
            np.testing.assert_array_equal(np.concatenate([parts[0][k][j], parts[1][k][j]]), ref[k][j])


# ---------------------------------------------------------------------------------------------- bench.py's own launcher (python bench.py --gpus N)
_STUB = '''
import json, os, sys, time
rank = int(os.environ["RANK"])
keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")
with open(os.path.join(sys.argv[1], f"rank{rank}.json"), "w") as f:
    json.dump({k: os.environ.get(k) for k in keys} | {"argv": sys.argv[2:]}, f)
mode = sys.argv[2]
if mode == "fail" and rank == 1:
    sys.exit(7)            # a rank that dies before the rendezvous
if mode in ("fail", "hang"):
    time.sleep(120)        # ... while its siblings wait for it
print(json.dumps({"metric": "stub", "rank": rank}), flush=True)
'''


def _stub(tmp_path):
    p = tmp_path / "stub_rank.py"
    p.write_text(_STUB)
    return str(p)


def test_spawn_ranks_sets_the_launcher_environment_and_passes_rank0_stdout(tmp_path, capfd):
    """The first real multi-GPU run must not die on plumbing: two children of a stub script see RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, and of their
    standard outputs only rank 0's (the one JSON line) reaches the parent's."""
    import json
    import bench
    rc = bench.spawn_ranks(2, [str(tmp_path), "ok", "--steps", "3"], script=_stub(tmp_path))
    assert rc == 0
    out = capfd.readouterr().out.strip().splitlines()
    assert [json.loads(l) for l in out] == [{"metric": "stub", "rank": 0}]
    seen = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    for r, d in enumerate(seen):
        assert d["RANK"] == str(r) and d["LOCAL_RANK"] == str(r) and d["WORLD_SIZE"] == "2" and d["LOCAL_WORLD_SIZE"] == "2"
        assert d["MASTER_ADDR"] == "127.0.0.1" and d["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and d["argv"] == ["ok", "--steps", "3"]
    assert seen[0]["MASTER_PORT"] == seen[1]["MASTER_PORT"] and int(seen[0]["MASTER_PORT"]) > 0


def test_spawn_ranks_stops_the_siblings_of_a_rank_that_fails(tmp_path):
    """One of three ranks exits 7 while the others would sleep for two minutes (a sibling stuck in the rendezvous): the launcher returns that code promptly and
    leaves no child behind."""
    import time
    import bench
    t0 = time.monotonic()
    rc = bench.spawn_ranks(3, [str(tmp_path), "fail"], script=_stub(tmp_path))
    assert rc == 7 and time.monotonic() - t0 < 30


def test_spawn_ranks_has_a_deadline(tmp_path):
    import time
    import bench
    t0 = time.monotonic()
    rc = bench.spawn_ranks(2, [str(tmp_path), "hang"], script=_stub(tmp_path), deadline_s=1.0)
    assert rc == 124 and time.monotonic() - t0 < 30
