"""C-ABI surface on the GPU: masked reset, state get/set round trip, capsule taps, kernel timer, HipVecEnv end to end."""
import ctypes

import numpy as np
import pytest

import human_robot_gym_amd as hrg
from helpers import RTOL, assert_state_close, make_pair

pytestmark = pytest.mark.gpu


def test_masked_reset_and_state_roundtrip():
    import torch
    O, G = make_pair(8, dict(shield_type="SSM", horizon=50))
    O.reset(); G.reset()
    rng = np.random.RandomState(0)
    for _ in range(3):
        a = rng.uniform(-1, 1, (8, 7))
        O.step(a); G.step(torch.from_numpy(a).cuda())
    mask = np.array([1, 0, 0, 1, 0, 1, 0, 0], np.uint8)
    before = [G.get_state(e) for e in range(8)]
    oo = O.reset(mask)
    og = G.reset(torch.from_numpy(mask).cuda()).cpu().numpy()
    np.testing.assert_allclose(og[mask == 1], oo[mask == 1], rtol=RTOL, atol=1e-6)
    for e in range(8):
        sg = G.get_state(e)
        assert_state_close(O.get_state(e), sg, f"env {e}")
        if not mask[e]:
            assert bytes(sg) == bytes(before[e])          # untouched envs keep their block bit for bit
        else:
            assert sg.episode == before[e].episode + 1 and sg.timestep == 0
    s = G.get_state(2)
    s.qpos[0] = 0.123
    G.set_state(5, s)
    assert bytes(G.get_state(5)) == bytes(s)
    O.close(); G.close()


def test_reach_capsule_taps_match_oracle():
    import torch
    O, G = make_pair(8, dict(shield_type="SSM", horizon=50))
    G.enable_taps(True)
    O.reset(); G.reset()
    rng = np.random.RandomState(1)
    for _ in range(4):
        a = rng.uniform(-1, 1, (8, 7))
        O.step(a); G.step(torch.from_numpy(a).cuda())
    ro, ho, no = O.capsules()
    rg, hg, ng = G.capsules()
    np.testing.assert_array_equal(ng, no)
    np.testing.assert_allclose(rg, ro, rtol=RTOL, atol=1e-7)   # SafetyShield.getRobotReachCapsules (failsafe_controller.py:393)
    for e in range(8):
        np.testing.assert_allclose(hg[e, :no[e]], ho[e, :no[e]], rtol=RTOL, atol=1e-7)  # getHumanReachCapsules (:416)
    assert (ro[:, :, 6] > 0).all() and no.min() > 30
    O.close(); G.close()


def test_kernel_timer_and_errors():
    import torch
    from human_robot_gym_amd._lib import HipBatch, HrgError
    clips = hrg.synthetic_clips(2, seed=0, min_frames=100, max_frames=120)
    G = HipBatch(hrg.build_model_desc(dict(shield_type="OFF"), n_clips=2), clips, 64)
    G.reset()
    assert G.kernel_time() == (0.0, 0)
    a = torch.zeros((64, 7), dtype=torch.float64, device="cuda")
    for _ in range(3):
        G.step(a)
    ms, n = G.kernel_time()
    assert n == 3 and 0 < ms < 1000
    with pytest.raises(ValueError):
        G.step(torch.zeros((63, 7), dtype=torch.float64, device="cuda"))
    G.close()
    bad = hrg.build_model_desc(dict(shield_type="SSM"), n_clips=2)
    bad.failsafe_sdot = 0.3                                # SSM / OFF brake to a full stop; PFL takes its speed from pfl_v_safe
    with pytest.raises(HrgError, match="failsafe_sdot"):
        HipBatch(bad, clips, 4)


def test_hip_vec_env_end_to_end_matches_oracle_backend():
    from human_robot_gym_amd.vec_env import HipVecEnv
    from helpers import OracleBackend
    kw = dict(shield_type="SSM", horizon=6, reward_shaping=True)
    cp = dict(replace_type=0, n_resamples=20)
    clips = hrg.synthetic_clips(2, seed=0, min_frames=200, max_frames=300)
    desc = hrg.build_model_desc(kw, n_clips=2, collision_prevention=cp)
    e_gpu = HipVecEnv(8, env_kwargs=kw, clips=clips, collision_prevention=cp)
    e_cpu = HipVecEnv(8, env_kwargs=kw, clips=clips, collision_prevention=cp, backend=OracleBackend(desc, clips, 8))
    np.testing.assert_allclose(e_gpu.reset(), e_cpu.reset(), rtol=RTOL, atol=1e-6)
    rng = np.random.RandomState(0)
    for k in range(14):
        a = rng.uniform(-1, 1, (8, 7))
        og, rg, dg, ig = e_gpu.step(a)
        oc, rc, dc, ic = e_cpu.step(a)
        np.testing.assert_allclose(og, oc, rtol=RTOL, atol=1e-6)
        np.testing.assert_allclose(rg, rc, rtol=RTOL, atol=1e-6)
        np.testing.assert_array_equal(dg, dc)
        for i in range(8):
            assert set(ig[i]) == set(ic[i])
            for key in ig[i]:
                if key in ("terminal_observation", "action"):
                    np.testing.assert_allclose(ig[i][key], ic[i][key], rtol=RTOL, atol=1e-6)
                elif key == "episode":
                    assert ig[i][key]["l"] == ic[i][key]["l"] and ig[i][key]["r"] == pytest.approx(ic[i][key]["r"], rel=1e-5)
                else:
                    assert ig[i][key] == ic[i][key], key
    e_gpu.close(); e_cpu.close()


def test_sharded_batches_reproduce_the_global_batch_bit_exactly():
    """(e) multi-GPU: rank r owns env ids [r*n, (r+1)*n); per-env streams are keyed by the global id, so two shards on one
    GPU must reproduce the single global batch bit for bit (no cross-env data path exists)."""
    import torch
    from human_robot_gym_amd._lib import HipBatch
    kw = dict(shield_type="SSM", horizon=12, reward_shaping=True, human_rand=[0.2, 0.2, 0.3], seed=9)
    clips = hrg.synthetic_clips(3, seed=0, min_frames=200, max_frames=300)
    mk = lambda n, id0: HipBatch(hrg.build_model_desc(kw, n_clips=3), clips, n, env_id0=id0)  # noqa: E731
    G, A, B = mk(64, 0), mk(32, 0), mk(32, 32)
    og = G.reset().clone()
    torch.testing.assert_close(torch.cat([A.reset(), B.reset()]), og, rtol=0, atol=0)
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    for k in range(30):
        a = torch.rand((64, 7), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
        G.step(a.clone()); A.step(a[:32].clone()); B.step(a[32:].clone())
        torch.cuda.synchronize()
        for name in ("obs", "term_obs", "reward", "done", "info"):
            torch.testing.assert_close(torch.cat([getattr(A, name), getattr(B, name)]), getattr(G, name), rtol=0, atol=0, msg=f"{name} step {k}")
    assert int(G.info[:, 7].sum()) >= 0
    for x in (G, A, B):
        x.close()


def test_long_run_is_deterministic_and_finite():
    """1000 policy steps (25 000 substeps) of 256 envs twice: identical outputs run to run, no NaN, crashes stay rare."""
    import torch
    from human_robot_gym_amd._lib import HipBatch
    kw = dict(shield_type="SSM", horizon=100, reward_shaping=True, done_at_success=True, seed=3)
    clips = hrg.synthetic_clips(4, seed=1, min_frames=400, max_frames=800)
    outs = []
    for rep in range(2):
        G = HipBatch(hrg.build_model_desc(kw, n_clips=4), clips, 256)
        G.reset()
        gen = torch.Generator(device="cuda"); gen.manual_seed(5)
        acc = torch.zeros(4, dtype=torch.float64, device="cuda")
        crashes = 0
        for k in range(1000):
            o, r, d, i = G.step(torch.rand((256, 7), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1)
            acc += torch.stack([o.double().sum(), r.double().sum(), d.double().sum(), i.double().sum()])
            crashes += int(i[:, 11].sum())
        torch.cuda.synchronize()
        assert torch.isfinite(acc).all() and crashes < 0.02 * 256 * 1000 / 100
        outs.append((acc.cpu(), crashes))
        G.close()
    assert torch.equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]


def test_batched_state_io_reference_state_initialisation():
    """hrg_batch_get_states / set_states: copying the states of one batch into (permuted) envs of another makes those envs
    continue identically — the batched form of get/set_environment_state used for reference-state initialisation."""
    import torch
    import human_robot_gym_amd as hrg
    from human_robot_gym_amd._lib import HipBatch
    clips = hrg.synthetic_clips(3, seed=0, min_frames=300, max_frames=600)
    kw = dict(shield_type="SSM", horizon=50, seed=3)
    A = HipBatch(hrg.build_model_desc(kw, n_clips=clips.n_clips, env_id="PickPlaceHumanCart"), clips, 8)
    B = HipBatch(hrg.build_model_desc(kw, n_clips=clips.n_clips, env_id="PickPlaceHumanCart"), clips, 8, env_id0=100)
    A.reset(); B.reset()
    g = torch.Generator().manual_seed(1)
    for _ in range(5):
        A.step((torch.rand((8, 7), generator=g, dtype=torch.float64) * 2 - 1).cuda())
    perm = np.array([3, 1, 7, 5], np.int32)
    st, bx = A.get_states([0, 1, 2, 3])
    B.set_states(perm, st, bx)
    st2, bx2 = B.get_states(perm)
    assert bytes(st2) == bytes(st) and bytes(bx2) == bytes(bx)
    a = (torch.rand((8, 7), generator=g, dtype=torch.float64) * 2 - 1)
    aB = a.clone()
    aB[torch.from_numpy(perm.astype(np.int64))] = a[:4]
    oA, rA, dA, iA = (x.cpu().numpy().copy() for x in A.step(a.cuda()))
    oB, rB, dB, iB = (x.cpu().numpy().copy() for x in B.step(aB.cuda()))
    # human placement / animation are part of the state block; per-env random streams (resets, resampling) are keyed by the env id
    # and not exercised within this step
    np.testing.assert_array_equal(oB[perm], oA[:4])
    np.testing.assert_array_equal(rB[perm], rA[:4])
    np.testing.assert_array_equal(iB[perm], iA[:4])
    A.close(); B.close()


@pytest.mark.gpu
def test_vec_env_state_snapshots_and_joint_pos():
    """HipVecEnv.get_environment_state / set_environment_state (HumanEnv.get/set_environment_state, human_env.py:1845-1900) and the
    `joint_pos` attribute the IK wrapper reads (ik_position_delta_wrapper.py:107): restoring a snapshot replays the same steps."""
    from human_robot_gym_amd.vec_env import HipVecEnv
    for env_id in ("ReachHuman", "PickPlaceHumanCart"):
        clips = hrg.synthetic_clips(3, seed=0, min_frames=300, max_frames=600)
        env = HipVecEnv(6, env_id=env_id, env_kwargs=dict(seed=4, horizon=50), clips=clips, obs_keys=["robot0_joint_pos", "robot0_eef_pos"])
        env.reset()
        rng = np.random.RandomState(0)
        acts = [rng.uniform(-1, 1, (6, 7)) for _ in range(8)]
        for a in acts[:3]:
            obs, _, _, _ = env.step(a)
        np.testing.assert_allclose(np.stack(env.get_attr("joint_pos")), obs[:, :6], rtol=0, atol=1e-6)
        assert len(env.get_attr("joint_pos", indices=[1, 4])) == 2
        snap = env.get_environment_state()
        assert len(snap) == 6 and (snap[0][1] is None) == (env_id == "ReachHuman")
        first = [env.step(a)[0] for a in acts[3:]]
        env.set_environment_state(snap)
        again = [env.step(a)[0] for a in acts[3:]]
        for x, y in zip(first, again):
            np.testing.assert_array_equal(x, y)
        part = env.get_environment_state(indices=[2, 5])
        env.set_environment_state(part, indices=[0, 1])           # envs 0, 1 continue as copies of 2, 5 (the states carry their random streams)
        o = env.step(np.tile(acts[0][:1], (6, 1)))[0]              # the same action everywhere
        np.testing.assert_array_equal(o[0], o[2]); np.testing.assert_array_equal(o[1], o[5])
        env.close()


@pytest.mark.gpu
def test_batch_sizes_from_one_env_to_65536():
    """Edge sizes: a single env, a ragged count (not a multiple of anything), and 65 536 envs (16 x the benchmark batch) run the same episodes as the
    corresponding rows of a 64-env batch; a batch of zero envs is refused."""
    import torch
    from human_robot_gym_amd._lib import HipBatch, HrgError
    kw = dict(shield_type="SSM", horizon=20, reward_shaping=True, seed=12)
    clips = hrg.synthetic_clips(3, seed=0, min_frames=200, max_frames=300)
    mk = lambda n: HipBatch(hrg.build_model_desc(kw, n_clips=3), clips, n)  # noqa: E731
    with pytest.raises(HrgError):
        mk(0)
    ref, one, ragged, big = mk(64), mk(1), mk(37), mk(65536)
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    o = ref.reset().clone()
    torch.testing.assert_close(one.reset(), o[:1], rtol=0, atol=0)
    torch.testing.assert_close(ragged.reset(), o[:37], rtol=0, atol=0)
    torch.testing.assert_close(big.reset()[:64], o, rtol=0, atol=0)
    for k in range(6):
        a = torch.rand((65536, 7), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
        ref.step(a[:64].clone()); one.step(a[:1].clone()); ragged.step(a[:37].clone()); big.step(a)
        torch.cuda.synchronize()
        for name in ("obs", "reward", "done", "info"):
            torch.testing.assert_close(getattr(one, name), getattr(ref, name)[:1], rtol=0, atol=0)
            torch.testing.assert_close(getattr(ragged, name), getattr(ref, name)[:37], rtol=0, atol=0)
            torch.testing.assert_close(getattr(big, name)[:64], getattr(ref, name), rtol=0, atol=0)
        assert torch.isfinite(big.obs).all() and torch.isfinite(big.reward).all()
    for x in (ref, one, ragged, big):
        x.close()


@pytest.mark.gpu
def test_bench_gather_branch_in_process():
    """bench.py's N > 1 exchange (`make_gather`: one RCCL all-gather of the packed head per step) run in-process with world size 1 — the only
    multi-GPU rehearsal this one-GPU box allows: the gathered block equals `packed_head` byte for byte, for the single-task and the mixed batch.
    No 1 -> 8 scaling curve exists yet."""
    import os
    import torch
    import torch.distributed as dist
    import bench
    from human_robot_gym_amd import mixed
    from human_robot_gym_amd._lib import HipBatch
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        clips = hrg.synthetic_clips(3, seed=0, min_frames=200, max_frames=300)
        G = HipBatch(hrg.build_model_desc(dict(shield_type="SSM", horizon=20, seed=1), n_clips=3), clips, 256)
        M = mixed.make_mixed_batch(66, seed=1, n_clips=3)
        for B in (G, M):
            B.reset()
            publish, finish, gathered = bench.make_gather(B, 1, "serial")
            gen = torch.Generator(device="cuda"); gen.manual_seed(0)
            for k in range(3):
                B.step(torch.rand((B.n, 7), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1)
                publish(k)
            finish()
            torch.cuda.synchronize()
            assert gathered.numel() == B.packed_head.numel() and torch.equal(gathered, B.packed_head)
            from human_robot_gym_amd.dist import unpack
            u = unpack(gathered.cpu().numpy(), B.n)
            np.testing.assert_array_equal(u["obs"], B.obs.cpu().numpy())
            np.testing.assert_array_equal(u["done"], B.done.cpu().numpy())
            B.close()
    finally:
        dist.destroy_process_group()


def test_launch_order_stays_a_permutation_with_the_busy_envs_in_front():
    """The step kernel files every env into the next launch's order (busy envs from the front).  Whatever the order, it has to be a permutation of the batch,
    and an env under a fail-safe manoeuvre has to start among the first."""
    import torch
    n = 512
    clips = hrg.synthetic_clips(5, seed=0)
    d = hrg.build_model_desc(dict(shield_type="SSM", horizon=60, seed=9), n_clips=clips.n_clips)
    G = hrg.HipBatch(d, clips, n)
    G.reset()
    order, nb = G.launch_order()
    assert np.array_equal(order, np.arange(n)) and nb == 0
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    seen_busy = 0
    for k in range(45):
        a = torch.rand((n, 7), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
        obs, rew, done, info = G.step(a)
        if k % 9 == 8:
            order, nb = G.launch_order()
            assert np.array_equal(np.sort(order), np.arange(n)), "the launch order lost or duplicated an env"
            assert 0 <= nb <= n
            st, _ = G.get_states(np.arange(n))
            unsafe = np.array([not s.is_safe for s in st]) & ~done.cpu().numpy().astype(bool)
            front = np.zeros(n, bool); front[order[:nb]] = True
            assert np.all(front[unsafe]), "an env under a fail-safe manoeuvre is not among the first to start"
            seen_busy = max(seen_busy, nb)
    assert seen_busy > 0          # random actions next to a moving human: some env brakes within 45 steps
    G.close()


def test_reference_demo_configuration_hip_matches_oracle_backend():
    """BASELINE configs[0] (demos/demo_reach_human_environment.py: one env, control_freq 5, horizon 1000, SSM, CollisionPreventionWrapper) through HipGymEnv:
    the HIP stepper and the oracle backend, the demo's loop, step by step."""
    from human_robot_gym_amd.vec_env import HipGymEnv
    from helpers import OracleBackend
    from test_vec_env import DEMO_KW, DEMO_CP, demo_rollout
    clips = hrg.synthetic_clips(3, seed=0, min_frames=600, max_frames=900)
    desc = hrg.build_model_desc(DEMO_KW, n_clips=3, collision_prevention=DEMO_CP)
    e_gpu = HipGymEnv(env_kwargs=DEMO_KW, clips=clips, collision_prevention=DEMO_CP)
    e_cpu = HipGymEnv(env_kwargs=DEMO_KW, clips=clips, collision_prevention=DEMO_CP, backend=OracleBackend(desc, clips, 1))
    og, oc = demo_rollout(e_gpu), demo_rollout(e_cpu)
    assert len(og) == len(oc) == 100
    for k, ((ob_g, r_g, d_g, i_g), (ob_c, r_c, d_c, i_c)) in enumerate(zip(og, oc)):
        np.testing.assert_allclose(ob_g, ob_c, rtol=RTOL, atol=1e-6, err_msg=f"step {k}")
        assert r_g == pytest.approx(r_c, rel=1e-5, abs=1e-6) and d_g == d_c
        for key in ("failsafe_interventions", "n_collisions", "n_goal_reached", "action_resamples"):
            assert i_g[key] == i_c[key], (k, key)
    e_gpu.close(); e_cpu.close()
