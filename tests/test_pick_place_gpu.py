"""PickPlaceHumanCart: HIP stepper (hrg_step_kernel_box) vs CPU oracle on identical seeded inputs, through the C ABI.  -m gpu."""
import numpy as np
import pytest

from helpers import ATOL, RTOL, assert_state_close, make_pair, record_live
from pp_scenarios import PP, deliver, grasp_and_carry, random_actions, tumble

pytestmark = pytest.mark.gpu


def _rollout(kw, n_envs, n_steps, seed, scenario, resync, min_live=0.9, name="", **desc_kw):
    """Oracle and HIP side by side.  resync=True copies the oracle's state into the HIP batch after every step (per-step
    parity); free-running mode drops an env once its oracle trajectory turned violent (see test_parity_gpu._rollout)."""
    import torch
    O, G = make_pair(n_envs, kw, **PP, **desc_kw)
    oo = O.reset()
    og = G.reset().cpu().numpy()
    np.testing.assert_allclose(og, oo, rtol=RTOL, atol=ATOL)
    for e in range(n_envs):
        assert_state_close(O.get_box(e), G.get_box(e), f"reset env {e} box")
    rng = np.random.RandomState(seed)
    live = np.ones(n_envs, bool)
    stats = dict(gripped=0, success=0, box_contacts=0)
    for k in range(n_steps):
        a = scenario(k, [O, G], rng, n_envs)
        o_o, r_o, d_o, i_o = O.step(a)
        o_g, r_g, d_g, i_g = G.step(torch.from_numpy(np.ascontiguousarray(a)).cuda())
        torch.cuda.synchronize()
        msg = f"step {k}"
        post = [O.get_state(e) for e in range(n_envs)]
        pbox = [O.get_box(e) for e in range(n_envs)]
        violent = np.array([i_o[e, 11] != 0 or max(abs(v) for v in post[e].qvel) > 5.0 or max(abs(v) for v in pbox[e].vel[:3]) > 5.0 for e in range(n_envs)])
        if not resync:
            live &= ~violent
        chk = live & ~violent if resync else live
        np.testing.assert_array_equal(i_g.cpu().numpy()[chk], i_o[chk], err_msg=msg)
        np.testing.assert_array_equal(d_g.cpu().numpy()[chk], d_o[chk], err_msg=msg)
        np.testing.assert_allclose(o_g.cpu().numpy()[chk], o_o[chk], rtol=RTOL, atol=1e-6, err_msg=msg)
        np.testing.assert_allclose(r_g.cpu().numpy()[chk], r_o[chk], rtol=RTOL, atol=1e-6, err_msg=msg)
        np.testing.assert_allclose(G.term_obs.cpu().numpy()[chk], O.term_obs[chk], rtol=RTOL, atol=1e-6, err_msg=msg)
        po, no = O.contacts()
        pg, ng = G.contacts()
        np.testing.assert_array_equal(ng[chk], no[chk], err_msg=msg)
        np.testing.assert_array_equal(pg[chk], po[chk], err_msg=msg)
        stats["gripped"] += int(o_o[chk][:, 39].sum())
        stats["success"] += int((r_o[chk] > 0).sum())
        stats["box_contacts"] += int((po[chk][:, :, 1] == 36).sum())
        for e in range(n_envs):
            if chk[e]:
                assert_state_close(post[e], G.get_state(e), f"{msg} env {e}")
                assert_state_close(pbox[e], G.get_box(e), f"{msg} env {e} box")
            if resync:
                G.set_state(e, post[e])
                G.set_box(e, pbox[e])
    record_live(f"test_pick_place_gpu::{name or getattr(scenario, '__name__', 'scenario')}_{kw.get('shield_type')}{'' if resync else '_free'}", live, min_live)
    desc = O.lib  # keep the library alive until both are closed
    O.close(); G.close()
    del desc
    return stats


@pytest.mark.parametrize("shield", ["OFF", "SSM"])
def test_random_actions_parity(shield):
    kw = dict(shield_type=shield, reward_shaping=True, horizon=25)
    st = _rollout(kw, 16, 40, 2, lambda k, b, rng, n: random_actions(k, b, rng, n), resync=False)
    assert st["box_contacts"] > 0


def test_grasp_carry_release_parity():
    """The cube held between the closing fingers, carried by the arm, released: finger - cube contacts, the coupled
    14-DoF Newton solve, the gripped flag and its reward."""
    import human_robot_gym_amd as hrg
    kw = dict(shield_type="OFF", horizon=200, seed=3)
    d = hrg.build_model_desc(kw, n_clips=3, **PP)
    st = _rollout(kw, 6, 40, 0, lambda k, b, rng, n: grasp_and_carry(k, b, rng, n, d), resync=True)
    assert st["gripped"] > 6 * 20


def test_grasp_free_running_parity():
    import human_robot_gym_amd as hrg
    kw = dict(shield_type="SSM", horizon=200, seed=4)
    d = hrg.build_model_desc(kw, n_clips=3, **PP)
    st = _rollout(kw, 4, 24, 0, lambda k, b, rng, n: grasp_and_carry(k, b, rng, n, d), resync=False, min_live=0.75, name="grasp_free")   # (measured: 4 of 4; one env of margin)
    assert st["gripped"] > 0


def test_tumbling_cube_parity():
    st = _rollout(dict(shield_type="OFF", horizon=100, seed=5), 12, 14, 3, tumble, resync=True)
    assert st["box_contacts"] > 0


def test_box_with_three_different_edges_parity():
    """object_full_size = a brick: per-axis half extents in the narrowphase, the world-frame rotational inertia R diag(I) R' with its
    gyroscopic torque in the solver (tumbling through the air and on the table), and random arm motion around it."""
    brick = dict(object_full_size=[0.04, 0.07, 0.03])
    st = _rollout(dict(shield_type="OFF", horizon=100, seed=5, **brick), 12, 14, 3, tumble, resync=True)
    assert st["box_contacts"] > 0
    _rollout(dict(shield_type="OFF", horizon=100, seed=8, control_freq=50, **brick), 8, 40, 4, tumble, resync=True)   # 5 substeps per step: in the air, too
    _rollout(dict(shield_type="SSM", horizon=25, reward_shaping=True, **brick), 12, 30, 2, lambda k, b, rng, n: random_actions(k, b, rng, n), resync=False)


def test_delivery_parity():
    st = _rollout(dict(shield_type="SSM", horizon=100, seed=6, reward_shaping=False), 8, 12, 1, deliver, resync=False)
    assert st["success"] == 16


def test_collision_prevention_parity():
    kw = dict(shield_type="SSM", horizon=30, seed=7)
    _rollout(kw, 12, 20, 5, lambda k, b, rng, n: random_actions(k, b, rng, n), resync=False, collision_prevention=dict(replace_type=0, n_resamples=20))


def test_full_size_properties():
    """BASELINE config 4 size (8192 envs): no crash, cubes at rest on the table, per-env determinism under re-run."""
    import torch
    import human_robot_gym_amd as hrg
    from human_robot_gym_amd._lib import HipBatch
    clips = hrg.synthetic_clips(3, seed=0, min_frames=300, max_frames=600)
    kw = dict(shield_type="SSM", horizon=1000, seed=9)
    outs = []
    for rep in range(2):
        d = hrg.build_model_desc(kw, n_clips=clips.n_clips, **PP)
        G = HipBatch(d, clips, 8192)
        G.reset()
        g = torch.Generator(device="cpu").manual_seed(0)
        for k in range(6):
            a = (torch.rand((8192, 7), generator=g, dtype=torch.float64) * 2 - 1).cuda()
            obs, r, dn, info = G.step(a)
        torch.cuda.synchronize()
        outs.append((obs.cpu().numpy().copy(), r.cpu().numpy().copy(), info.cpu().numpy().copy()))
        assert not info[:, 11].any().item()
        z = obs[:, 49].cpu().numpy()
        assert np.all(np.abs(z - (d.table_top_z + d.box_half[2])) < 2e-3)
        G.close()
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)


def test_cartesian_action_front_end_parity():
    """IKPositionDeltaWrapper in the kernel: [dx, dy, dz, gripper] rows are converted by the damped-least-squares IK, screened
    by the collision prevention, executed; the executed joint actions written back must agree with the oracle's."""
    import torch
    kw = dict(shield_type="SSM", horizon=40, seed=8)
    ik = dict(action_limit=0.15)
    O, G = make_pair(12, kw, **PP, ik_position_delta=ik, collision_prevention=dict(replace_type=0, n_resamples=20))
    np.testing.assert_allclose(G.reset().cpu().numpy(), O.reset(), rtol=RTOL, atol=ATOL)
    rng = np.random.RandomState(4)
    moved = 0.0
    for k in range(30):
        a = np.zeros((12, 7))
        a[:, :3] = rng.uniform(-0.2, 0.2, (12, 3))
        a[:, 3] = rng.uniform(-1.5, 1.5, 12)
        ag = torch.from_numpy(a.copy()).cuda()
        o_o, r_o, d_o, i_o = O.step(a)
        o_g, r_g, d_g, i_g = G.step(ag)
        torch.cuda.synchronize()
        np.testing.assert_allclose(ag.cpu().numpy(), O.last_actions, rtol=1e-6, atol=1e-9, err_msg=f"executed joint actions, step {k}")
        moved = max(moved, float(np.abs(O.last_actions[:, :6]).max()))
        np.testing.assert_array_equal(i_g.cpu().numpy(), i_o, err_msg=f"step {k}")
        np.testing.assert_array_equal(d_g.cpu().numpy(), d_o)
        np.testing.assert_allclose(o_g.cpu().numpy(), o_o, rtol=RTOL, atol=1e-6, err_msg=f"step {k}")
        for e in range(12):
            assert_state_close(O.get_state(e), G.get_state(e), f"step {k} env {e}")
    assert moved > 0.05
    O.close(); G.close()


def test_pointing_variant_parity():
    """PickPlacePointingHumanCart: the target follows the pointing arm of the human; deliveries re-place the object."""
    import torch
    import human_robot_gym_amd as hrg
    clips = hrg.synthetic_clips(2, seed=0, min_frames=300, max_frames=600)
    clips.infos[1]["pointing_hand"] = "left"
    kw = dict(shield_type="SSM", horizon=30, seed=11, human_rand=[0.2, 0.2, 0.5], reward_shaping=True)
    O, G = make_pair(8, kw, clips=clips, env_id="PickPlacePointingHumanCart")
    np.testing.assert_allclose(G.reset().cpu().numpy(), O.reset(), rtol=RTOL, atol=ATOL)
    rng = np.random.RandomState(2)
    wins = 0
    for k in range(36):
        if k in (5, 14):   # put the cube where the human points
            for e in range(8):
                bx = O.get_box(e)
                for B_ in (O, G):
                    b = B_.get_box(e)
                    b.pos[:] = [bx.target[0], bx.target[1], bx.target[2] + 0.021]
                    for i in range(6):
                        b.vel[i] = 0.0
                    B_.set_box(e, b)
        a = rng.uniform(-1, 1, (8, 7))
        o_o, r_o, d_o, i_o = O.step(a)
        o_g, r_g, d_g, i_g = G.step(torch.from_numpy(a).cuda())
        torch.cuda.synchronize()
        np.testing.assert_array_equal(i_g.cpu().numpy(), i_o, err_msg=f"step {k}")
        np.testing.assert_array_equal(d_g.cpu().numpy(), d_o)
        np.testing.assert_allclose(o_g.cpu().numpy(), o_o, rtol=RTOL, atol=1e-6, err_msg=f"step {k}")
        np.testing.assert_allclose(r_g.cpu().numpy(), r_o, rtol=RTOL, atol=1e-6, err_msg=f"step {k}")
        wins += int(i_o[:, 9].max() > 0)
        for e in range(8):
            assert_state_close(O.get_box(e), G.get_box(e), f"step {k} env {e} box")
    O.close(); G.close()


def test_long_run_stays_finite_and_on_the_table():
    """Soak: 2048 envs x 300 policy steps of random joint + gripper actions with auto-resets (horizon 120).  The cube must stay a
    rigid body on or above the table (or in the gripper, or on the floor once knocked off), nothing may turn non-finite,
    simulation crashes must stay rare."""
    import torch
    import human_robot_gym_amd as hrg
    from human_robot_gym_amd._lib import HipBatch
    clips = hrg.synthetic_clips(3, seed=0, min_frames=300, max_frames=600)
    n = 2048
    d = hrg.build_model_desc(dict(shield_type="SSM", horizon=120, seed=21), n_clips=clips.n_clips, **PP)
    G = HipBatch(d, clips, n)
    G.reset()
    g = torch.Generator(device="cpu").manual_seed(3)
    crashes = dones = 0
    for k in range(300):
        a = (torch.rand((n, 7), generator=g, dtype=torch.float64) * 2 - 1).cuda()
        obs, r, dn, info = G.step(a)
        if k % 25 == 24:
            torch.cuda.synchronize()
            o = obs.cpu().numpy()
            assert np.isfinite(o).all() and np.isfinite(r.cpu().numpy()).all()
            z = o[:, 49]
            on_floor = np.abs(z - (d.floor_z + d.box_half[2])) < 5e-3          # knocked off the table: rests on the floor
            odd = ~(on_floor | (z > d.table_top_z + d.box_half[2] - 5e-3))
            assert odd.sum() <= 2 and (z < 2.0).all(), (k, np.sort(z[odd])[:20], o[odd][:6, 47:50], int(odd.sum()))   # a cube in free fall at the sampling instant
            assert on_floor.mean() < 0.05
        crashes += int(info[:, 11].sum().item()); dones += int(dn.sum().item())
    st, bx = G.get_states(np.arange(0, n, 64))
    for b in bx:
        assert abs(np.linalg.norm(list(b.quat)) - 1.0) < 1e-9 and np.isfinite(list(b.vel)).all()
    assert dones >= 2 * n and crashes <= 0.002 * n * 300, (dones, crashes)
    G.close()


def test_scripted_expert_delivers_on_the_hip_stepper_end_to_end():
    """The product path alone (HipVecEnv + in-kernel IK front-end, no oracle in the loop): the scripted pick-and-place expert of
    tests/test_pick_place.py delivers its cube on (nearly) every env while the human stands out of the way."""
    import human_robot_gym_amd as hrg
    from human_robot_gym_amd.vec_env import HipVecEnv
    n = 32
    env = HipVecEnv(n, env_id="PickPlaceHumanCart", env_kwargs=dict(seed=4, horizon=300, shield_type="SSM"), clips=hrg.static_clip(900, pelvis=(0.0, 1.0, 2.5)),
                    obs_keys=["vec_eef_to_object", "vec_eef_to_target", "object_gripped"], ik_position_delta=dict(action_limit=0.15))
    obs = env.reset()
    wins = np.zeros(n, int)
    for k in range(260):
        v_obj, v_tgt, gr = obs[:, 0:3].astype(float), obs[:, 3:6].astype(float), obs[:, 6] != 0
        a = np.zeros((n, 4))
        for e in range(n):
            if not gr[e]:
                over = np.linalg.norm(v_obj[e, :2]) <= 0.012
                tgt = np.array([v_obj[e, 0], v_obj[e, 1], v_obj[e, 2] + (0.0 if over else 0.08)])
                g = 1.0 if over and abs(v_obj[e, 2]) < 0.02 else -1.0
            else:
                far = np.linalg.norm(v_tgt[e, :2]) > 0.03
                tgt, g = np.array([v_tgt[e, 0], v_tgt[e, 1], max(v_tgt[e, 2] + 0.06, 0.0) if far else v_tgt[e, 2] + 0.03]), 1.0
            a[e, :3], a[e, 3] = np.clip(tgt, -0.05, 0.05), g
        obs, rew, done, infos = env.step(a)
        assert np.isfinite(obs).all() and not done.any()
        wins = np.maximum(wins, [i["n_goal_reached"] for i in infos])
    assert (wins >= 1).sum() >= n - 2, wins
    env.close()


def test_reach_human_with_its_small_box_parity():
    """ReachHuman + smallBox (task HRG_TASK_REACH_BOX, cube kernel with ReachHuman's task logic): HIP vs oracle, incl. an arm that is handed the box."""
    import torch
    import human_robot_gym_amd as hrg
    from oracle.oracle import OracleBatch
    from human_robot_gym_amd._lib import HipBatch
    clips = hrg.synthetic_clips(3, seed=0, min_frames=300, max_frames=600)
    kw = dict(shield_type="SSM", horizon=25, reward_shaping=True, seed=6)
    O = OracleBatch(hrg.build_model_desc(kw, n_clips=3, reach_box=True), clips, 12)
    G = HipBatch(hrg.build_model_desc(kw, n_clips=3, reach_box=True), clips, 12)
    L = HipBatch(hrg.build_model_desc(kw, n_clips=3), clips, 12)                     # the lean kernel: same episodes while nothing touches the box
    np.testing.assert_allclose(G.reset().cpu().numpy(), O.reset(), rtol=RTOL, atol=ATOL)
    L.reset()
    rng = np.random.RandomState(3)
    n_static = 0
    for k in range(40):
        a = rng.uniform(-1, 1, (12, 7))
        if k == 8:    # hand the box to the arm
            for e in range(0, 12, 2):
                s, bx = O.get_state(e), O.get_box(e)
                bx.pos[:] = [s.eef_pos[0], s.eef_pos[1], s.eef_pos[2] - 0.02]
                bx.vel[:] = [0.0] * 6
                O.set_box(e, bx); G.set_box(e, bx)
        o_o, r_o, d_o, i_o = O.step(a)
        o_g, r_g, d_g, i_g = G.step(torch.from_numpy(a).cuda())
        o_l, r_l, d_l, i_l = L.step(torch.from_numpy(a).cuda())
        torch.cuda.synchronize()
        np.testing.assert_array_equal(i_g.cpu().numpy(), i_o, err_msg=f"step {k}")
        np.testing.assert_array_equal(d_g.cpu().numpy(), d_o)
        np.testing.assert_allclose(o_g.cpu().numpy(), o_o, rtol=RTOL, atol=1e-6)
        np.testing.assert_allclose(r_g.cpu().numpy(), r_o, rtol=RTOL, atol=1e-6)
        if k < 8:
            np.testing.assert_allclose(o_l.cpu().numpy(), o_o, rtol=RTOL, atol=1e-6)    # lean model == box model while the box is left alone
        n_static += int(i_o[:, 3].sum())
        for e in range(12):
            post, pbox = O.get_state(e), O.get_box(e)
            assert_state_close(post, G.get_state(e), f"step {k} env {e}")
            assert_state_close(pbox, G.get_box(e), f"step {k} env {e} box")
            G.set_state(e, post); G.set_box(e, pbox)
    assert n_static > 0
    O.close(); G.close(); L.close()
