"""HumanRobotHandoverCart (environments/manipulation/human_robot_handover_cartesian_env.py): the human holds the object by a weld
equality between the object and a mocap body at the holding hand (870-903), presents it in a looping animation, lets go when the robot
has gripped it, waits until it is placed, retreats.  CPU tests run the oracle; the `gpu` test checks the HIP kernel against it.
PARITY UNPINNED; the object is the cube of the pick-place tasks (the reference's HammerObject is robosuite-internal)."""
import numpy as np
import pytest

import human_robot_gym_amd as hrg
from human_robot_gym_amd._cstruct import CONST
from pp_scenarios import between_fingers, put_box

H2R = dict(env_id="HumanRobotHandoverCart")
APPROACH, PRESENT, WAIT, RETREAT, COMPLETE = range(5)
KW = dict(shield_type="OFF", horizon=400, seed=3, done_at_success=False, object_at_target_reward=-0.5)


def _clips():
    return hrg.synthetic_clips(2, seed=0, min_frames=300, max_frames=400, handover=True)


def _scenario(k, batches, n_envs, desc):
    """Even envs: at steps 16-20 the cube is taken out of the hand (weld off) and held between the closing fingers -> gripped -> WAIT;
    from step 30 on it lies at the target -> RETREAT -> COMPLETE.  Odd envs: the human keeps presenting."""
    a = np.zeros((n_envs, 7))
    a[:, 6] = 1.0 if 14 <= k < 30 else -1.0
    for e in range(0, n_envs, 2):
        if 16 <= k <= 20:
            st = batches[0].get_state(e)
            mid, q = between_fingers(desc, list(st.qpos))
            for B in batches:
                bx = B.get_box(e)
                bx.weld_active = 0
                bx.pos[:] = [float(x) for x in mid]; bx.quat[:] = [float(x) for x in q]
                for i in range(6):
                    bx.vel[i] = 0.0
                B.set_box(e, bx)
        elif k >= 30:
            bx = batches[0].get_box(e)
            if bx.task_phase in (WAIT, RETREAT) and not bx.weld_active:
                put_box(batches, e, pos=[bx.target[0], bx.target[1], bx.target[2]], quat=[1, 0, 0, 0], vel=[0] * 6, zero_warm=False)
    return a


def test_desc_defaults_and_clip_info():
    clips = _clips()
    d = hrg.build_model_desc(None, n_clips=clips.n_clips, **H2R)
    assert d.task == CONST["HRG_TASK_HANDOVER_H2R"] and d.shield_type == CONST["HRG_SHIELD_PFL"] and d.done_at_success == 1
    assert list(d.table_half) == [0.5, 1.0] and abs(d.anim_step_length - 250 / 90) < 1e-12 and d.n_targets == 30 and d.n_anim_ids == 20
    np.testing.assert_allclose(list(d.tgt_bin), [0.45 * 0.45, 0.45 * 0.85, -0.95 * 0.15, 0.95 * 0.15])        # 713-730
    assert d.object_at_target_reward == 0.0 and d.object_gripped_reward == -0.25 and d.collision_reward == -1.0
    t = clips.table()
    assert t.clip_n_loop[0] == 2 and t.clip_n_loop2[0] == 1 and [t.clip_holding_hand[0], t.clip_holding_hand[1]] == [0, 1]


def test_weld_follows_the_hand_then_handover_phases():
    from oracle.oracle import OracleBatch
    clips = _clips()
    d = hrg.build_model_desc(KW, n_clips=clips.n_clips, **H2R)
    B = OracleBatch(d, clips, 4)
    B.reset()
    for e in range(4):
        bx, s = B.get_box(e), B.get_state(e)
        assert bx.weld_active == 1 and bx.task_phase == APPROACH
        hand = s.human_site[d.site_lhand] if abs(bx.mocap_pos[0] - s.human_site[d.site_lhand][0]) < 1e-12 else s.human_site[d.site_rhand]
        np.testing.assert_allclose(list(bx.pos), list(hand), atol=1e-12)                 # starts in the holding hand
        assert abs(np.linalg.norm(list(bx.quat)) - 1) < 1e-12
    ph, rew, handed, lag, goals = [], [], [], [], []
    for k in range(75):
        a = _scenario(k, [B], 4, d)
        o, r, dn, info = B.step(a)
        assert not info[:, 11].any()
        bxs = [B.get_box(e) for e in range(4)]
        ph.append([b.task_phase for b in bxs]); rew.append(r.copy()); handed.append(info[:, 13].copy()); goals.append(info[:, 9].copy())
        lag.append([np.linalg.norm(np.array(b.pos) - np.array(b.mocap_pos)) if b.weld_active else np.nan for b in bxs])
    ph, rew, handed, lag, goals = map(np.array, (ph, rew, handed, lag, goals))
    # while welded the object trails the (fast, synthetic) hand by what the soft constraint allows
    assert np.nanmax(lag[:14]) < 0.2 and np.nanmedian(lag[:14]) < 0.1
    assert (ph[:8, 1] == APPROACH).all() and (ph[14:, 1] == PRESENT).all() and (handed[:, 1] == 0).all() and (rew[:, 1] == -1).all()
    # env 0: gripped while PRESENT -> the human lets go, WAIT; object_gripped_reward while held
    first_wait = int(np.argmax(ph[:, 0] == WAIT))
    assert 16 <= first_wait <= 21 and handed[first_wait + 1, 0] == 1 and (rew[first_wait:26, 0] == -0.25).all()   # the info dict predates the transition
    # placed at the target -> RETREAT (object_at_target_reward) -> COMPLETE -> task reward, next animation, the human holds it again
    done_step = int(np.argmax(goals[:, 0] > 0))
    assert done_step > 31 and rew[done_step, 0] == 1.0 and RETREAT in ph[31:done_step, 0] and (rew[32:done_step, 0] == -0.5).all()
    bx = B.get_box(0)
    assert bx.tgt_index == 1 and B.get_state(0).anim_index == 1 and handed[-1, 0] == 1
    assert ph[done_step, 0] == APPROACH and np.isfinite(lag[done_step, 0])               # welded again
    B.close()


@pytest.mark.gpu
def test_hip_matches_oracle_on_the_handover_task():
    import torch
    from helpers import ATOL, RTOL, assert_state_close, make_pair
    clips = _clips()
    kw = dict(KW, shield_type="PFL")
    O, G = make_pair(6, kw, clips=clips, **H2R)
    d = hrg.build_model_desc(kw, n_clips=clips.n_clips, **H2R)
    np.testing.assert_allclose(G.reset().cpu().numpy(), O.reset(), rtol=RTOL, atol=ATOL)
    for e in range(6):
        assert_state_close(O.get_box(e), G.get_box(e), f"reset env {e} box")
    rng = np.random.RandomState(0)
    wins = 0
    for k in range(70):
        a = _scenario(k, [O, G], 6, d)
        a[:, :6] = rng.uniform(-0.2, 0.2, (6, 6)) if not 14 <= k <= 22 else 0.0
        o_o, r_o, d_o, i_o = O.step(a)
        o_g, r_g, d_g, i_g = G.step(torch.from_numpy(a).cuda())
        torch.cuda.synchronize()
        np.testing.assert_array_equal(i_g.cpu().numpy(), i_o, err_msg=f"step {k}")
        np.testing.assert_array_equal(d_g.cpu().numpy(), d_o)
        np.testing.assert_allclose(o_g.cpu().numpy(), o_o, rtol=RTOL, atol=2e-6, err_msg=f"step {k}")
        np.testing.assert_allclose(r_g.cpu().numpy(), r_o, rtol=RTOL, atol=1e-6, err_msg=f"step {k}")
        wins += int((r_o > 0).sum())
        for e in range(6):
            assert_state_close(O.get_state(e), G.get_state(e), f"step {k} env {e}")
            assert_state_close(O.get_box(e), G.get_box(e), f"step {k} env {e} box")
            if k % 8 == 7:
                G.set_state(e, O.get_state(e))
                G.set_box(e, O.get_box(e))
    assert wins >= 3 and i_o[:, 13].max() >= 1
    O.close(); G.close()


# ------------------------------------------------------------------------------------------------ robot -> human
R2H = dict(env_id="RobotHumanHandoverCart")
R_APPROACH, R_REACH_OUT, R_RETREAT, R_COMPLETE = range(4)
# control_freq 50: 5 shield cycles per policy step, so that a cube put into the palm is still there when the step's last collision
# phase looks for the palm contact (at 10 Hz it would have fallen 20 cm by then: only a gripper can hold it there)
KW2 = dict(shield_type="OFF", horizon=1000, seed=3, done_at_success=False, object_in_human_hand_reward=-0.5, control_freq=50)
T_HAND = 40   # policy step of the handover (the hand is held out from ~step 25 on)


def _clips2():
    return hrg.synthetic_clips(2, seed=0, min_frames=120, max_frames=160, handover="r2h")


def _scenario2(k, batches, n_envs):
    """Even envs: at step T_HAND (the human holds the hand out) the cube is put into the palm; odd envs never hand it over."""
    if k == T_HAND:
        for e in range(0, n_envs, 2):
            bx = batches[0].get_box(e)
            put_box(batches, e, pos=list(bx.mocap_pos), vel=[0] * 6, zero_warm=False)
    return np.zeros((n_envs, 7))


def test_robot_to_human_handover_on_the_oracle():
    from oracle.oracle import OracleBatch
    clips = _clips2()
    d = hrg.build_model_desc(KW2, n_clips=clips.n_clips, **R2H)
    assert d.task == CONST["HRG_TASK_HANDOVER_R2H"] and d.goal_dist == 0.06 and d.n_targets == 1
    np.testing.assert_allclose(list(d.obj_bin), [0.70 * 0.45, 0.70 * 0.75, -0.95 * 0.15, 0.95 * 0.15])        # 750-765 on the 1.5 m table (305; RHH-*.yaml:63-66)
    B = OracleBatch(d, clips, 4)
    obs = B.reset()
    for e in range(4):
        bx = B.get_box(e)
        assert bx.weld_active == 0 and d.obj_bin[0] <= bx.pos[0] <= d.obj_bin[1]
        np.testing.assert_allclose(obs[e, 50:53], list(bx.mocap_pos), rtol=1e-6)          # the target is the hand held out (448-450)
    ph, rew, weld, lag = [], [], [], []
    for k in range(130):
        o, r, dn, info = B.step(_scenario2(k, [B], 4))
        assert not info[:, 11].any()
        bxs = [B.get_box(e) for e in range(4)]
        ph.append([b.task_phase for b in bxs]); rew.append(r.copy()); weld.append([b.weld_active for b in bxs])
        want = [np.array(b.mocap_pos) + hrg.model._quat2mat(list(b.mocap_quat)) @ np.array(b.weld_off) for b in bxs]
        lag.append([np.linalg.norm(np.array(b.pos) - w) for b, w in zip(bxs, want)])
        np.testing.assert_allclose(o[0, 50:53], list(bxs[0].mocap_pos) if not (r[0] > 0) else o[0, 50:53], rtol=1e-6)
    ph, rew, weld, lag = map(np.array, (ph, rew, weld, lag))
    assert (ph[:15, 1] == R_APPROACH).all() and (ph[32:, 1] == R_REACH_OUT).all() and (weld[:, 1] == 0).all() and (rew[:, 1] == -1).all()
    T = T_HAND
    assert ph[T - 1, 0] == R_REACH_OUT and ph[T, 0] == R_RETREAT and weld[T, 0] == 1           # palm contact -> the human holds the object
    done_step = int(np.argmax(rew[:, 0] > 0))
    assert done_step > T + 2 and (rew[T + 1:done_step, 0] == -0.5).all() and (ph[T + 1:done_step, 0] == R_RETREAT).all()
    assert np.percentile(lag[T + 2:done_step, 0], 25) < 0.1 and np.max(lag[T + 2:done_step, 0]) < 0.45   # carried along with the hand (the un-choreographed synthetic arm sweeps it through the
                                                                                                 # edge of the 1.5 m table on the way: the soft weld stretches while the cube drags over it)
    assert ph[done_step, 0] == R_APPROACH and weld[done_step, 0] == 0 and B.get_box(0).obj_index == 1   # next round: object back in its bin
    B.close()


@pytest.mark.gpu
def test_hip_matches_oracle_on_the_robot_to_human_handover():
    import torch
    from helpers import ATOL, RTOL, assert_state_close, make_pair
    clips = _clips2()
    kw = dict(KW2, shield_type="PFL")
    O, G = make_pair(6, kw, clips=clips, **R2H)
    np.testing.assert_allclose(G.reset().cpu().numpy(), O.reset(), rtol=RTOL, atol=ATOL)
    for e in range(6):
        assert_state_close(O.get_box(e), G.get_box(e), f"reset env {e} box")
    rng = np.random.RandomState(0)
    wins = 0
    for k in range(130):
        a = _scenario2(k, [O, G], 6)
        a[:, :6] = rng.uniform(-0.2, 0.2, (6, 6))
        o_o, r_o, d_o, i_o = O.step(a)
        o_g, r_g, d_g, i_g = G.step(torch.from_numpy(a).cuda())
        torch.cuda.synchronize()
        np.testing.assert_array_equal(i_g.cpu().numpy(), i_o, err_msg=f"step {k}")
        np.testing.assert_array_equal(d_g.cpu().numpy(), d_o)
        np.testing.assert_allclose(o_g.cpu().numpy(), o_o, rtol=RTOL, atol=2e-6, err_msg=f"step {k}")
        np.testing.assert_allclose(r_g.cpu().numpy(), r_o, rtol=RTOL, atol=1e-6, err_msg=f"step {k}")
        wins += int((r_o > 0).sum())
        for e in range(6):
            assert_state_close(O.get_state(e), G.get_state(e), f"step {k} env {e}")
            assert_state_close(O.get_box(e), G.get_box(e), f"step {k} env {e} box")
            if k % 8 == 7:
                G.set_state(e, O.get_state(e))
                G.set_box(e, O.get_box(e))
    assert wins >= 3
    O.close(); G.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env_id,shield", [("HumanRobotHandoverCart", "PFL"), ("RobotHumanHandoverCart", "PFL"), ("HumanObjectInspectionCart", "SSM")])
def test_collaboration_tasks_long_run_stays_finite(env_id, shield):
    """Soak: 1024 envs x 200 policy steps of random actions with auto-resets (horizon 150): nothing turns non-finite, no simulation
    crashes, the object stays a rigid body, the human's phase machine moves on (the idle loops are reached)."""
    import torch
    from human_robot_gym_amd import mixed
    from human_robot_gym_amd._lib import HipBatch
    n = 1024
    clips = mixed.task_clips(env_id, 5, min_frames=300, max_frames=600)
    d = hrg.build_model_desc(dict(shield_type=shield, horizon=150, seed=31), n_clips=clips.n_clips, env_id=env_id)
    G = HipBatch(d, clips, n)
    G.reset()
    g = torch.Generator(device="cpu").manual_seed(4)
    crashes = 0
    phases = np.zeros(8, int)
    for k in range(200):
        a = (torch.rand((n, 7), generator=g, dtype=torch.float64) * 2 - 1).cuda()
        obs, r, dn, info = G.step(a)
        crashes += int(info[:, 11].sum().item())
        if k % 40 == 39:
            o = obs.cpu().numpy()
            assert np.isfinite(o).all() and np.isfinite(r.cpu().numpy()).all()
            assert (o[:, 49] > 0.015).all() and (o[:, 49] < 2.5).all()         # object height: on the table, in a hand, or -- pushed off the table by the human -- on the floor (z = 0; the cube's half size is 0.02)
            _, bx = G.get_states(np.arange(0, n, 16))
            phases += np.bincount([b.task_phase for b in bx], minlength=8)
            assert max(abs(np.linalg.norm(list(b.quat)) - 1) for b in bx) < 1e-12
    assert crashes == 0
    assert phases[1:].sum() > 0, phases                                       # beyond APPROACH
    G.close()


def test_choreographed_handover_clips_hold_the_hand_out_within_reach():
    """synthetic_clips(handover=..., choreographed=True): the human stands still, facing the robot, and stretches the holding arm out over the table
    between the keyframes (the stand-in for what the HumanRobotHandover / RobotHumanHandover recordings do)."""
    from human_robot_gym_amd.animation import hand_sites
    for ho in (True, "r2h"):
        clips = hrg.synthetic_clips(2, seed=0, min_frames=300, max_frames=400, fps=90.0, handover=ho, choreographed=True)
        o = 0
        for c in range(2):
            F, info = clips.frames[o:o + clips.lengths[c]], clips.infos[c]
            o += clips.lengths[c]
            lh, rh = hand_sites(F, info)
            k0, k1 = info["keyframes"]
            left = info["object_holding_hand"] == "left"
            h, other = (lh, rh) if left else (rh, lh)
            assert np.allclose(h[k0:k1 + 1], h[k0]) and 0.6 < h[k0][0] < 0.72 and 0.95 < h[k0][2] < 1.08 and abs(abs(h[k0][1]) - 0.2) < 0.04   # held out, over the table
            assert (h[k0][1] < 0) == left                                       # facing the robot: left hand at -y
            assert h[0][0] > 1.2 and h[-1][0] > 1.2 and h[0][2] < 0.7           # arms down at the start and the end
            assert np.allclose(other, other[0])                                 # the other arm never moves


@pytest.mark.parametrize("env_id", ["HumanRobotHandoverCart", "RobotHumanHandoverCart", "CollaborativeLiftingCart"])
def test_relative_quaternion_observable_follows_the_reference_formula(env_id):
    """quat_eef_to_object / quat_eef_to_board (human_robot_handover_cartesian_env.py:916-924, collaborative_lifting_cartesian_env.py:1057-1065): the reference feeds
    its (x, y, z, w) observables to quat_to_rot, which expects (w, x, y, z).  The superset's columns 57:61 reproduce exactly that computation -- checked here by
    running the reference's own expression with scipy on the observed object quaternion and the end-effector orientation (sign = convention)."""
    from scipy.spatial.transform import Rotation
    from oracle.oracle import OracleBatch
    from human_robot_gym_amd.mixed import task_clips
    from human_robot_gym_amd.model import robot_fk_numpy
    clips = task_clips(env_id, 2, min_frames=200, max_frames=260)
    d = hrg.build_model_desc(dict(shield_type="OFF", horizon=50, seed=3), n_clips=clips.n_clips, env_id=env_id)
    B = OracleBatch(d, clips, 3)
    obs = B.reset()
    quat_to_rot = lambda q: Rotation.from_quat([q[1], q[2], q[3], q[0]])       # utils/mjcf_utils.py:94-97
    rot_to_quat = lambda r: [r.as_quat()[3], r.as_quat()[0], r.as_quat()[1], r.as_quat()[2]]   # utils/mjcf_utils.py:88-91
    for e in range(3):
        st, bx = B.get_state(e), B.get_box(e)
        R, _ = robot_fk_numpy(d, list(st.qpos))
        eef_xyzw = Rotation.from_matrix(R[6]).as_quat()                          # robot0_eef_quat: T.convert_quat(body_xquat(right_hand), "xyzw")
        obj_xyzw = np.array([bx.quat[1], bx.quat[2], bx.quat[3], bx.quat[0]])   # object_quat / board_quat: (x, y, z, w)
        if env_id != "CollaborativeLiftingCart":
            np.testing.assert_allclose(obs[e, 12:16], obj_xyzw, atol=1e-6)
        q = rot_to_quat(quat_to_rot(obj_xyzw) * quat_to_rot(eef_xyzw).inv())
        want = np.array([q[1], q[2], q[3], q[0]])                                # T.convert_quat(quat, "xyzw")
        got = obs[e, 57:61].astype(np.float64)
        assert np.linalg.norm(got) == pytest.approx(1.0, abs=1e-6)
        assert min(np.abs(got - want).max(), np.abs(got + want).max()) < 2e-6
    B.close()
