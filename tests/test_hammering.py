"""CollaborativeHammeringCart on the CPU oracle (collaborative_hammering_cartesian_env.py): model constants, box-box contacts of boxes with different
extents, reset state, observation columns, the three-phase machine and its rewards, the nail's slide joint under the hammer.  The GPU legs (HIP vs
oracle) live in tests/test_hammering_gpu.py."""
import ctypes

import numpy as np
import pytest

import human_robot_gym_amd as hrg
from human_robot_gym_amd._cstruct import CONST
from human_robot_gym_amd.mixed import task_clips, task_env_kwargs

ENV = "CollaborativeHammeringCart"


def _batch(n=2, clips=None, **kw):
    from oracle.oracle import OracleBatch
    clips = clips or task_clips(ENV, 3, min_frames=300, max_frames=400)
    env_kw = dict(shield_type="SSM", horizon=400, **task_env_kwargs(ENV))
    env_kw.update(kw)
    d = hrg.build_model_desc(env_kw, n_clips=clips.n_clips, env_id=ENV)
    return OracleBatch(d, clips, n), d, clips


def _quat2mat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def test_model_constants_follow_the_reference_files():
    d = hrg.build_model_desc(None, n_clips=1, env_id=ENV)
    assert d.task == CONST["HRG_TASK_HAMMERING"]
    np.testing.assert_allclose(d.hm_board_half[:], [0.5, 0.2, 0.015])                       # board_full_size (1.0, 0.4, 0.03), 305
    assert np.isclose(d.hm_board_mass, 12.0)                                                # BoxObject default density 1000
    np.testing.assert_allclose(d.hm_board_inertia[:], [12 * (0.2 ** 2 + 0.015 ** 2) / 3, 12 * (0.5 ** 2 + 0.015 ** 2) / 3, 12 * (0.5 ** 2 + 0.2 ** 2) / 3])
    np.testing.assert_allclose(d.hm_anchor[0][:], [-0.1, 0.2, 0.0]); np.testing.assert_allclose(d.hm_anchor[1][:], [-0.5, -0.2, 0.0])   # 1001-1002
    np.testing.assert_allclose(d.hm_weld_relquat[:], [0, 0, 0, 1])                          # relpose "0 0 0 0 0 0 1", 1128
    assert np.isclose(d.hm_nail_mass, 1000 * np.pi * 0.02 ** 2 * 0.004)                     # nail.xml:5
    assert np.isclose(d.hm_nail_z0, 0.015 + 0.001 + 0.01 + 0.06) and d.hm_nail_range == 0.06 and d.hm_nail_frictionloss == 10000.0
    np.testing.assert_allclose(d.hm_nail_bin[:], [0.05, 0.45, -0.18, 0.18])                 # 838-853
    assert d.n_obj_placements == 10 and d.gripper_controllable == 0 and d.hm_goal_tolerance == 0.05   # horizon 1000 x 1 / 100
    np.testing.assert_allclose(d.init_qpos[:], [0, 0, -np.pi / 2, 0, -np.pi / 2, np.pi / 4])          # 720
    # the stand-in hammer: COM between handle and head, parallel-axis inertia
    m1, m2 = 175 * 0.035 * 0.035 * 0.175, 350 * 0.1232 * 0.0385 * 0.0385
    assert np.isclose(d.hm_hammer_mass, m1 + m2)
    assert np.isclose(d.hm_hammer_com[2], m2 * (0.0875 + 0.01925) / (m1 + m2))
    assert d.hm_hammer_inertia[2] < d.hm_hammer_inertia[0] < d.hm_hammer_inertia[1]


def _boxbox2(lib, pa, qa, ha, pb, qb, hb):
    out = np.zeros(28)
    a = [np.ascontiguousarray(x, np.float64) for x in (pa, qa, ha, pb, qb, hb)]
    n = lib.hrgo_test_boxbox2(*[x.ctypes.data_as(ctypes.c_void_p) for x in a], out.ctypes.data_as(ctypes.c_void_p))
    return out[:7 * n].reshape(n, 7)


def test_box_box_with_different_extents(oracle_lib):
    board, head, nail = [0.5, 0.2, 0.015], [0.0616, 0.01925, 0.01925], [0.02, 0.02, 0.002]
    pen = 2e-4
    # the hammer head lying flat on the board: the four corners of ITS face (the smaller one), normal from the head down into the board
    c = _boxbox2(oracle_lib, [0.1, 0.05, 0.015 + 0.01925 - pen], [1, 0, 0, 0], head, [0, 0, 0], [1, 0, 0, 0], board)
    assert c.shape == (4, 7) and np.allclose(c[:, 6], -pen) and np.allclose(c[:, 3:6], [0, 0, -1.0])
    assert np.isclose(c[:, 0].max() - c[:, 0].min(), 2 * head[0]) and np.isclose(c[:, 1].max() - c[:, 1].min(), 2 * head[1])
    assert np.isclose(c[:, 0].mean(), 0.1) and np.isclose(c[:, 1].mean(), 0.05)
    # the head (larger) on the nail head (smaller): the nail's whole top face is covered -> its four corners
    c = _boxbox2(oracle_lib, [0, 0, 0.002 + 0.01925 - pen], [1, 0, 0, 0], head, [0, 0, 0], [1, 0, 0, 0], nail)
    assert c.shape == (4, 7) and np.allclose(c[:, 6], -pen)
    assert np.allclose(sorted(map(tuple, np.round(c[:, :2] / 0.02, 9))), [(-1, -1), (-1, 1), (1, -1), (1, 1)][:4]) or np.isclose(c[:, 0].max() - c[:, 0].min(), 0.04)
    assert np.isclose(c[:, 1].max() - c[:, 1].min(), 2 * min(head[1], nail[1]))
    # equal extents reproduce the single-extent generator
    out = np.zeros(28)
    args = [np.ascontiguousarray(x, np.float64) for x in ([0, 0, 0], [1, 0, 0, 0], [0.01, 0, 0.0448], [0.98, 0.1, 0.05, 0.15], [0.0225] * 3)]
    args[3] /= np.linalg.norm(args[3])
    n = oracle_lib.hrgo_test_boxbox(*[x.ctypes.data_as(ctypes.c_void_p) for x in args], out.ctypes.data_as(ctypes.c_void_p))
    c2 = _boxbox2(oracle_lib, args[0], args[1], args[4], args[2], args[3], args[4])
    np.testing.assert_array_equal(out[:7 * n].reshape(n, 7), c2)
    # separated -> nothing
    assert len(_boxbox2(oracle_lib, [0, 0, 0.2], [1, 0, 0, 0], head, [0, 0, 0], [1, 0, 0, 0], board)) == 0


def test_reset_puts_board_in_the_hands_and_hammer_in_the_gripper():
    B, d, _ = _batch()
    obs = B.reset()
    for e in range(2):
        hm, st = B.get_hammer(e), B.get_state(e)
        Rb = _quat2mat(hm.quat[0])
        # the weld is satisfied exactly: right grip at the right-hand mocap body; the synthetic human holds the board level, nail side towards the robot
        np.testing.assert_allclose(np.array(hm.pos[0]) + Rb @ np.array(d.hm_anchor[1][:]), hm.mocap_pos[1], atol=1e-12)
        assert Rb[2, 2] > 0.999 and Rb[0, 0] < -0.99
        assert np.linalg.norm(np.array(hm.pos[0]) + Rb @ np.array(d.hm_anchor[0][:]) - np.array(hm.mocap_pos[0])) < 0.08    # the connect's anchor nearly at the left hand
        assert 0.05 <= hm.nail_xy[0] <= 0.45 and abs(hm.nail_xy[1]) <= 0.18 and hm.nail_q == 0 and hm.task_phase == CONST["HRG_HM_APPROACH"]
        # hammer root body at the grip site, turned 90 deg about y: handle along world x, head in front
        np.testing.assert_allclose(hm.obs_pos[1], st.eef_pos, atol=1e-12)
        np.testing.assert_allclose(hm.quat[1], [np.sqrt(0.5), 0, np.sqrt(0.5), 0], atol=1e-12)
        assert hm.pos[1][0] - st.eef_pos[0] == pytest.approx(d.hm_hammer_com[2])
        np.testing.assert_allclose(st.qpos[6:8], d.hm_finger_grip_qpos[:])
        o = obs[e]
        np.testing.assert_allclose(o[12:16], hm.quat[1], atol=1e-6)                      # hammer_quat (w, x, y, z)
        np.testing.assert_allclose(o[33:36], hm.pos[0], atol=1e-6)                       # board_pos
        np.testing.assert_allclose(o[36:39], np.array(hm.pos[0]) - st.eef_pos, atol=1e-6)
        np.testing.assert_allclose(o[40:43], 0.0, atol=1e-6)                             # vec_eef_to_hammer
        np.testing.assert_allclose(o[47:50], st.eef_pos, atol=1e-6)                      # hammer_pos
        nail = np.array(hm.pos[0]) + Rb @ np.array([hm.nail_xy[0], hm.nail_xy[1], d.hm_nail_z0])
        np.testing.assert_allclose(o[50:53], nail, atol=1e-6)                            # nail_pos
        np.testing.assert_allclose(o[43:46], nail - st.eef_pos, atol=1e-6)
        np.testing.assert_allclose(o[57:61], [hm.quat[0][1], hm.quat[0][2], hm.quat[0][3], hm.quat[0][0]], atol=1e-6)   # board_quat (x, y, z, w)
        assert o[61] == 0.0 and o[39] == 0.0
    B.close()


def test_gripper_holds_the_hammer_and_hammer_contacts_are_whitelisted():
    B, d, _ = _batch()
    B.reset()
    G = 10 + 24 + 2   # GEOM_BOX of the oracle's geom numbering: robot capsules, human bodies, table, floor
    for k in range(10):
        obs, r, dn, info = B.step(np.zeros((2, 7)))
        pairs, nc = B.contacts()
        hm = B.get_hammer(0)
        assert hm.gripped == 1 and obs[0, 39] == 1.0
        fingers = {tuple(p) for p in pairs[0][:nc[0]].tolist()}
        assert (8, G + 1) in fingers and (9, G + 1) in fingers                           # both finger bars on the handle
        assert info[0, 3] == 0                                                           # no static collision: the hammer is white-listed (1325-1337)
    assert info[0, 1] in (0, CONST["HRG_COL_ALLOWED"])
    hm, st = B.get_hammer(0), B.get_state(0)
    assert np.linalg.norm(np.array(hm.obs_pos[1]) - np.array(st.eef_pos)) < 0.02          # still in the gripper after a second
    # ... while a robot contact with the board is a static collision ("the board is not white-listed")
    hm.pos[0][:] = [st.eef_pos[0] + 0.3, st.eef_pos[1], st.eef_pos[2] + 0.06]
    B.set_hammer(0, hm)
    _, _, _, info = B.step(np.zeros((2, 7)))
    assert info[0, 1] & CONST["HRG_COL_STATIC"] and info[0, 3] >= 1
    # ... and so is one with the nail head (not white-listed either): the board moved so that the nail sits at a finger bar
    hm, st = B.get_hammer(1), B.get_state(1)
    Rb = _quat2mat(hm.quat[0])
    nail = Rb @ np.array([hm.nail_xy[0], hm.nail_xy[1], d.hm_nail_z0])
    hm.pos[0][:] = (np.array(st.eef_pos) + [0.0, 0.03, -0.02] - nail).tolist()
    hm.pos[1][:] = [0.3, -0.8, 3.0]                                                      # the hammer out of the way
    B.set_hammer(1, hm)
    before = B.step(np.zeros((2, 7)))[3][1, 3] if False else None
    info = B.step(np.zeros((2, 7)))[3]   # (the weld pulls the board back within the step: the contact shows in the step's collision counters, not in its last substep's list)
    assert info[1, 1] & CONST["HRG_COL_STATIC"]
    assert info[1, 3] >= 1
    B.close()


def test_phase_machine_rewards_and_next_nail():
    B, d, clips = _batch(n=1, task_reward=5.0, nail_hammered_in_reward=-0.5, hammer_gripped_reward_bonus=0.25, hm_pad=None) if False else _batch(
        n=1, task_reward=5.0, nail_hammered_in_reward=-0.5, hammer_gripped_reward_bonus=0.25)
    B.reset()
    P = {k: CONST["HRG_HM_" + k] for k in ("APPROACH", "PRESENT", "RETREAT", "COMPLETE")}
    seen, rewards, at = [], [], []
    hammered_at = None
    first_nail = tuple(B.get_hammer(0).nail_xy)
    for k in range(120):
        hm = B.get_hammer(0)
        if hm.task_phase == P["PRESENT"] and hammered_at is None and k >= 25:   # drive the nail in by hand
            hm.nail_q, hm.nail_v = d.hm_nail_range, 0.0
            B.set_hammer(0, hm)
            hammered_at = k
        elif hammered_at is None:                                               # keep gravity's creep (soft friction row) from finishing the job
            hm.nail_q, hm.nail_v = 0.0, 0.0
            B.set_hammer(0, hm)
        obs, r, dn, info = B.step(np.zeros((1, 7)))
        seen.append(B.get_hammer(0).task_phase); rewards.append(float(r[0])); at.append(B.get_state(0).animation_time)
        if info[0, 9] == 1 and seen[-1] == P["APPROACH"]:
            break
    assert hammered_at is not None
    i_present = seen.index(P["PRESENT"])
    assert set(seen[:i_present]) == {P["APPROACH"]}
    # while presenting the animation idles around the middle of the keyframes: it does not run on towards the clip's end
    clip_len = max(at) + 1
    assert max(at[i_present:hammered_at]) < 0.62 * 400
    assert seen[hammered_at] == P["RETREAT"] and rewards[hammered_at] == pytest.approx(-0.5 + 0.25)   # nail_hammered_in_reward + gripped bonus
    assert rewards[hammered_at - 1] == pytest.approx(-1.0 + 0.25)
    assert rewards[-1] == pytest.approx(5.0)                                                          # task_reward once the animation is complete
    hm = B.get_hammer(0)
    assert hm.task_phase == P["APPROACH"] and hm.nail_index == 1 and hm.nail_q == 0.0 and tuple(hm.nail_xy) != first_nail   # _on_goal_reached: next nail, next animation
    assert B.get_state(0).anim_index == 1
    B.close()


def _nail_run(steps, hammer=None, **kw):
    """The board brought in by the human (12 steps), then `steps` policy steps with the nail pulled out: hammer None = far away, "rest" = lying on the nail head
    (handle level, free: a weight of 1 N), a number = dropped onto the head with that downward speed [m/s].  Returns (nail travel, its largest value on the way)."""
    B, d, _ = _batch(n=1, **kw)
    B.reset()
    for k in range(12):
        B.step(np.zeros((1, 7)))
    hm = B.get_hammer(0)
    Rb = _quat2mat(hm.quat[0])
    hm.nail_q = hm.nail_v = 0.0
    top = np.array(hm.pos[0]) + Rb @ np.array([hm.nail_xy[0], hm.nail_xy[1], d.hm_nail_z0 + 0.003])
    hm.vel[1][:] = [0.0] * 6
    if hammer is None:
        hm.pos[1][:] = [0.3, -0.8, 3.0]
    else:
        hm.quat[1][:] = [np.sqrt(0.5), 0, np.sqrt(0.5), 0]
        head = np.array(d.hm_geom_pos[CONST["HRG_HG_HEAD"]][:])
        Rh = _quat2mat(hm.quat[1])
        gap = 0.0 if hammer == "rest" else 0.01
        hm.pos[1][:] = (top + [0, 0, d.hm_geom_half[CONST["HRG_HG_HEAD"]][0] + gap] - Rh @ head).tolist()
        if hammer != "rest":
            hm.vel[1][2] = -float(hammer)
    B.set_hammer(0, hm)
    q0, far = B.get_hammer(0).nail_q, 0.0
    for k in range(steps):
        B.step(np.zeros((1, 7)))
        far = max(far, abs(B.get_hammer(0).nail_q - q0))
    out = B.get_hammer(0).nail_q - q0
    B.close()
    return out, far


def test_noslip_holds_the_nail_under_its_own_weight_and_under_a_resting_hammer():
    """MuJoCo's noslip post-pass (collaborative_hammering_cartesian_env.py:1161, noslip_iterations = 20) on the nail's friction-loss row (nail.xml:7, 10 000 N): the
    nail does not move under its own weight, nor under a hammer lying on it, within 1e-6 m in 10 s.  Without the pass (round 2's model) the soft row lets gravity
    drive the 5 g nail in at about 1 cm/s and a resting hammer many times faster -- the task completed itself."""
    assert _nail_run(100)[1] < 1e-6
    assert _nail_run(100, hammer="rest")[1] < 1e-6
    alone, _ = _nail_run(2, noslip_iterations=0)
    pressed, _ = _nail_run(2, hammer="rest", noslip_iterations=0)
    assert 0.0015 < alone < 0.0025 and pressed > 3 * alone


def test_a_nail_yields_only_to_a_force_above_its_friction_loss():
    """With the pass the nail's acceleration is its friction row's reference (-b v) while the force that takes stays inside +-frictionloss.
    Static: the stand-in hammer lying on the head weighs 1 N -- a nail with a friction loss of 5 N holds it (1e-6 m in 2 s), one with 0.5 N is pushed in.
    Struck: the same hammer hitting the head at 3 m/s (m b v ~ 30 N) drives the 5 N nail in.  At nail.xml's 10 000 N no blow of a 100 g hammer comes near the
    friction loss: the nail stays out; what it does move (< 2 mm, either way) is the residual the Gauss-Seidel pass leaves when its 20 sweeps over eight
    simultaneous contacts end before they have converged -- MuJoCo's pass has the same cap."""
    assert _nail_run(20, hammer="rest", nail_frictionloss=5.0)[1] < 1e-6
    assert _nail_run(10, hammer="rest", nail_frictionloss=0.5)[0] > 0.03
    assert _nail_run(3, hammer=3.0, nail_frictionloss=5.0)[0] > 0.03
    assert _nail_run(30, hammer=3.0)[1] < 2e-3


def test_state_round_trip_and_determinism():
    B, d, clips = _batch(n=2)
    B.reset()
    rng = np.random.RandomState(3)
    acts = rng.uniform(-1, 1, (6, 2, 7)) * 0.3
    for k in range(3):
        B.step(acts[k])
    snap = [(B.get_state(e), B.get_hammer(e)) for e in range(2)]
    ref = [B.step(acts[k]) for k in range(3, 6)]
    for e, (st, hm) in enumerate(snap):
        B.set_state(e, st); B.set_hammer(e, hm)
    again = [B.step(acts[k]) for k in range(3, 6)]
    for a, b in zip(ref, again):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
    B.close()
