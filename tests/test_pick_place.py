"""PickPlaceHumanCart on the CPU oracle: behaviour of the restated task (pick_place_human_cartesian_env.py) and of the cube's
contact model.  PARITY UNPINNED: the reference holds no fixtures for this path; these tests pin behaviour
the reference documents (reward terms, success rule, target / placement cycling, observables) and physical sanity."""
import ctypes

import numpy as np
import pytest

import human_robot_gym_amd as hrg
from human_robot_gym_amd._cstruct import CONST
from oracle.oracle import OracleBatch, load
from pp_scenarios import PP, between_fingers, deliver, grasp_and_carry, put_box, tumble

KW = dict(shield_type="OFF", horizon=200, seed=3)


def _batch(n, kw=KW):
    clips = hrg.synthetic_clips(2, seed=0, min_frames=200, max_frames=300)
    d = hrg.build_model_desc(kw, n_clips=clips.n_clips, **PP)
    return OracleBatch(d, clips, n), d


def test_desc_follows_the_reference_defaults():
    d = hrg.build_model_desc(None, **PP)
    assert d.task == CONST["HRG_TASK_PICK_PLACE"] and d.horizon == 1000 and d.n_targets == 30 and d.n_obj_placements == 30
    np.testing.assert_allclose(list(d.init_qpos), [0, 0, -np.pi / 2, 0, -np.pi / 2, np.pi / 4])
    np.testing.assert_allclose(list(d.obj_bin), [0.7 * 0.35, 0.7 * 0.6, 0.95 * 0.25, 0.95 * 0.45])      # 843-858
    np.testing.assert_allclose(list(d.tgt_bin), [0.7 * 0.35, 0.7 * 0.6, 0.95 * -0.45, 0.95 * -0.25])    # 860-875
    assert list(d.box_half) == [0.02] * 3 and list(d.box_inertia) == [d.box_inertia_mean] * 3 and abs(d.box_mass - 0.064) < 1e-12 and d.object_gripped_reward == -0.25 and d.obstacle_margin == 0.0
    assert abs(d.obj_z - 0.82) < 1e-12 and abs(d.tgt_z - 0.84) < 1e-12
    assert hrg.build_model_desc(None).task == CONST["HRG_TASK_REACH"]


def test_reset_places_object_and_target_in_their_bins_and_observes_them():
    B, d = _batch(16)
    obs = B.reset()
    for e in range(16):
        bx = B.get_box(e)
        assert d.obj_bin[0] <= bx.pos[0] <= d.obj_bin[1] and d.obj_bin[2] <= bx.pos[1] <= d.obj_bin[3] and bx.pos[2] == d.obj_z
        assert d.tgt_bin[0] <= bx.target[0] <= d.tgt_bin[1] and d.tgt_bin[2] <= bx.target[1] <= d.tgt_bin[3] and bx.target[2] == d.tgt_z
        assert list(bx.quat) == [1, 0, 0, 0] and not any(bx.vel)
        s = B.get_state(e)
        np.testing.assert_allclose(obs[e, 47:50], list(bx.pos), rtol=1e-6)
        np.testing.assert_allclose(obs[e, 50:53], list(bx.target), rtol=1e-6)
        np.testing.assert_allclose(obs[e, 40:43], np.array(bx.pos) - np.array(s.eef_pos), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(obs[e, 43:46], np.array(bx.target) - np.array(s.eef_pos), rtol=1e-5, atol=1e-7)
        assert obs[e, 39] == 0 and abs(obs[e, 46] - 1.0) < 1e-6          # not gripped, gripper fully open (qpos_range[1])
        assert list(obs[e, 12:16]) == [0, 0, 0, 1] and not obs[e, 16:18].any() and not obs[e, 33:39].any()   # object_quat (x, y, z, w); no joint-space columns
    assert len({round(B.get_box(e).pos[0], 9) for e in range(16)}) == 16   # per-env streams
    B.close()


def test_cube_settles_on_the_table_and_stays():
    """The sampler drops the cube centre at the table surface (reference_pos z 0.8 + half edge): the contact solver has to
    push it out and bring it to rest on its face."""
    B, d = _batch(4)
    B.reset()
    for _ in range(8):
        _, r, _, info = B.step(np.zeros((4, 7)))
    for e in range(4):
        bx = B.get_box(e)
        assert abs(bx.pos[2] - (d.table_top_z + d.box_half[2])) < 3e-4 and max(abs(v) for v in bx.vel) < 1e-3
        assert abs(bx.quat[0]) > 1 - 1e-6
    pairs, ncon = B.contacts()
    assert (ncon == 4).all() and (pairs[:, :4, 0] == 34).all() and (pairs[:, :4, 1] == 36).all()   # table (34) - cube (36) corners
    assert (r == -1).all() and not info[:, 0].any()                        # sparse reward, no robot collision
    B.close()


def test_tumbling_cube_comes_to_rest_on_a_face():
    B, d = _batch(8)
    B.reset()
    rng = np.random.RandomState(0)
    for k in range(40):
        a = tumble(k, [B], rng, 8)
        _, _, _, info = B.step(a)
        assert not info[:, 11].any()
    for e in range(8):
        bx = B.get_box(e)
        assert abs(bx.pos[2] - (d.table_top_z + d.box_half[2])) < 1e-3, bx.pos[2]
        assert max(abs(v) for v in bx.vel) < 0.05
        assert abs(np.linalg.norm(list(bx.quat)) - 1) < 1e-12
    B.close()


def test_grasp_carry_release():
    B, d = _batch(2)
    B.reset()
    rng = np.random.RandomState(0)
    grip, rew, dz = [], [], []
    for k in range(40):
        a = grasp_and_carry(k, [B], rng, 2, d)
        obs, r, _, info = B.step(a)
        assert not info[:, 11].any()
        grip.append(obs[:, 39].copy()); rew.append(r.copy())
        dz.append([np.linalg.norm(obs[e, 40:43]) for e in range(2)])
        assert not (info[:, 1] & (4 | 8 | 16)).any()          # finger - cube contacts are ALLOWED, not illegal collisions
    grip, rew, dz = np.array(grip), np.array(rew), np.array(dz)
    assert (grip[3:30] == 1).all() and (rew[3:30] == -0.25).all()          # object_gripped_reward while held
    assert (dz[5:30] < 0.045).all()                                         # carried along with the end effector
    assert (grip[33:] == 0).all() and (rew[33:] == -1).all()                # released
    for e in range(2):
        assert abs(B.get_box(e).pos[2] - (d.table_top_z + d.box_half[2])) < 1e-3   # and back on the table
    B.close()


def test_success_advances_target_and_placement_and_pays_task_reward():
    B, d = _batch(4, dict(KW, reward_shaping=False))
    B.reset()
    rng = np.random.RandomState(1)
    first = [(list(B.get_box(e).target), list(B.get_box(e).pos)) for e in range(4)]
    hits = 0
    for k in range(12):
        a = deliver(k, [B], rng, 4)
        pre = [B.get_box(e) for e in range(4)]
        obs, r, done, info = B.step(a)
        if k in (3, 9):
            hits += 1
            assert (r == 1.0).all() and not done.any()                       # task_reward, done_at_success False in the yaml
            assert (info[:, 9] == hits).all()
            for e in range(4):
                bx = B.get_box(e)
                assert bx.tgt_index == hits and bx.obj_index == hits
                assert list(bx.target) != list(pre[e].target)
                assert d.obj_bin[0] <= bx.pos[0] <= d.obj_bin[1] and d.obj_bin[2] <= bx.pos[1] <= d.obj_bin[3] and bx.pos[2] == d.obj_z
                np.testing.assert_allclose(obs[e, 50:53], list(pre[e].target), rtol=1e-6)   # the observation still shows the reached target
        else:
            assert (r == -1.0).all()
    assert first[0][0] != list(B.get_box(0).target)
    B.close()


def test_dense_reward_terms():
    B, d = _batch(3, dict(KW, reward_shaping=True))
    B.reset()
    obs, r, _, _ = B.step(np.zeros((3, 7)))
    for e in range(3):
        e2o, o2t = np.linalg.norm(obs[e, 40:43]), np.linalg.norm(obs[e, 50:53] - obs[e, 47:50])
        assert abs(r[e] - (-1 + 1 - (0.2 * e2o + o2t) * 0.1)) < 1e-5        # human_env.py:650-651 + _dense_reward 502-526
    B.close()


def test_seg_box_known_answers():
    lib = load()
    out = (ctypes.c_double * 7)()
    def sb(p1, p2, c=(0, 0, 0), q=(1, 0, 0, 0), hb=0.5):
        hb3 = np.broadcast_to(np.asarray(hb, float), (3,))
        lib.hrgo_test_segbox((ctypes.c_double * 3)(*p1), (ctypes.c_double * 3)(*p2), (ctypes.c_double * 3)(*c), (ctypes.c_double * 4)(*q), (ctypes.c_double * 3)(*hb3), out)
        return np.array(out[:])
    o = sb((2, 0, 0), (3, 0, 0))                      # pointing away: closest at the first end point, face +x
    assert abs(o[0] - 1.5 ** 2) < 1e-14 and np.allclose(o[1:4], (2, 0, 0)) and np.allclose(o[4:7], (0.5, 0, 0))
    o = sb((-3, 2, 0), (3, 2, 0))                     # parallel above the +y face: distance 1.5, closest point over the face
    assert abs(o[0] - 1.5 ** 2) < 1e-14 and abs(o[2] - 2) < 1e-14 and abs(o[5] - 0.5) < 1e-14 and abs(o[1]) <= 0.5 + 1e-12
    o = sb((-2, 2, 0), (2, -2, 0))                    # through the centre
    assert o[0] == 0
    o = sb((0, 2, 2), (2, 0, 2))                      # diagonal past the +x+y+z corner region: closest to the edge x=y=.5... at the corner (0.5,0.5,0.5)
    assert np.allclose(o[4:7], (0.5, 0.5, 0.5)) and np.allclose(o[1:4], (1, 1, 2)) and abs(o[0] - (0.25 + 0.25 + 2.25)) < 1e-12
    rng = np.random.RandomState(0)                    # against dense sampling of the segment, rotated cubes
    for _ in range(200):
        p1, p2, c = rng.uniform(-2, 2, 3), rng.uniform(-2, 2, 3), rng.uniform(-0.5, 0.5, 3)
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        hb = rng.uniform(0.05, 0.6, 3) if _ % 2 else np.full(3, 0.4)          # boxes with three different half extents, and cubes
        o = sb(p1, p2, c, q, hb)
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        t = np.linspace(0, 1, 4001)[:, None]
        P = (p1 + t * (p2 - p1) - c) @ R
        e = P - np.clip(P, -hb, hb)
        assert o[0] <= (e ** 2).sum(1).min() + 1e-12 and o[0] >= (e ** 2).sum(1).min() - 1e-5


def test_vec_env_serves_the_pp_sac_observation_layout():
    """`make_vec_env("PickPlaceHumanCart")` surface: default obs_keys = the observables the pick-place policies are trained on
    (object_gripped, vec_eef_to_object, vec_eef_to_target, gripper_aperture, 3 human distances = 11 values)."""
    from helpers import OracleBackend
    from human_robot_gym_amd.vec_env import HipVecEnv, PICK_PLACE_OBS_KEYS
    clips = hrg.synthetic_clips(2, seed=0, min_frames=200, max_frames=300)
    kw = dict(shield_type="SSM", horizon=6, seed=2)
    desc = hrg.build_model_desc(kw, n_clips=clips.n_clips, **PP)
    back = OracleBackend(desc, clips, 3)
    env = HipVecEnv(3, env_id="PickPlaceHumanCart", env_kwargs=kw, clips=clips, backend=back)
    assert env.obs_keys == PICK_PLACE_OBS_KEYS and env.observation_space.shape == (11,)
    obs = env.reset()
    full = back.B.obs
    np.testing.assert_array_equal(obs[:, 0], full[:, 39])
    np.testing.assert_array_equal(obs[:, 1:4], full[:, 40:43])
    np.testing.assert_array_equal(obs[:, 4:7], full[:, 43:46])
    np.testing.assert_array_equal(obs[:, 7], full[:, 46])
    np.testing.assert_array_equal(obs[:, 8:11], full[:, [11, 3, 7]])          # dist to head, left hand, right hand
    for k in range(6):
        obs, rew, done, infos = env.step(np.zeros((3, 7)))
    assert done.all() and infos[0]["TimeLimit.truncated"] and infos[0]["terminal_observation"].shape == (11,)
    env2 = HipVecEnv(2, env_id="PickPlaceHumanCart", env_kwargs=kw, clips=clips, obs_keys=["object_pos", "target_pos", "robot0_eef_pos"],
                     backend=OracleBackend(hrg.build_model_desc(kw, n_clips=clips.n_clips, **PP), clips, 2))
    assert env2.reset().shape == (2, 9)


def test_pick_place_variants_close_and_pointing():
    """PickPlaceCloseHumanCart = the same task on 60 Hz clips; PickPlacePointingHumanCart takes its target from the human: the
    elbow -> hand ray of the pointing arm extended to the table (pick_place_pointing_human_cartesian_env.py:336-360)."""
    clips = hrg.synthetic_clips(2, seed=0, min_frames=200, max_frames=300)
    for c in range(2):
        clips.infos[c]["pointing_hand"] = "left" if c else "right"
    dc = hrg.build_model_desc(None, n_clips=2, env_id="PickPlaceCloseHumanCart")
    assert dc.task == CONST["HRG_TASK_PICK_PLACE"] and abs(dc.anim_step_length - 250 / 60) < 1e-12
    kw = dict(shield_type="OFF", seed=5, human_rand=[0.2, 0.2, 0.5])
    d = hrg.build_model_desc(kw, n_clips=2, env_id="PickPlacePointingHumanCart")
    assert d.task == CONST["HRG_TASK_POINTING"] and d.horizon == 500
    np.testing.assert_allclose(list(d.obj_bin), [0.7 * 0.35, 0.7 * 0.75, -0.95 * 0.15, 0.95 * 0.15])
    t = clips.table()
    assert [t.clip_pointing_hand[0], t.clip_pointing_hand[1]] == [0, 1]
    B = OracleBatch(d, clips, 6)
    B.reset()
    seen = set()
    for k in range(4):
        obs, r, dn, info = B.step(np.zeros((6, 7)))
        for e in range(6):
            s = B.get_state(e)
            best = None
            for left in (0, 1):
                hand = np.array(s.human_site[d.site_lhand if left else d.site_rhand])
                elbow = np.array(s.human_site[d.site_lelbow if left else d.site_relbow])
                dirv = hand - elbow
                tgt = hand - (hand[2] - d.table_top_z) / dirv[2] * dirv
                if np.allclose(obs[e, 50:53], tgt, rtol=1e-5, atol=1e-6):
                    best = left
            assert best is not None and abs(obs[e, 52] - d.table_top_z) < 1e-6
            seen.add(best)
            np.testing.assert_allclose(obs[e, 43:46], obs[e, 50:53] - np.array(s.eef_pos), rtol=1e-5, atol=1e-6)
    assert seen == {0, 1}                       # both pointing hands occur over the envs' clips
    B.close()




def _brick_batch(n, size=(0.04, 0.07, 0.03), control_freq=250):
    clips = hrg.synthetic_clips(2, seed=0, min_frames=200, max_frames=300)
    d = hrg.build_model_desc(dict(KW, object_full_size=list(size), control_freq=control_freq), n_clips=clips.n_clips, **PP)
    return OracleBatch(d, clips, n), d


def _rot(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def test_box_object_with_three_different_edges():
    """object_full_size need not be a cube (pick_place_human_cartesian_env.py:262): principal inertia per axis, and in free flight the
    angular momentum R diag(I) R' w stays constant while w itself wanders (the gyroscopic term of the free joint)."""
    B, d = _brick_batch(2)                     # control_freq 250: one 4 ms substep per step, so every substep can be inspected
    m, h = d.box_mass, np.array(list(d.box_half))
    assert np.allclose(h, [0.02, 0.035, 0.015]) and abs(m - 1000 * 0.04 * 0.07 * 0.03) < 1e-12
    assert np.allclose(list(d.box_inertia), [m * (h[1] ** 2 + h[2] ** 2) / 3, m * (h[0] ** 2 + h[2] ** 2) / 3, m * (h[0] ** 2 + h[1] ** 2) / 3])
    assert abs(d.box_inertia_mean - np.mean(list(d.box_inertia))) < 1e-18 and abs(d.box_invweight_rot - np.mean(1 / np.array(list(d.box_inertia)))) < 1e-9
    B.reset()
    I = np.array(list(d.box_inertia))
    put_box([B], 0, pos=[0.5, 0.0, 1.6], quat=[1, 0, 0, 0], vel=[0, 0, 0, 6.0, 0.5, 4.0])     # spin about no principal axis, 0.8 m above the table
    Ls, ws = [], []
    for k in range(60):                        # 0.24 s of free fall (0.28 m): no contact yet
        bx = B.get_box(0)
        R, w = _rot(list(bx.quat)), np.array(list(bx.vel)[3:])
        Ls.append(R @ (I * (R.T @ w))); ws.append(w)
        B.step(np.zeros((2, 7)))
    Ls, ws = np.array(Ls), np.array(ws)
    assert np.abs(Ls - Ls[0]).max() < 0.02 * np.linalg.norm(Ls[0])          # first-order integrator: conserved to a few per cent over 60 substeps
    assert np.abs(ws - ws[0]).max() > 0.15 * np.linalg.norm(ws[0])          # ... while the angular velocity moves a lot (it would be constant for a cube)
    assert abs(B.get_box(0).vel[2] - (-9.81 * 0.24)) < 1e-9
    B.close()
    B, d = _brick_batch(4, control_freq=10)
    B.reset()
    rng = np.random.RandomState(1)
    for k in range(50):
        _, _, _, info = B.step(tumble(k, [B], rng, 4))
        assert not info[:, 11].any()
    for e in range(4):                         # at rest on one of its faces: centre height = one of the half extents, that axis vertical
        bx = B.get_box(e)
        R = _rot(list(bx.quat))
        ax = int(np.argmax(np.abs(R[2])))
        assert abs(abs(R[2, ax]) - 1) < 1e-3 and abs(bx.pos[2] - (d.table_top_z + h[ax])) < 1e-3 and max(abs(v) for v in bx.vel) < 0.05
    B.close()


def test_scripted_expert_picks_the_cube_up_and_delivers_it():
    """The task end to end through the Cartesian front-end (IKPositionDeltaWrapper actions [dx, dy, dz, gripper]): hover over the cube, descend,
    close, carry it to the target (experts/pick_place_human_cart_expert.py plays this role in the reference).  With the human standing out of
    the way every env delivers its cube: the task reward is paid and the next object / target are drawn."""
    clips = hrg.static_clip(900, pelvis=(0.0, 1.0, 2.5))                          # 2.5 m in front of the robot
    d = hrg.build_model_desc(dict(seed=4, horizon=300, shield_type="SSM"), n_clips=clips.n_clips, ik_position_delta=dict(action_limit=0.15), **PP)
    n = 6
    B = OracleBatch(d, clips, n)
    obs = B.reset()
    wins, grip_steps, paid = np.zeros(n, int), 0, 0
    for k in range(250):
        v_obj, v_tgt, gr = obs[:, 40:43].astype(float), obs[:, 43:46].astype(float), obs[:, 39] != 0
        a = np.zeros((n, 7))
        for e in range(n):
            if not gr[e]:
                over = np.linalg.norm(v_obj[e, :2]) <= 0.012
                tgt = np.array([v_obj[e, 0], v_obj[e, 1], v_obj[e, 2] + (0.0 if over else 0.08)])
                g = 1.0 if over and abs(v_obj[e, 2]) < 0.02 else -1.0            # the fingertips reach the table 1.6 cm before the grip site reaches the cube's centre
            else:
                far = np.linalg.norm(v_tgt[e, :2]) > 0.03
                tgt, g = np.array([v_tgt[e, 0], v_tgt[e, 1], max(v_tgt[e, 2] + 0.06, 0.0) if far else v_tgt[e, 2] + 0.03]), 1.0
            a[e, :3], a[e, 3] = np.clip(tgt, -0.05, 0.05), g
        obs, r, dn, info = B.step(a)
        assert not info[:, 11].any() and not (info[:, 1] & (4 | 8 | 16)).any()    # no crash, no illegal collision on the way
        paid += int((r > 0).sum())
        wins, grip_steps = np.maximum(wins, info[:, 9]), grip_steps + int(gr.sum())
    assert (wins >= 1).all() and paid >= n and grip_steps > 400, (wins, paid, grip_steps)
    B.close()


def test_reach_human_with_its_small_box():
    """ReachHuman's free smallBox (reach_human_env.py:573-579) on the cube path of the oracle: the task is unchanged (the box is not observed and
    not whitelisted) -- identical observations and rewards to the lean model while nothing touches the box -- and an arm that hits it collects a STATIC collision."""
    from oracle.oracle import OracleBatch
    clips = hrg.synthetic_clips(3, seed=0, min_frames=300, max_frames=600)
    kw = dict(shield_type="OFF", horizon=40, reward_shaping=True, seed=4)
    A = OracleBatch(hrg.build_model_desc(kw, n_clips=3), clips, 6)
    d = hrg.build_model_desc(kw, n_clips=3, reach_box=True)
    assert d.task == CONST["HRG_TASK_REACH_BOX"] and list(d.box_half) == [0.025] * 3 and d.obj_bin[0] == -d.obj_bin[1]
    B = OracleBatch(d, clips, 6)
    np.testing.assert_array_equal(A.reset(), B.reset())
    rng = np.random.RandomState(0)
    for k in range(12):
        a = rng.uniform(-1, 1, (6, 7))
        xa, xb = A.step(a), B.step(a)
        for u, v in zip(xa, xb):
            np.testing.assert_array_equal(u, v)
        z = np.array([B.get_box(e).pos[2] for e in range(6)])
    assert np.all(np.abs(z - (d.table_top_z + 0.025)) < 2e-3)              # the box has popped out of the table top and rests on it
    # put the box where the forearm sweeps: a robot - box contact is a static collision (the box is not whitelisted)
    for e in range(6):
        s, bx = B.get_state(e), B.get_box(e)
        bx.pos[:] = [s.eef_pos[0], s.eef_pos[1], s.eef_pos[2] - 0.02]
        bx.vel[:] = [0.0] * 6
        B.set_box(e, bx)
    before = B.info[:, CONST["HRG_INFO_N_COLLISIONS_STATIC"]].copy()
    flags = np.zeros(6, np.int32)
    for k in range(3):
        o, r, dn, info = B.step(np.zeros((6, 7)))
        flags |= info[:, CONST["HRG_INFO_COLLISION_TYPE"]]          # (edge triggered: the flag is raised in the step a contact is new)
    assert np.all(info[:, CONST["HRG_INFO_N_COLLISIONS_STATIC"]] > before) and np.all(flags & CONST["HRG_COL_STATIC"])
    A.close(); B.close()
