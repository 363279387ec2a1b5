"""Human capsule x manipulation object contacts (human.xml:5: contype / conaffinity 7 -- the human's geoms collide with the object like everything else; the
animated human does not yield).  A human walks past the table and its leg / pelvis capsules sweep the cube along: oracle behaviour, then HIP vs oracle on the GPU.
PARITY UNPINNED (MuJoCo cannot run here); the human is the set of bounding capsules of D1, so WHERE it touches differs from the reference's mesh hulls."""
import numpy as np
import pytest

import human_robot_gym_amd as hrg
from human_robot_gym_amd import animation as A
from human_robot_gym_amd._cstruct import CONST as C

GH0, GBOX = C["HRG_NRCAP"], C["HRG_NRCAP"] + C["HRG_NHB"] + 2
N_STEPS = 64        # (the 800-frame clip lasts 66 policy steps)


def _walking_clip(n, p0, p1):
    """a T-pose human whose pelvis moves on a straight line from p0 to p1 over the clip"""
    anim = {k: np.zeros(n) for k in A._qpos_joint_order()}
    t = np.linspace(0, 1, n)
    for a, k in enumerate(("Pelvis_pos_x", "Pelvis_pos_y", "Pelvis_pos_z")):
        anim[k] = p0[a] + (p1[a] - p0[a]) * t
    anim["Pelvis_quat"] = np.tile(np.array([0.0, 0.0, 0.0, 1.0]), (n, 1))
    clips = A.ClipSet([(anim, None)])
    for c in clips.infos:
        c["position_offset"] = [0.0, 0.0, 0.0]
    return clips


def _scene():
    # animation axes (x, y, z) = world (y, z, x): a standing human (pelvis 1 m up) walks 1 m along world y, through the table and over the cube's place
    clips = _walking_clip(800, (1.2, 1.0, 0.395), (0.2, 1.0, 0.395))
    kw = dict(shield_type="OFF", horizon=400, seed=1, done_at_collision=False)
    return clips, (lambda: hrg.build_model_desc(kw, n_clips=1, env_id="PickPlaceHumanCart"))


def _human_pairs(pairs, n):
    return [(int(p[0]) - GH0) for p in pairs[:n].tolist() if GH0 <= p[0] < GBOX - 2 and p[1] == GBOX]


def test_a_passing_human_pushes_the_cube_along_the_table():
    from oracle.oracle import OracleBatch
    clips, mk = _scene()
    d = mk()
    B = OracleBatch(d, clips, 1)
    B.reset()
    p0 = np.array(B.get_box(0).pos[:])
    first, touched, track = None, 0, []
    for k in range(N_STEPS):
        o, r, dn, i = B.step(np.zeros((1, 7)))
        pairs, n = B.contacts()
        hp = _human_pairs(pairs[0], n[0])
        if hp and first is None:
            first = k
        touched += bool(hp)
        track.append(np.array(B.get_box(0).pos[:]))
        assert i[0, 11] == 0                                      # no solver failure
    track = np.array(track)
    assert first is not None and first > 20, "the human starts clear of the cube and reaches it while walking"
    assert np.abs(track[5:first - 1] - track[5]).max() < 1e-3     # the cube rests (once settled after the reset) until it is touched
    assert touched >= 5
    # pushed along world -y at the human's pace (1 m in 66 steps), over the table top, not sideways
    walked = (N_STEPS - 1 - (first + 2)) * (1.0 / 66.0)
    moved = track[-1] - track[first + 2]
    assert abs(-moved[1] - walked) < 0.03 and abs(moved[0]) < 0.02
    assert np.abs(track[first:, 2] - (d.table_top_z + d.box_half[2])).max() < 0.01
    B.close()


def test_no_human_contacts_for_an_object_the_human_holds():
    """The human is a set of BOUNDING capsules: an object in its hands lies partly inside them (the lifting task's board reaches into the capsules of pelvis and
    thighs) -- no contacts while the weld / the connects hold it, and none in the lifting task."""
    from oracle.oracle import OracleBatch
    from human_robot_gym_amd import mixed
    for env_id in ("CollaborativeLiftingCart", "HumanRobotHandoverCart"):
        clips = mixed.task_clips(env_id, 13)
        d = hrg.build_model_desc(dict(shield_type="SSM", seed=3, **mixed.task_env_kwargs(env_id)), n_clips=clips.n_clips, env_id=env_id)
        B = OracleBatch(d, clips, 4)
        B.reset()
        rng = np.random.RandomState(0)
        for k in range(40):
            B.step(rng.uniform(-1, 1, (4, 7)) * 0.3)
            pairs, n = B.contacts()
            for e in range(4):
                if B.get_box(e).weld_active or env_id == "CollaborativeLiftingCart":
                    assert not _human_pairs(pairs[e], n[e])
        B.close()


@pytest.mark.gpu
def test_hip_human_object_contacts_match_oracle():
    import torch
    from helpers import RTOL, assert_state_close
    from oracle.oracle import OracleBatch
    from human_robot_gym_amd._lib import HipBatch
    clips, mk = _scene()
    n = 4
    O, G = OracleBatch(mk(), clips, n), HipBatch(mk(), clips, n)
    O.reset(); G.reset()
    for e in range(n):          # the same scene four times, the cube shifted a little: different capsules reach it at different times
        bx = O.get_box(e)
        bx.pos[0] += 0.01 * e; bx.pos[1] += 0.015 * e
        O.set_box(e, bx); G.set_box(e, bx)
    touched = 0
    for k in range(N_STEPS):
        a = np.zeros((n, 7))
        o_o, r_o, d_o, i_o = O.step(a)
        o_g, r_g, d_g, i_g = G.step(torch.from_numpy(a).cuda())
        torch.cuda.synchronize()
        po, no = O.contacts()
        pg, ng = G.contacts()
        post_b = [O.get_box(e) for e in range(n)]
        ok = np.array([max(abs(v) for v in post_b[e].vel[:]) < 5.0 for e in range(n)])   # (a cube the capsule caught deep is thrown: chaotic, as everywhere)
        np.testing.assert_array_equal(ng[ok], no[ok], err_msg=f"step {k}")
        np.testing.assert_array_equal(pg[ok], po[ok], err_msg=f"step {k}")
        np.testing.assert_array_equal(i_g.cpu().numpy()[ok], i_o[ok], err_msg=f"step {k}")
        np.testing.assert_allclose(o_g.cpu().numpy()[ok], o_o[ok], rtol=RTOL, atol=1e-6, err_msg=f"step {k}")
        for e in range(n):
            touched += bool(_human_pairs(po[e], no[e]))
            if ok[e]:
                assert_state_close(post_b[e], G.get_box(e), f"step {k} env {e} box")
            G.set_state(e, O.get_state(e)); G.set_box(e, post_b[e])
    assert touched >= 6 and ok.mean() >= 0.75
    O.close(); G.close()
