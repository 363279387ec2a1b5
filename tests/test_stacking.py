"""CollaborativeStackingCart on the CPU oracle: known-answer tests of the box-box contact generator, physical sanity of stacks of free cubes, and the
task's phase machine / rewards (collaborative_stacking_cartesian_env.py).  The GPU legs (HIP vs oracle) live in tests/test_stacking_gpu.py."""
import ctypes

import numpy as np
import pytest

import human_robot_gym_amd as hrg
from human_robot_gym_amd._cstruct import CONST
from human_robot_gym_amd.mixed import task_clips

H = 0.0225   # half edge of the reference's cubes (object_full_size 0.045)


def _quat(axis, ang):
    ax = np.asarray(axis, float) / np.linalg.norm(axis)
    return np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])


def _boxbox(oracle_lib, pa, qa, pb, qb, half=(H, H, H)):
    out = np.zeros(28)
    a = [np.ascontiguousarray(x, np.float64) for x in (pa, qa, pb, qb, half)]
    n = oracle_lib.hrgo_test_boxbox(*[x.ctypes.data_as(ctypes.c_void_p) for x in a], out.ctypes.data_as(ctypes.c_void_p))
    return out[:7 * n].reshape(n, 7)


def test_box_box_aligned_faces_give_the_four_corners(oracle_lib):
    pen = 4e-4
    c = _boxbox(oracle_lib, [0, 0, 0], [1, 0, 0, 0], [0, 0, 2 * H - pen], [1, 0, 0, 0])
    assert c.shape == (4, 7)
    np.testing.assert_allclose(c[:, 3:6], np.tile([0, 0, 1.0], (4, 1)), atol=1e-12)        # normal from the lower to the upper cube
    np.testing.assert_allclose(c[:, 6], -pen, atol=1e-12)
    np.testing.assert_allclose(sorted(map(tuple, np.round(c[:, :2] / H))), [(-1, -1), (-1, 1), (1, -1), (1, 1)])
    np.testing.assert_allclose(c[:, 2], H - 0.5 * pen, atol=1e-12)                          # midway between the two faces
    # swapping the boxes flips the normal and keeps the points
    c2 = _boxbox(oracle_lib, [0, 0, 2 * H - pen], [1, 0, 0, 0], [0, 0, 0], [1, 0, 0, 0])
    np.testing.assert_allclose(c2[:, 3:6], np.tile([0, 0, -1.0], (4, 1)), atol=1e-12)
    assert {tuple(np.round(p, 9)) for p in c2[:, :3]} == {tuple(np.round(p, 9)) for p in c[:, :3]}


def test_box_box_yawed_and_offset_faces_keep_a_spanning_support(oracle_lib):
    pen = 3e-4
    # 45 deg yaw: the overlap is an octagon with no vertex of either face inside the other -- edge crossings only
    c = _boxbox(oracle_lib, [0, 0, 0], [1, 0, 0, 0], [0, 0, 2 * H - pen], _quat([0, 0, 1], np.pi / 4))
    assert len(c) == 4 and np.allclose(c[:, 6], -pen) and np.allclose(c[:, 5], 1.0)
    assert np.all(np.abs(c[:, :2]) <= H + 1e-12)
    hull_span = c[:, :2].max(0) - c[:, :2].min(0)
    assert np.all(hull_span > 1.5 * H)                                                      # the four points span the overlap, not one corner of it
    assert abs(c[:, 0].mean()) < 1e-9 and abs(c[:, 1].mean()) < 1e-9
    # upper cube shifted by 60 % of an edge along x: the support ends at the lower cube's edge
    c = _boxbox(oracle_lib, [0, 0, 0], [1, 0, 0, 0], [1.2 * H, 0, 2 * H - pen], [1, 0, 0, 0])
    assert len(c) == 4 and np.allclose(c[:, 6], -pen)
    assert np.isclose(c[:, 0].max(), H) and np.isclose(c[:, 0].min(), 0.2 * H)
    # a 10 deg tilt about y: only the low edge of the upper cube penetrates, two contacts on it
    q = _quat([0, 1, 0], np.radians(10))
    z = H + H * (np.cos(np.radians(10)) + np.sin(np.radians(10))) - 2e-4
    c = _boxbox(oracle_lib, [0, 0, 0], [1, 0, 0, 0], [0, 0, z], q)
    assert len(c) == 2 and np.allclose(c[:, 6], -2e-4, atol=1e-9) and np.allclose(np.abs(c[:, 1]), H) and np.allclose(c[:, 5], 1.0)


def test_box_box_edge_edge_and_separated(oracle_lib):
    assert len(_boxbox(oracle_lib, [0, 0, 0], [1, 0, 0, 0], [0, 0, 2 * H + 1e-6], [1, 0, 0, 0])) == 0
    assert len(_boxbox(oracle_lib, [0, 0, 0], [1, 0, 0, 0], [3 * H, 3 * H, 0], _quat([0, 0, 1], 0.3))) == 0
    # crossed edges: lower cube turned 45 deg about x (an edge along x on top), upper cube turned 45 deg about y (an edge along y at the bottom)
    r2 = np.sqrt(2.0) * H
    pen = 5e-4
    c = _boxbox(oracle_lib, [0, 0, 0], _quat([1, 0, 0], np.pi / 4), [0, 0, 2 * r2 - pen], _quat([0, 1, 0], np.pi / 4))
    assert c.shape == (1, 7)
    np.testing.assert_allclose(c[0, 3:6], [0, 0, 1], atol=1e-9)
    np.testing.assert_allclose(c[0, 6], -pen, atol=1e-9)
    np.testing.assert_allclose(c[0, :3], [0, 0, r2 - 0.5 * pen], atol=1e-9)


def _batch(n=2, **kw):
    from oracle.oracle import OracleBatch
    clips = task_clips("CollaborativeStackingCart", 2, min_frames=3000, max_frames=3200)
    d = hrg.build_model_desc(dict(dict(seed=5, horizon=400, shield_type="OFF", done_at_success=False), **kw), n_clips=clips.n_clips, env_id="CollaborativeStackingCart")
    return OracleBatch(d, clips, n), d


def _place(B, e, poses):
    sk = B.get_stack(e)
    for c, (p, q) in poses.items():
        sk.pos[c][:] = list(p); sk.quat[c][:] = list(q)
        sk.vel[c][:] = [0.0] * 6; sk.acc_warmstart[c][:] = [0.0] * 6
        sk.obs_pos[c][:] = list(p)
    B.set_stack(e, sk)


def test_stacks_of_free_cubes_rest_and_an_overhanging_cube_falls():
    B, d = _batch(3)
    B.reset()
    top = d.table_top_z
    x0, y0 = 0.45, 0.35           # away from the robot's cubes' bin and the arm's rest posture
    id4 = [1, 0, 0, 0]
    # env 0: robot cubes stacked straight; env 1: upper cube yawed by 45 deg; env 2: upper cube overhanging by 70 % of an edge
    _place(B, 0, {0: ([x0, y0, top + H], id4), 1: ([x0, y0, top + 3 * H], id4)})
    _place(B, 1, {0: ([x0, y0, top + H], id4), 1: ([x0, y0, top + 3 * H], _quat([0, 0, 1], np.pi / 4))})
    _place(B, 2, {0: ([x0, y0, top + H], id4), 1: ([x0 + 1.4 * H, y0, top + 3 * H], id4)})
    for k in range(12):          # 1.2 s
        B.step(np.zeros((3, 7)))
    for e in (0, 1):
        sk = B.get_stack(e)
        p0, p1 = np.array(sk.pos[0]), np.array(sk.pos[1])
        assert abs(p0[2] - (top + H)) < 1.5e-3 and abs(p1[2] - (top + 3 * H)) < 3e-3, (e, p0, p1)   # soft contacts: sub-millimetre sink per interface
        assert np.linalg.norm(p1[:2] - [x0, y0]) < 2e-3 and np.linalg.norm(np.array(sk.vel[1])) < 5e-3, (e, p1, list(sk.vel[1]))
        pairs, ncon = B.contacts()
        assert ((pairs[e, :ncon[e], 0] == 36) & (pairs[e, :ncon[e], 1] == 37)).sum() == 4              # cube a - cube b: four contacts
    sk = B.get_stack(2)
    assert sk.pos[1][2] < top + 2 * H and abs(sk.pos[0][2] - (top + H)) < 2e-3                      # the overhanging cube has come down beside the other
    B.close()


def test_phase_machine_rewards_and_stack_bookkeeping():
    """The episode scripted through its phases: the human's cubes are released at their keyframes, the robot's cubes are teleported onto the stack
    when it is the robot's turn; success when the animation has run to its end (collaborative_stacking_cartesian_env.py:550-588, 700-778, 825-897)."""
    B, d = _batch(1, second_cube_at_target_reward=-0.5, fourth_cube_at_target_reward=-0.25, object_gripped_reward=0.0, task_reward=2.0)
    B.reset()
    seen, heights, rewards = [], [], []
    sk = B.get_stack(0)
    assert sk.task_phase == CONST["HRG_STK_APPROACH"] and list(sk.weld_active) == [1, 1] and sk.n_stack == 0
    placed = {2: False, 4: False}
    for k in range(400):
        sk = B.get_stack(0)
        ph = sk.task_phase
        if ph in (CONST["HRG_STK_WAIT_FOR_SECOND"], CONST["HRG_STK_WAIT_FOR_FOURTH"]) and sk.has_target and not placed[ph]:
            below = sk.stack_ids[sk.n_stack - 1]
            if np.linalg.norm(np.array(sk.vel[below])) < 0.02:      # the cube below has come to rest: put the robot's cube on it
                cube = 1 if ph == CONST["HRG_STK_WAIT_FOR_SECOND"] else 0
                _place(B, 0, {cube: (np.array(sk.pos[below]) + [0, 0, 2 * H + 2e-4], [1, 0, 0, 0])})
                placed[ph] = True
        obs, r, dn, info = B.step(np.zeros((1, 7)))
        seen.append(B.get_stack(0).task_phase); heights.append(int(info[0, 13])); rewards.append(float(r[0]))
        if dn[0] or info[0, CONST["HRG_INFO_N_GOAL_REACHED"]] > 0:
            break
    order = [p for i, p in enumerate(seen) if i == 0 or p != seen[i - 1]]
    # every phase in order; COMPLETE is reached inside the last step, whose _on_goal_reached starts the next animation (done_at_success=False)
    assert order == [0, 1, 2, 3, 4, 5, 0], order
    assert info[0, CONST["HRG_INFO_N_GOAL_REACHED"]] == 1 and not dn[0]
    assert max(heights) == 4 and heights == sorted(heights)                                                 # max_stack_height grows to 4
    assert rewards[-1] == pytest.approx(2.0)                                                               # task_reward on success
    sk = B.get_stack(0)
    assert list(sk.weld_active) == [1, 1] and sk.n_stack == 0 and sk.obj_index == 1                        # both cubes back in the human's hands, next placements
    assert -0.5 in rewards and -0.25 in rewards and rewards[0] == -1.0                                      # the sub-objective rewards were paid on the way
    B.close()


def test_toppled_stack_ends_the_episode():
    B, d = _batch(1, stack_toppled_reward=-7.0)
    B.reset()
    sk = B.get_stack(0)
    sk.n_stack = 2; sk.stack_ids[0] = 2; sk.stack_ids[1] = 1; sk.task_phase = CONST["HRG_STK_PLACE_THIRD"]; sk.weld_active[0] = 0
    B.set_stack(0, sk)
    top = d.table_top_z
    _place(B, 0, {2: ([0.45, 0.35, top + H], [1, 0, 0, 0]), 1: ([0.45 + 3 * H, 0.35, top + H], [1, 0, 0, 0])})   # the "second" cube lies beside the first
    obs, r, dn, info = B.step(np.zeros((1, 7)))
    assert dn[0] == 1 and r[0] == pytest.approx(-7.0)
    B.close()
