"""Convex hulls of the arm links as collision geometry (build_model_desc(robot_geometry="hull"); robot.xml:29-55: mesh geoms, which MuJoCo convexifies at compile
time).  Known answers for the pieces on the CPU oracle -- the compiled vertex tables, the support mapping against brute force, the GJK distance against closed forms
(a cube) and against its own optimality certificate (the real hulls), the lowest point over a plane at known poses -- the contact lists of the two geometries side
by side, and HIP vs oracle parity on the GPU.  PARITY UNPINNED like everything else here: MuJoCo's own hulls (Qhull) and narrowphase (MPR) cannot run."""
import ctypes

import numpy as np
import pytest

import human_robot_gym_amd as hrg
from human_robot_gym_amd._cstruct import CONST
from human_robot_gym_amd.model import load_robot_hulls

NH = CONST["HRG_NHULL"]


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _rot(rng):
    q = rng.randn(4); q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _desc_with_hull(verts):
    """a desc whose seven hulls are all `verts` (known-answer shapes)"""
    d = hrg.build_model_desc(None, robot_geometry="hull")
    V = np.ascontiguousarray(np.tile(np.asarray(verts, float), (NH, 1)))
    d.hull_off[:] = [len(verts) * h for h in range(NH + 1)]
    d.hull_verts = V.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    d._hull_keep = V
    return d


def test_hull_tables_are_the_convex_hulls_of_the_link_meshes():
    V, off = load_robot_hulls()
    assert off.tolist() == [0, 128, 439, 924, 1235, 3446, 4199, 4321] and V.shape == (4321, 3)     # link0 .. link6 (arm_4_link: 2211 hull vertices of 9236 mesh vertices)
    d = hrg.build_model_desc(None, robot_geometry="hull")
    from scipy.spatial import ConvexHull
    for h in range(NH):
        H = V[off[h]:off[h + 1]]
        assert len(ConvexHull(H).vertices) == len(H)                                               # every stored vertex is a vertex of the hull
        p1, p2, r = np.array(d.rcap_p1[h][:]), np.array(d.rcap_p2[h][:]), d.rcap_r[h]              # ... and the link's bounding capsule bounds it (the broadphase)
        ab = p2 - p1
        u = np.clip((H - p1) @ ab / (ab @ ab), 0, 1) if ab @ ab > 0 else np.zeros(len(H))           # links 0 and 6 are spheres
        assert np.linalg.norm(H - (p1 + u[:, None] * ab), axis=1).max() <= r + 1e-9
    assert hrg.build_model_desc(None).robot_hulls == 0                                             # the capsule model stays the default


def test_support_mapping_against_brute_force(oracle_lib):
    V, off = load_robot_hulls()
    d = hrg.build_model_desc(None, robot_geometry="hull")
    rng = np.random.RandomState(0)
    out = np.zeros(3)
    for trial in range(200):
        h = trial % NH
        R, p, dirn = _rot(rng), rng.uniform(-1, 1, 3), rng.randn(3)
        i = oracle_lib.hrgo_test_hull_support(ctypes.byref(d), h, _p(np.ascontiguousarray(R)), _p(p), _p(dirn), _p(out))
        W = V[off[h]:off[h + 1]] @ R.T + p
        assert i == int(np.argmax(W @ dirn)) and np.allclose(out, W[i], atol=1e-14)


def test_gjk_distance_to_a_cube_known_answers(oracle_lib):
    cube = [[sx * 0.5, sy * 0.5, sz * 0.5] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]
    d = _desc_with_hull(cube)
    I, z = np.ascontiguousarray(np.eye(3)), np.zeros(3)
    out = np.zeros(7)
    run = lambda s1, s2, R=I, p=z: (oracle_lib.hrgo_test_hull_segment(ctypes.byref(d), 0, _p(np.ascontiguousarray(R)), _p(np.asarray(p, float)), _p(np.asarray(s1, float)),
                                                                      _p(np.asarray(s2, float)), _p(out)), out.copy())[1]
    r = run([2, 0, 0], [3, 0, 0]);            assert r[0] == pytest.approx(1.5) and np.allclose(r[1:4], [0.5, 0, 0]) and np.allclose(r[4:7], [2, 0, 0])       # end point - face
    r = run([2, -3, 0.2], [2, 3, 0.2]);       assert r[0] == pytest.approx(1.5) and r[1] == pytest.approx(0.5) and r[4] == pytest.approx(2.0)                  # segment parallel to a face
    r = run([1, 1, -3], [1, 1, 3]);           assert r[0] == pytest.approx(np.sqrt(0.5)) and np.allclose(r[1:3], [0.5, 0.5]) and np.allclose(r[4:6], [1, 1])  # segment - edge (parallel)
    r = run([1, 1, 1], [2, 2, 2]);            assert r[0] == pytest.approx(np.sqrt(0.75)) and np.allclose(r[1:4], [0.5, 0.5, 0.5])                            # end point - vertex
    r = run([1.5, -1, 1], [1.5, 1, -1]);      assert r[0] == pytest.approx(1.0) and np.allclose(r[1:4], [0.5, 0, 0], atol=1e-9) and np.allclose(r[4:7], [1.5, 0, 0], atol=1e-9)  # interior of the segment - face
    r = run([0.2, 0.1, -2], [0.2, 0.1, 2]);   assert r[0] == 0.0                                                                                               # the segment pierces the cube
    # a posed cube: turned 45 deg about z and moved, an edge faces the segment
    c, s = np.cos(np.pi / 4), np.sin(np.pi / 4)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    r = run([3, 1, -1], [3, 1, 1], R, [1, 1, 0])
    assert r[0] == pytest.approx(2 - np.sqrt(0.5)) and np.allclose(r[1:3], [1 + np.sqrt(0.5), 1])


def test_gjk_distance_to_the_real_hulls_carries_its_own_certificate(oracle_lib):
    """The distance GJK returns is attained by its witness points (an upper bound) and the direction between them separates the two sets by exactly that much -- no
    hull vertex reaches further along it than the witness, no point of the segment comes closer (a lower bound): the two bounds meet, so the distance is THE distance."""
    V, off = load_robot_hulls()
    d = hrg.build_model_desc(None, robot_geometry="hull")
    rng = np.random.RandomState(1)
    out = np.zeros(7)
    n_sep = 0
    for trial in range(280):
        h = trial % NH
        R, p = _rot(rng), rng.uniform(-0.2, 0.2, 3)
        c = p + rng.randn(3) * 0.25
        s1, s2 = c + rng.randn(3) * 0.15, c - rng.randn(3) * 0.15
        oracle_lib.hrgo_test_hull_segment(ctypes.byref(d), h, _p(np.ascontiguousarray(R)), _p(p), _p(s1), _p(s2), _p(out))
        W = V[off[h]:off[h + 1]] @ R.T + p
        dist, wa, wb = out[0], out[1:4], out[4:7]
        if dist == 0.0:
            continue
        n_sep += 1
        n = (wb - wa) / dist
        assert np.linalg.norm(wb - wa) == pytest.approx(dist, rel=1e-12)
        t = (wb - s1) @ (s2 - s1) / ((s2 - s1) @ (s2 - s1))
        assert -1e-9 <= t <= 1 + 1e-9 and np.linalg.norm(s1 + t * (s2 - s1) - wb) < 1e-9          # wb lies on the segment
        assert (W @ n).max() <= wa @ n + 1e-9 * (1 + dist)                                          # the hull ends at wa along n ...
        assert min(s1 @ n, s2 @ n) >= wb @ n - 1e-9 * (1 + dist)                                    # ... and the segment begins at wb
    assert n_sep > 200


def test_lowest_point_of_link4_over_a_plane_at_known_poses(oracle_lib):
    V, off = load_robot_hulls()
    d = hrg.build_model_desc(None, robot_geometry="hull")
    H = V[off[4]:off[5]]                                                                             # link4: the largest mesh (arm_4_link.stl)
    out = np.zeros(3)
    rng = np.random.RandomState(2)
    for R in [np.eye(3), np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0.0]]), np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0.0]])] + [_rot(rng) for _ in range(20)]:
        p = np.array([0.3, -0.2, 1.0])
        oracle_lib.hrgo_test_hull_lowest(ctypes.byref(d), 4, _p(np.ascontiguousarray(R)), _p(p), _p(out))
        W = H @ R.T + p
        low = W[W[:, 2] <= W[:, 2].min() + 1e-6]
        assert out[2] == pytest.approx(W[:, 2].min(), abs=1e-14) and np.allclose(out[:2], low[:, :2].mean(0), atol=1e-12)
    # a cube lying on a face: the contact is the middle of that face, not one of its corners; tipped onto an edge: the middle of the edge
    cube = [[sx * 0.5, sy * 0.5, sz * 0.5] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]
    dc = _desc_with_hull(cube)
    oracle_lib.hrgo_test_hull_lowest(ctypes.byref(dc), 0, _p(np.ascontiguousarray(np.eye(3))), _p(np.array([2.0, 3.0, 1.0])), _p(out))
    assert np.allclose(out, [2.0, 3.0, 0.5], atol=1e-14)
    c, s = np.cos(np.pi / 4), np.sin(np.pi / 4)
    oracle_lib.hrgo_test_hull_lowest(ctypes.byref(dc), 0, _p(np.ascontiguousarray(np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]))), _p(np.array([2.0, 3.0, 1.0])), _p(out))
    assert np.allclose(out, [2.0, 3.0, 1.0 - np.sqrt(0.5)], atol=1e-12)


def _scene(geometry, n=8, steps=24):
    """the contact scenario of tests/test_parity_gpu.py: a T-pose human whose hand is 0.3 m from the upright arm, the shoulder driven into it"""
    from oracle.oracle import OracleBatch
    clips = hrg.static_clip(600, pelvis=(-0.8, 1.0, 0.3))
    for c in clips.infos:
        c["position_offset"] = [0.0, 0.0, 0.0]
    kw = dict(shield_type="OFF", horizon=60, done_at_collision=False, collision_reward=-10)
    B = OracleBatch(hrg.build_model_desc(kw, n_clips=clips.n_clips, robot_geometry=geometry), clips, n)
    B.reset()
    rng = np.random.RandomState(5)
    log = []
    for k in range(steps):
        a = rng.uniform(-1, 1, (n, 7))
        a[:, 1] = np.where(np.arange(n) % 2 == 0, 1.0, -1.0)
        a[:, [0, 2, 3, 4, 5]] *= 0.2
        o, r, dn, i = B.step(a)
        pairs, ncon = B.contacts()
        log.append((ncon.copy(), i[:, 2].copy()))
    B.close()
    return log


def test_hull_contacts_are_a_subset_of_the_capsule_contacts_and_come_later():
    """The hull lies inside the bounding capsule: with hull geometry the arm gets closer to the human before the first contact, and at no substep end does it report
    more contacts than the capsule model would for the same motion up to then."""
    cap, hul = _scene("capsule"), _scene("hull")
    first = lambda log: next((k for k, (nc, _) in enumerate(log) if nc.sum() > 0), None)
    assert first(cap) is not None and first(hul) is not None and first(hul) >= first(cap)
    assert cap[-1][1].sum() > 0 and hul[-1][1].sum() > 0                                           # both count collisions in the end
    assert sum(int(nc.sum()) for nc, _ in hul[:first(hul) + 1]) <= sum(int(nc.sum()) for nc, _ in cap[:first(hul) + 1])


def test_hull_geometry_needs_the_hull_variant_on_the_product_path():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from human_robot_gym_amd._lib import HipBatch
    clips = hrg.synthetic_clips(1, seed=0, min_frames=100, max_frames=120)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        HipBatch(hrg.build_model_desc(None, n_clips=1, robot_geometry="hull"), clips, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("shield", ["OFF", "SSM"])
def test_hip_hull_variant_matches_oracle(shield):
    """hrg_step_kernel_hull vs the oracle with hull geometry: the contact scenario (arm links against the human's arm), resynchronised every step."""
    import torch
    from helpers import ATOL, RTOL, assert_state_close
    from oracle.oracle import OracleBatch
    from human_robot_gym_amd._lib import HipBatch
    clips = hrg.static_clip(600, pelvis=(-0.8, 1.0, 0.3))
    for c in clips.infos:
        c["position_offset"] = [0.0, 0.0, 0.0]
    kw = dict(shield_type=shield, horizon=30, done_at_collision=False, collision_reward=-10)
    n = 16
    mk = lambda: hrg.build_model_desc(kw, n_clips=clips.n_clips, robot_geometry="hull")  # noqa: E731
    O, G = OracleBatch(mk(), clips, n), HipBatch(mk(), clips, n)
    np.testing.assert_allclose(G.reset().cpu().numpy(), O.reset(), rtol=RTOL, atol=ATOL)
    rng = np.random.RandomState(5)
    n_con = 0
    for k in range(26):
        a = rng.uniform(-1, 1, (n, 7))
        a[:, 1] = np.where(np.arange(n) % 2 == 0, 1.0, -1.0)
        a[:, [0, 2, 3, 4, 5]] *= 0.2
        o_o, r_o, d_o, i_o = O.step(a)
        o_g, r_g, d_g, i_g = G.step(torch.from_numpy(a).cuda())
        torch.cuda.synchronize()
        post = [O.get_state(e) for e in range(n)]
        ok = np.array([i_o[e, 11] == 0 and max(abs(v) for v in post[e].qvel) < 5.0 for e in range(n)])
        po, no = O.contacts()
        pg, ng = G.contacts()
        np.testing.assert_array_equal(ng[ok], no[ok], err_msg=f"step {k}")
        np.testing.assert_array_equal(pg[ok], po[ok], err_msg=f"step {k}")
        np.testing.assert_array_equal(i_g.cpu().numpy()[ok], i_o[ok], err_msg=f"step {k}")
        np.testing.assert_allclose(o_g.cpu().numpy()[ok], o_o[ok], rtol=RTOL, atol=1e-6, err_msg=f"step {k}")
        np.testing.assert_allclose(r_g.cpu().numpy()[ok], r_o[ok], rtol=RTOL, atol=1e-6)
        n_con += int(no[ok].sum())
        for e in range(n):
            if ok[e]:
                assert_state_close(post[e], G.get_state(e), f"step {k} env {e}")
            G.set_state(e, post[e])
    assert (n_con > 0 or shield == "SSM") and ok.mean() >= 0.75     # (the SSM shield stops the arm before the human touches it)
    O.close(); G.close()


@pytest.mark.gpu
def test_hip_hull_variant_against_the_table():
    """An arm folded onto the table: link hulls against the table plane (the lowest point of the hull), HIP vs oracle."""
    import torch
    from helpers import RTOL, assert_state_close
    from oracle.oracle import OracleBatch
    from human_robot_gym_amd._lib import HipBatch
    clips = hrg.synthetic_clips(1, seed=0, min_frames=200, max_frames=300)
    kw = dict(shield_type="OFF", horizon=80, done_at_collision=False)
    n = 8
    mk = lambda: hrg.build_model_desc(kw, n_clips=1, robot_geometry="hull")  # noqa: E731
    O, G = OracleBatch(mk(), clips, n), HipBatch(mk(), clips, n)
    O.reset(); G.reset()
    rng = np.random.RandomState(3)
    table = 0
    for k in range(60):
        a = rng.uniform(-0.3, 0.3, (n, 7))
        a[:, 1] = 1.0                                                        # keep folding the shoulder towards the table
        a[:, 2] = np.where(np.arange(n) % 2 == 0, 0.6, -0.2)
        o_o, r_o, d_o, i_o = O.step(a)
        o_g, r_g, d_g, i_g = G.step(torch.from_numpy(a).cuda())
        torch.cuda.synchronize()
        post = [O.get_state(e) for e in range(n)]
        ok = np.array([i_o[e, 11] == 0 and max(abs(v) for v in post[e].qvel) < 5.0 for e in range(n)])
        po, no = O.contacts()
        pg, ng = G.contacts()
        np.testing.assert_array_equal(ng[ok], no[ok], err_msg=f"step {k}")
        np.testing.assert_array_equal(pg[ok], po[ok], err_msg=f"step {k}")
        np.testing.assert_array_equal(i_g.cpu().numpy()[ok], i_o[ok], err_msg=f"step {k}")
        np.testing.assert_allclose(o_g.cpu().numpy()[ok], o_o[ok], rtol=RTOL, atol=1e-6, err_msg=f"step {k}")
        table += int(((po[ok][:, :, 1] == 10 + 24) & (po[ok][:, :, 0] < 7)).sum())
        for e in range(n):
            if ok[e]:
                assert_state_close(post[e], G.get_state(e), f"step {k} env {e}")
            G.set_state(e, post[e])
    assert table > 0, "the scenario was meant to bring an arm link onto the table"
    O.close(); G.close()


@pytest.mark.gpu
def test_hip_wave_routines_match_the_oracle_on_random_poses(oracle_lib):
    """hrg_test_hull_queries: the step kernel's own wave routines (support mapping with lanes = vertices, GJK with its simplex in LDS, lowest point), one wavefront
    per query, against the oracle's scalar restatement -- 3500 random poses over the seven hulls, separated / touching / piercing segments."""
    from human_robot_gym_amd._lib import load_library
    lib = load_library()
    V, off = load_robot_hulls()
    d = hrg.build_model_desc(None, robot_geometry="hull")
    rng = np.random.RandomState(7)
    n = 3500
    q = np.zeros(n, dtype=[("R", "f8", 9), ("p", "f8", 3), ("s1", "f8", 3), ("s2", "f8", 3), ("hull", "i4"), ("pad", "i4")])
    assert q.dtype.itemsize == 152
    want = np.zeros((n, 10))
    o7, o3 = np.zeros(7), np.zeros(3)
    for k in range(n):
        h = k % NH
        R, p = np.ascontiguousarray(_rot(rng)), rng.uniform(-0.5, 0.5, 3)
        c = p + rng.randn(3) * (0.02 if k % 5 == 0 else 0.25)              # every fifth segment starts inside the link's bounding capsule: piercing / touching cases
        s1, s2 = c + rng.randn(3) * 0.15, c - rng.randn(3) * 0.15
        q[k] = (R.ravel(), p, s1, s2, h, 0)
        oracle_lib.hrgo_test_hull_segment(ctypes.byref(d), h, _p(R), _p(p), _p(s1), _p(s2), _p(o7))
        oracle_lib.hrgo_test_hull_lowest(ctypes.byref(d), h, _p(R), _p(p), _p(o3))
        want[k, :7], want[k, 7:] = o7, o3
    got = np.zeros((n, 10))
    Vc, offc = np.ascontiguousarray(V), np.ascontiguousarray(off, np.int32)
    assert lib.hrg_test_hull_queries(_p(Vc), _p(offc), q.ctypes.data_as(ctypes.c_void_p), n, _p(got)) == 0
    pierced = want[:, 0] == 0.0
    assert 50 < pierced.sum() < n // 2
    np.testing.assert_array_equal(got[:, 0] == 0.0, pierced)                 # the same verdict on "the axis pierces the hull"
    np.testing.assert_allclose(got[:, 0], want[:, 0], rtol=1e-10, atol=1e-12)   # distance
    sep = ~pierced
    np.testing.assert_allclose(got[sep, 1:7], want[sep, 1:7], rtol=0, atol=1e-9)   # witness points (a witness can slide along parallel features: both attain the distance)
    np.testing.assert_allclose(got[:, 7:], want[:, 7:], rtol=0, atol=1e-12)  # lowest point
