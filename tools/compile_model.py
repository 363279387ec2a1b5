#!/usr/bin/env python3
"""Model compiler: reference MJCF/STL assets -> compact JSON tables for the batched stepper.

Runs ONCE, in the build container, where the reference assets are readable as data:
    python tools/compile_model.py /root/reference/human_robot_gym/models/assets \
        human-robot-gym_amd/assets/reach_human_schunk.json
The JSON it writes is committed; nothing at run time (tests, bench, smoke) reads /root/reference.

What it extracts (data only, no code):
  * Schunk LWA-4P kinematic chain, joint axes/ranges/damping/frictionloss, link inertials,
    motor ctrl ranges            <- robots/schunk/robot.xml:4-9, 25-71
  * bounding capsules of the 7 arm collision meshes  <- robots/schunk/meshes/*.stl (robot.xml:12-18)
  * human kinematic tree (24 bodies, 23x3 hinges, joint anchors = site positions), body inertials
                                  <- human/human.xml:42-330
  * bounding capsules of the 24 human collision meshes <- human/meshes/*.stl
  * table slab / floor            <- arenas/table_arena.xml:22,25 + reach_human_env.py:309-312
Parts of the stack that are NOT in the reference tree (robosuite RethinkGripper/RethinkMount MJCF,
sara-shield YAML parameter files) are filled with documented SYNTHETIC defaults.
"""
import json
import struct
import sys
import xml.etree.ElementTree as ET

import numpy as np


def load_stl(path):
    b = open(path, "rb").read()
    n = struct.unpack("<I", b[80:84])[0]
    assert len(b) == 84 + 50 * n, path
    a = np.frombuffer(b[84:], dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]))
    return np.unique(a["v"].reshape(-1, 3).astype(np.float64), axis=0)


def quat_to_mat(q):
    w, x, y, z = np.asarray(q, float) / np.linalg.norm(q)
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
    ])


def fit_capsule(V):
    """Tight bounding capsule of a point set for the PCA axis through the centroid.

    Returns (p1, p2, r). Every vertex lies inside the capsule (checked)."""
    c = V.mean(0)
    X = V - c
    w, U = np.linalg.eigh(X.T @ X)
    d = U[:, -1]
    # deterministic sign
    k = int(np.argmax(np.abs(d)))
    if d[k] < 0:
        d = -d
    t = X @ d
    rho = np.linalg.norm(X - np.outer(t, d), axis=1)
    r = float(rho.max()) * (1 + 1e-9) + 1e-9
    s = np.sqrt(np.maximum(r * r - rho * rho, 0.0))
    a_hi = float((t - s).max())
    a_lo = float((t + s).min())
    if a_lo > a_hi:  # degenerate -> sphere-like; grow radius
        m = 0.5 * (a_lo + a_hi)
        a_lo = a_hi = m
        r = float(np.sqrt(((t - m) ** 2 + rho ** 2).max())) * (1 + 1e-9)
    p1 = c + a_lo * d
    p2 = c + a_hi * d
    # verify
    ab = p2 - p1
    L2 = float(ab @ ab)
    for v in V:
        u = 0.0 if L2 == 0 else min(1.0, max(0.0, float((v - p1) @ ab) / L2))
        assert np.linalg.norm(v - (p1 + u * ab)) <= r * (1 + 1e-6) + 1e-9
    return p1.tolist(), p2.tolist(), r


def fnum(s):
    return [float(x) for x in s.split()]


def compile_robot(assets):
    root = ET.parse(f"{assets}/robots/schunk/robot.xml").getroot()
    meshes = {m.get("name"): m.get("file") for m in root.find("asset").findall("mesh")}
    ctrl = [fnum(m.get("ctrlrange")) for m in root.find("actuator").findall("motor")]
    links = []
    fixed_geoms = []

    def walk(body, chain):
        name = body.get("name")
        pos = fnum(body.get("pos", "0 0 0"))
        quat = fnum(body.get("quat", "1 0 0 0"))
        inert = body.find("inertial")
        joint = body.find("joint")
        geoms = [g for g in body.findall("geom") if g.get("contype") != "0"]
        rec = dict(name=name, pos=pos, quat=quat,
                   mass=float(inert.get("mass")), ipos=fnum(inert.get("pos")),
                   diaginertia=fnum(inert.get("diaginertia")))
        if joint is not None:
            rec.update(joint=joint.get("name"), axis=fnum(joint.get("axis")),
                       range=fnum(joint.get("range")), damping=float(joint.get("damping")),
                       frictionloss=float(joint.get("frictionloss")), armature=0.0)
        caps = []
        for g in geoms:
            V = load_stl(f"{assets}/robots/schunk/{meshes[g.get('mesh')]}")
            gp = np.array(fnum(g.get("pos", "0 0 0")))
            gR = quat_to_mat(fnum(g.get("quat", "1 0 0 0")))
            V = V @ gR.T + gp  # into body frame
            p1, p2, r = fit_capsule(V)
            caps.append(dict(name=g.get("name"), p1=p1, p2=p2, r=r, nvert=int(len(V))))
        rec["capsules"] = caps
        chain.append(rec)
        for c in body.findall("body"):
            walk(c, chain)

    base = root.find("worldbody").find("body")
    chain = []
    for c in base.findall("body"):
        walk(c, chain)
    return dict(chain=chain, ctrlrange=ctrl)


def compile_human(assets):
    root = ET.parse(f"{assets}/human/human.xml").getroot()
    meshes = {m.get("name"): m.get("file") for m in root.find("asset").findall("mesh")}
    dflt = root.find("default")
    armature = float(dflt.find("joint").get("armature"))
    margin = float(dflt.find("geom").get("margin").strip())
    bodies = []

    def walk(body, parent):
        name = body.get("name")
        inert = body.find("inertial")
        joints = body.findall("joint")
        g = [x for x in body.findall("geom")][0]
        V = load_stl(f"{assets}/human/{meshes[g.get('mesh')]}")
        p1, p2, r = fit_capsule(V)
        rec = dict(name=name, parent=parent, mass=float(inert.get("mass")),
                   diaginertia=fnum(inert.get("diaginertia")),
                   anchor=fnum(joints[0].get("pos")) if joints else [0.0, 0.0, 0.0],
                   joint_axes=[fnum(j.get("axis")) for j in joints],
                   joint_names=[j.get("name") for j in joints],
                   geom=g.get("name"), capsule=dict(p1=p1, p2=p2, r=r, nvert=int(len(V))))
        idx = len(bodies)
        bodies.append(rec)
        for c in body.findall("body"):
            walk(c, idx)

    pelvis = root.find("worldbody").find("body").find("body").find("body")
    assert pelvis.get("name") == "Pelvis"
    walk(pelvis, -1)
    return dict(bodies=bodies, armature=armature, margin=margin)


def compile_hulls(assets, chain):
    """Convex hulls of the seven arm collision meshes (robot.xml:29-55; MuJoCo convexifies every collision mesh at compile time [UPSTREAM: qhull]): the hull's
    vertices in the frame of the body the geom hangs on, in the order of the chain's capsules (link0 .. link6).  scipy's Qhull wrapper runs here, at model-compile
    time only; the stepper reads the table.  Returns (verts [N, 3] float64, offsets [8] int32)."""
    from scipy.spatial import ConvexHull
    root = ET.parse(f"{assets}/robots/schunk/robot.xml").getroot()
    meshes = {m.get("name"): m.get("file") for m in root.find("asset").findall("mesh")}
    geoms = {g.get("name"): g for g in root.iter("geom") if g.get("contype") != "0" and g.get("type") == "mesh"}
    verts, offs = [], [0]
    for rec in chain:
        for cap in rec["capsules"]:
            g = geoms[cap["name"]]
            V = load_stl(f"{assets}/robots/schunk/{meshes[g.get('mesh')]}")
            V = V @ quat_to_mat(fnum(g.get("quat", "1 0 0 0"))).T + np.array(fnum(g.get("pos", "0 0 0")))
            H = V[np.sort(ConvexHull(V).vertices)]
            # the bounding capsule of the mesh bounds its hull: the capsule narrowphase stays a valid broadphase for the hull
            p1, p2, r = np.array(cap["p1"]), np.array(cap["p2"]), cap["r"]
            ab = p2 - p1
            u = np.clip((H - p1) @ ab / max(float(ab @ ab), 1e-300), 0, 1)
            assert np.linalg.norm(H - (p1 + u[:, None] * ab), axis=1).max() <= r * (1 + 1e-6) + 1e-9
            verts.append(H)
            offs.append(offs[-1] + len(H))
    return np.concatenate(verts), np.array(offs, np.int32)


def main():
    assets, out = sys.argv[1], sys.argv[2]
    model = dict(
        provenance=("compiled by tools/compile_model.py from the reference's MJCF/STL data files "
                    "(robots/schunk/robot.xml, human/human.xml, meshes/*.stl, arenas/table_arena.xml)"),
        robot=compile_robot(assets),
        human=compile_human(assets),
        # reach_human_env.py:309-312 (table_offset z=0.82), config default/reach_human.yaml table_full_size
        arena=dict(table_top_z=0.82, table_half=[0.75, 1.0], floor_z=0.0),
    )
    with open(out, "w") as f:
        json.dump(model, f, indent=1)
    hv, ho = compile_hulls(assets, model["robot"]["chain"])
    import os
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(out)), "schunk_hulls.npz"), verts=hv, offsets=ho)
    print(f"hull vertices per link {np.diff(ho).tolist()}")
    nb = len(model["human"]["bodies"])
    print(f"robot bodies {len(model['robot']['chain'])}, human bodies {nb}")


if __name__ == "__main__":
    main()
