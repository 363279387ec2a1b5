#!/usr/bin/env python3
"""Mint the golden rollouts under tests/golden/ with the CPU oracle.

The reference holds no golden vectors for this path and cannot run here, so these fixtures pin
ORACLE <-> HIP agreement (and guard the oracle against silent drift) — not agreement with MuJoCo/sara-shield.
    python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import human_robot_gym_amd as hrg  # noqa: E402
from oracle.oracle import OracleBatch  # noqa: E402

CASES = {
    "reach_off": dict(shield_type="OFF", reward_shaping=True, horizon=25),
    "reach_ssm": dict(shield_type="SSM", reward_shaping=True, horizon=25),
    "reach_ssm_freq5": dict(shield_type="SSM", control_freq=5, horizon=1000, human_rand=[0.2, 0.2, 0.2]),  # config 1 shape: 50 cycles/step
    "contact_static": dict(shield_type="OFF", horizon=60, done_at_collision=False, collision_reward=-10),
    # PickPlaceHumanCart (BASELINE config 4 shape): the cube pops out of the table, rests, gets re-placed after a delivery
    "pick_place_ssm": dict(env_id="PickPlaceHumanCart", shield_type="SSM", reward_shaping=True, horizon=30),
    # the collaboration tasks (short clips, so that their phase machines move within 40 steps); small arm actions keep them out of chaotic regimes
    "inspection_ssm": dict(env_id="HumanObjectInspectionCart", shield_type="SSM", horizon=60),
    "handover_h2r_pfl": dict(env_id="HumanRobotHandoverCart", shield_type="PFL", horizon=60),
    "handover_r2h_pfl": dict(env_id="RobotHumanHandoverCart", shield_type="PFL", horizon=60),
    "lifting_ssm": dict(env_id="CollaborativeLiftingCart", shield_type="SSM", horizon=60),
    # four free cubes: two resting on the table, two welded to the hands; the human drops its first cube at its keyframe (it falls to the table / floor)
    "stacking_ssm": dict(env_id="CollaborativeStackingCart", shield_type="SSM", horizon=60),
    # board (weld + connect to the hands) with the nail on its slide joint, hammer in the closed gripper; the nail creeps in under its own weight
    "hammering_ssm": dict(env_id="CollaborativeHammeringCart", shield_type="SSM", horizon=60),
}
# cases whose free-running GPU rollout is compared against the fixture (tests/test_golden.py): all of them
GPU_CASES = list(CASES)


def object_rows(B, env_id, n):
    """(float rows, integer rows) of the manipulation object(s) of every env: what the fixtures store as `box` / `phase`."""
    if env_id == "CollaborativeStackingCart":
        sks = [B.get_stack(e) for e in range(n)]
        fl = [[x for c in range(4) for x in list(s.pos[c]) + list(s.quat[c]) + list(s.vel[c])] + list(s.target) for s in sks]
        it = [[s.task_phase, s.weld_active[0], s.weld_active[1], s.gripped, s.n_stack, s.max_stack_height, s.has_target] + list(s.stack_ids) for s in sks]
        return np.array(fl), np.array(it, np.int32)
    if env_id == "CollaborativeHammeringCart":
        hms = [B.get_hammer(e) for e in range(n)]
        fl = [[x for b in range(2) for x in list(h.pos[b]) + list(h.quat[b]) + list(h.vel[b])] + [h.nail_q, h.nail_v] + list(h.nail_xy) for h in hms]
        it = [[h.task_phase, h.gripped, h.nail_index, h.n_delayed] for h in hms]
        return np.array(fl), np.array(it, np.int32)
    bxs = [B.get_box(e) for e in range(n)]
    return (np.array([list(b.pos) + list(b.quat) + list(b.vel) + list(b.target) for b in bxs]),
            np.array([[b.task_phase, b.weld_active, b.gripped, b.n_handed_over] for b in bxs], np.int32))


def clips_for(name):
    if name == "contact_static":  # T-pose human whose left hand is 0.3 m from the upright arm (collision scenario)
        return hrg.static_clip(600, pelvis=(-0.8, 1.0, 0.3))
    env_id = CASES[name].get("env_id", "ReachHuman")
    if env_id not in ("ReachHuman", "PickPlaceHumanCart"):
        from human_robot_gym_amd.mixed import task_clips
        if env_id == "CollaborativeStackingCart":   # a human moving at a quarter of the synthetic clips' speed: its welded cubes are carried, not flung
            return task_clips(env_id, 3, min_frames=480, max_frames=640, fps=480.0)
        return task_clips(env_id, 3, min_frames=120, max_frames=160)
    return hrg.synthetic_clips(3, seed=0, min_frames=300, max_frames=600)


def run(name, kw, n_envs=8, n_steps=40, seed=11):
    clips = clips_for(name)
    kw = dict(kw)
    env_id = kw.pop("env_id", "ReachHuman")
    from human_robot_gym_amd.mixed import task_env_kwargs
    kw.update(task_env_kwargs(env_id))
    d = hrg.build_model_desc(kw, n_clips=clips.n_clips, env_id=env_id)
    B = OracleBatch(d, clips, n_envs)
    out = dict(obs0=B.reset())
    rng = np.random.RandomState(seed)
    acts, obs, rew, done, info, qpos, qvel, ncon, pairs, box, phase = [], [], [], [], [], [], [], [], [], [], []
    for k in range(n_steps):
        a = rng.uniform(-1, 1, (n_envs, 7))
        if env_id not in ("ReachHuman", "PickPlaceHumanCart"):
            a[:, :6] *= 0.2
        if env_id == "PickPlaceHumanCart" and k == 20:  # a delivery: cube teleported next to its target
            for e in range(n_envs):
                bx = B.get_box(e)
                bx.pos[:] = [bx.target[0] + 0.02, bx.target[1], 0.845]
                B.set_box(e, bx)
        if name == "contact_static":  # tilt the arm towards / away from the hand
            a[:, 1] = np.where(np.arange(n_envs) % 2 == 0, 1.0, -1.0)
            a[:, [0, 2, 3, 4, 5]] *= 0.2
        o, r, dn, i = B.step(a)
        acts.append(a); obs.append(o); rew.append(r); done.append(dn); info.append(i)
        st = [B.get_state(e) for e in range(n_envs)]
        qpos.append([list(s.qpos) for s in st]); qvel.append([list(s.qvel) for s in st])
        p, n = B.contacts()
        ncon.append(n); pairs.append(p)
        fl, it = object_rows(B, env_id, n_envs)
        box.append(fl); phase.append(it)
    out.update(phase=np.array(phase, np.int32))
    out.update(actions=np.array(acts), obs=np.array(obs), reward=np.array(rew), done=np.array(done), info=np.array(info),
               qpos=np.array(qpos), qvel=np.array(qvel), ncon=np.array(ncon), pairs=np.array(pairs).astype(np.int8), box=np.array(box))
    return out


if __name__ == "__main__":
    only = sys.argv[1:]
    for name, kw in CASES.items():
        if only and name not in only:
            continue
        out = run(name, kw, n_steps=24 if name == "contact_static" else 40)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"{name}.npz"), **out)
        print(name, {k: v.shape for k, v in out.items()}, "dones", int(out["done"].sum()), "failsafe", int(out["info"][-1, :, 8].sum()))
