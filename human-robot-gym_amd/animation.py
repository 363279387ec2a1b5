"""Human animation clips: packed frame table shared by all envs of a batch.

Reference format (utils/animation_utils.py:11-59, utils/convert_bvh.py:76-126): per clip a dict of per-frame
arrays `Pelvis_pos_{x,y,z}`, `Pelvis_quat` (x,y,z,w), `<Joint>_{x,y,z}` for the 23 joints (radians, clipped to
+-1.56) plus an info dict {position_offset, orientation_quat (x,y,z,w), scale}.  The mocap submodule that holds
the real CMU clips is empty in the reference checkout, so benches and tests use SYNTHETIC clips of the same
schema (`synthetic_clips`), as SURVEY.md §8(d) prescribes.
"""
import ctypes
import json

import numpy as np

from ._cstruct import CONST, ClipTable
from .model import load_assets

FRAME_DIM = CONST["HRG_FRAME_DIM"]

DEFAULT_CLIP_NAMES = [  # config/environment/default/reach_human.yaml:18-31
    "CMU/62_01", "CMU/62_03", "CMU/62_04", "CMU/62_07", "CMU/62_09", "CMU/62_10", "CMU/62_12",
    "CMU/62_13", "CMU/62_14", "CMU/62_15", "CMU/62_16", "CMU/62_18", "CMU/62_19",
]


def _qpos_joint_order():
    """Names of the 69 human hinge joints in qpos order (human.xml body DFS order, per body z,y,x)."""
    A = load_assets()
    out = []
    for b in A["human"]["bodies"][1:]:
        out += b["joint_names"]
    return out


class ClipSet:
    """A list of (animation dict, info dict) pairs packed into one float64 frame table."""

    def __init__(self, clips):
        if not 1 <= len(clips) <= CONST["HRG_MAX_CLIPS"]:
            raise ValueError(f"need 1..{CONST['HRG_MAX_CLIPS']} clips")
        order = _qpos_joint_order()
        frames, self.lengths, self.infos = [], [], []
        for anim, info in clips:
            n = len(anim["Pelvis_pos_x"])
            F = np.zeros((n, FRAME_DIM))
            F[:, 0], F[:, 1], F[:, 2] = anim["Pelvis_pos_x"], anim["Pelvis_pos_y"], anim["Pelvis_pos_z"]
            F[:, 3:7] = np.asarray(anim["Pelvis_quat"])
            for k, name in enumerate(order):
                F[:, 7 + k] = anim[name]
            frames.append(F)
            self.lengths.append(n)
            self.infos.append(info or {"position_offset": [0.0, 0.0, 0.0], "orientation_quat": [0.0, 0.0, 0.0, 1.0], "scale": 1.0})
        self.frames = np.ascontiguousarray(np.concatenate(frames, 0))
        self.n_clips = len(clips)

    def table(self):
        """ctypes `hrg_clip_table` pointing at self.frames (keep `self` alive while it is in use)."""
        t = ClipTable()
        t.n_clips = self.n_clips
        off = 0
        for i, n in enumerate(self.lengths):
            t.clip_len[i] = n
            t.clip_offset[i] = off
            t.clip_pos_offset[i][:] = [float(x) for x in self.infos[i]["position_offset"]]
            t.clip_quat[i][:] = [float(x) for x in self.infos[i]["orientation_quat"]]
            info = self.infos[i]
            t.clip_pointing_hand[i] = int(info.get("pointing_hand", "right") == "left")   # pick_place_pointing_human_cartesian_env.py:344-347
            t.clip_holding_hand[i] = int(info.get("object_holding_hand", "right") == "left")   # human_robot_handover_cartesian_env.py:459-463
            if "first_placing_hand" in info:   # stacking clips (collaborative_stacking_cartesian_env.py:512-520): five keyframes, two waiting loops
                amps, speeds = info.get("loop_amplitudes", {}), info.get("loop_speeds", {})
                kf = [int(x) for x in info["keyframes"]]
                if len(kf) < 5 or set(amps) != {"wait_for_second", "wait_for_fourth"}:
                    raise NotImplementedError("stacking animation info: five keyframes and the loops 'wait_for_second' / 'wait_for_fourth'")
                t.clip_stack_keyframes[i][:] = kf[:5]
                t.clip_holding_hand[i] = int(info["first_placing_hand"] == "left")
                for stage, (na, aa, ss) in (("wait_for_second", (None, t.clip_loop_amp, t.clip_loop_speed)), ("wait_for_fourth", (None, t.clip_loop2_amp, t.clip_loop2_speed))):
                    a_, s_ = amps[stage], speeds[stage]
                    if len(a_) > CONST["HRG_MAX_LOOP"] or len(a_) != len(s_):
                        raise NotImplementedError("animation info: up to 4 layered loop sines per stage")
                    for k in range(len(a_)):
                        aa[i][k], ss[i][k] = float(a_[k]), float(s_[k])
                t.clip_n_loop[i], t.clip_n_loop2[i] = len(amps["wait_for_second"]), len(amps["wait_for_fourth"])
                t.clip_loop_amp_std[i] = float(info.get("loop_amplitude_std_factor", 1.0))
                t.clip_loop_speed_std[i] = float(info.get("loop_speed_std_factor", 1.0))
            elif "keyframes" in info:   # animation info of the collaboration tasks (human_object_inspection_cartesian_env.py:447-459, 602-652)
                amps, speeds = info.get("loop_amplitudes", []), info.get("loop_speeds", [])
                if isinstance(amps, dict) and set(amps) == {"present", "wait"}:   # handover clips: two loop stages (440-457)
                    a2, s2 = amps["wait"], speeds["wait"]
                    if len(a2) > CONST["HRG_MAX_LOOP"] or len(a2) != len(s2):
                        raise NotImplementedError("animation info: up to 4 layered loop sines per stage")
                    t.clip_n_loop2[i] = len(a2)
                    for k in range(len(a2)):
                        t.clip_loop2_amp[i][k], t.clip_loop2_speed[i][k] = float(a2[k]), float(s2[k])
                    amps, speeds = amps["present"], speeds["present"]
                if isinstance(amps, dict) or len(amps) > CONST["HRG_MAX_LOOP"] or len(amps) != len(speeds) or len(info["keyframes"]) < 2:
                    raise NotImplementedError("animation info: need two keyframes and up to 4 layered loop sines given as lists")
                t.clip_keyframes[i][:] = [int(info["keyframes"][0]), int(info["keyframes"][1])]
                t.clip_target_pos[i][:] = [float(x) for x in info.get("target_pos", [0.0, 0.0, 0.0])]
                t.clip_n_loop[i] = len(amps)
                for k in range(len(amps)):
                    t.clip_loop_amp[i][k], t.clip_loop_speed[i][k] = float(amps[k]), float(speeds[k])
                t.clip_loop_amp_std[i] = float(info.get("loop_amplitude_std_factor", 1.0))
                t.clip_loop_speed_std[i] = float(info.get("loop_speed_std_factor", 1.0))
            off += n
        t.frames = self.frames.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        t.total_frames = off
        return t


def _rot(axis, ang):
    c, s_ = np.cos(ang), np.sin(ang)
    i, j = [(1, 2), (2, 0), (0, 1)][axis]
    R = np.zeros(ang.shape + (3, 3))
    R[..., axis, axis] = 1.0
    R[..., i, i], R[..., j, j], R[..., i, j], R[..., j, i] = c, c, -s_, s_
    return R


def _quat_xyzw_to_mat(q):
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0], R[..., 0, 1], R[..., 0, 2] = 1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)
    R[..., 1, 0], R[..., 1, 1], R[..., 1, 2] = 2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)
    R[..., 2, 0], R[..., 2, 1], R[..., 2, 2] = 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)
    return R


def hand_sites(frames, info, with_rot=False):
    """World positions of the L_Hand / R_Hand sites for every frame [n, FRAME_DIM] of a clip placed by `info` (no per-episode offsets):
    the pose chain of HumanEnv._control_human (human_env.py:1736-1763) and the tree of human.xml, vectorised over frames.  Host-side
    helper for shaping synthetic clips; the steppers have their own kinematics."""
    bodies = load_assets()["human"]["bodies"]
    n = len(frames)
    Rb = _quat_xyzw_to_mat(np.array([0.5, 0.5, 0.5, 0.5]))          # human_base_quat (human_env.py:373)
    Ri = _quat_xyzw_to_mat(np.asarray(info["orientation_quat"], float))
    Rbi = Rb @ Ri
    R = [None] * len(bodies)
    p = [None] * len(bodies)
    R[0] = Rbi @ _quat_xyzw_to_mat(frames[:, 3:7])
    p[0] = (frames[:, 0:3] + np.asarray(info["position_offset"], float)) @ Rbi.T
    for b in range(1, len(bodies)):
        par, anc = bodies[b]["parent"], np.asarray(bodies[b]["anchor"], float)
        q = frames[:, 7 + 3 * (b - 1): 10 + 3 * (b - 1)]
        R[b] = R[par] @ _rot(2, q[:, 0]) @ _rot(1, q[:, 1]) @ _rot(0, q[:, 2])      # z, y, x
        p[b] = p[par] + R[par] @ anc - R[b] @ anc
    names = [b["name"] for b in bodies]
    out, rot = [], []
    for nm in ("L_Hand", "R_Hand"):
        b = names.index(nm)
        out.append(p[b] + np.einsum("nij,j->ni", R[b], np.asarray(bodies[b]["anchor"], float)))
        rot.append(R[b])
    assert out[0].shape == (n, 3)
    if with_rot:
        return out[0], out[1], rot[0], rot[1]
    return out[0], out[1]


def synthetic_clips(n_clips=13, seed=0, min_frames=1200, max_frames=3000, fps=120.0, stand_off=1.2, inspection=False, handover=False, lifting=None, lift_height=0.7,
                    choreographed=False, stacking=False, hammering=None):
    """Band-limited random joint motion (sigma 0.3 rad, 2 Hz cut-off, clipped to +-1.56) and a slow pelvis
    random walk within +-0.3 m around a standing pose, in the BVH (Y-up) frame the reference clips use."""
    rng = np.random.RandomState(seed)
    order = _qpos_joint_order()
    clips = []
    for _ in range(n_clips):
        n = int(rng.randint(min_frames, max_frames + 1))
        t = np.arange(n) / fps

        def band(sigma, cutoff, k=6):
            f = rng.uniform(0.05, cutoff, size=k)
            ph = rng.uniform(0, 2 * np.pi, size=k)
            a = rng.randn(k)
            a *= sigma / np.sqrt(0.5 * np.sum(a * a) + 1e-12)
            return (a[:, None] * np.sin(2 * np.pi * f[:, None] * t[None, :] + ph[:, None])).sum(0)

        rigid = lifting is not None or hammering is not None
        anim = {}
        anim["Pelvis_pos_x"] = np.clip(band(0.15, 0.3), -0.3, 0.3)
        anim["Pelvis_pos_y"] = 1.0 + np.clip(band(0.01, 1.0), -0.03, 0.03)
        anim["Pelvis_pos_z"] = np.clip(band(0.15, 0.3), -0.3, 0.3)
        yaw = band(0.4, 0.2)
        anim["Pelvis_quat"] = np.stack([np.zeros(n), np.sin(yaw / 2), np.zeros(n), np.cos(yaw / 2)], 1)  # about Y (up)
        for name in order:
            sigma = 0.0 if name.split("_")[-2] in ("Spine", "Toe", "Hand") else (0.0 if rigid else 0.3)  # convert_bvh.py:55-72 None joints (lifting / hammering: a rigid posture, the motion is the pelvis track)
            mean = {"L_Shoulder_z": -1.1, "R_Shoulder_z": 1.1}.get(name, 0.0)  # arms hang down instead of the T-pose
            if rigid:  # ... and a little closer: the hands half a metre apart, like the board's grips
                mean = {"L_Shoulder_z": -1.43, "R_Shoulder_z": 1.43}.get(name, 0.0)
            anim[name] = np.clip(mean + band(sigma, 2.0), -1.56, 1.56) if sigma > 0 else np.full(n, mean if rigid else 0.0)
        # the clip's info file places the human at the table edge in front of the robot: BVH +z maps to world +x
        # under human_base_quat (human_env.py:373), so 1.2 m along z = 1.2 m in front of the robot base
        info = {"position_offset": [0.0, 0.0, stand_off], "orientation_quat": [0.0, 0.0, 0.0, 1.0], "scale": 1.0}
        if inspection:  # stand-in for the ObjectInspection/* info files: approach until 30 %, inspect until 70 %, idle loop of two layered sines
            info.update(keyframes=[int(0.3 * n), int(0.7 * n)], target_pos=[0.55, float(rng.uniform(-0.1, 0.1)), 1.15],
                        loop_amplitudes=[25.0, 8.0], loop_speeds=[1.0, 0.45], loop_amplitude_std_factor=1.1, loop_speed_std_factor=1.1)
        if handover == "r2h":   # stand-in for the RobotHumanHandover/* info files: the hand is held out between 30 % and 60 %
            info.update(keyframes=[int(0.3 * n), int(0.6 * n)], object_holding_hand="left" if len(clips) % 2 else "right",
                        loop_amplitudes=[15.0, 5.0], loop_speeds=[1.0, 0.5], loop_amplitude_std_factor=1.1, loop_speed_std_factor=1.1)
        elif handover:  # stand-in for the HumanRobotHandover/* info files: present from 30 %, wait at 60 %, two loop stages
            info.update(keyframes=[int(0.3 * n), int(0.6 * n)], object_holding_hand="left" if len(clips) % 2 else "right",
                        loop_amplitudes=dict(present=[15.0, 5.0], wait=[12.0]), loop_speeds=dict(present=[1.0, 0.5], wait=[0.8]),
                        loop_amplitude_std_factor=1.1, loop_speed_std_factor=1.1)
        if hammering is not None:
            # stand-in for the CollaborativeHammering/* recordings and info files: the board is presented from 30 % on, idle loop around the middle of the
            # keyframes.  `hammering` = world position of the middle between the hands while the board is presented; a rigid posture turned HAMMERING_YAW
            # about the vertical, so that the line between the hands matches the line between the board's two grips (the left one 0.4 m closer to
            # the robot); the human comes 0.4 m closer during the approach and steps back after the second keyframe
            info.update(keyframes=[int(0.3 * n), int(0.6 * n)], loop_amplitudes=[15.0, 5.0], loop_speeds=[1.0, 0.5], loop_amplitude_std_factor=1.1, loop_speed_std_factor=1.1)
            anim["Pelvis_quat"] = np.tile(np.array([0.0, np.sin(HAMMERING_YAW / 2), 0.0, np.cos(HAMMERING_YAW / 2)]), (n, 1))
            F = np.zeros((n, FRAME_DIM))
            F[:, 0], F[:, 1], F[:, 2] = anim["Pelvis_pos_x"], anim["Pelvis_pos_y"], anim["Pelvis_pos_z"]
            F[:, 3:7] = anim["Pelvis_quat"]
            for k, name in enumerate(order):
                F[:, 7 + k] = anim[name]
            lh, rh = hand_sites(F, info)
            k0, k1 = info["keyframes"]
            f = np.arange(n, dtype=float)
            up = np.clip(f / max(k0, 1.0), 0.0, 1.0)
            down = np.clip((f - k1) / max(0.5 * (n - k1), 1.0), 0.0, 1.0)
            w = (up * up * (3 - 2 * up)) * (1 - down * down * (3 - 2 * down))
            u = t / t[-1]
            sway = 0.02 * np.sin(2 * np.pi * u * rng.uniform(0.5, 1.5))
            want = np.asarray(hammering, float)[None, :] + np.stack([0.4 * (1 - w), sway, 0.0 * u], 1)
            Rb = _quat_xyzw_to_mat(np.array([0.5, 0.5, 0.5, 0.5])) @ _quat_xyzw_to_mat(np.asarray(info["orientation_quat"], float))
            shift = (want - 0.5 * (lh + rh)) @ Rb
            anim["Pelvis_pos_x"] = anim["Pelvis_pos_x"] + shift[:, 0]
            anim["Pelvis_pos_y"] = anim["Pelvis_pos_y"] + shift[:, 1]
            anim["Pelvis_pos_z"] = anim["Pelvis_pos_z"] + shift[:, 2]
        if stacking:   # stand-in for the CollaborativeStacking/* info files: keyframes at 15 / 25 / 35 / 55 / 65 % of the clip, alternating first hand, two waiting loops
            info.update(keyframes=[int(f * n) for f in (0.15, 0.25, 0.35, 0.55, 0.65)], first_placing_hand="left" if len(clips) % 2 else "right",
                        loop_amplitudes=dict(wait_for_second=[15.0, 5.0], wait_for_fourth=[12.0]), loop_speeds=dict(wait_for_second=[1.0, 0.5], wait_for_fourth=[0.8]),
                        loop_amplitude_std_factor=1.1, loop_speed_std_factor=1.1)
        if handover and choreographed:
            # a handover one can act on: the human stands still 1.2 m in front of the robot, facing it, arms down; between the keyframes the
            # holding arm is stretched out over the table (hand ~0.66 m in front of the robot base at the default stand-off, 1.02 m high), blending in before the first
            # keyframe and out after the second
            k0, k1 = info["keyframes"]
            f = np.arange(n, dtype=float)
            up = np.clip((f - 0.5 * k0) / max(0.5 * k0, 1.0), 0.0, 1.0)
            down = np.clip((f - k1) / max(0.5 * (n - k1), 1.0), 0.0, 1.0)
            w = (up * up * (3 - 2 * up)) * (1 - down * down * (3 - 2 * down))
            left = info["object_holding_hand"] == "left"
            for name in order:
                anim[name] = np.zeros(n)
            anim["L_Shoulder_z"], anim["R_Shoulder_z"] = np.full(n, -1.43), np.full(n, 1.43)
            side, sg = ("L", -1.0) if left else ("R", 1.0)
            anim[f"{side}_Shoulder_z"] = sg * (1.43 + w * (1.25 - 1.43))
            anim[f"{side}_Shoulder_y"] = sg * 1.25 * w
            anim[f"{side}_Shoulder_x"] = -0.75 * w
            anim["Pelvis_pos_x"], anim["Pelvis_pos_y"], anim["Pelvis_pos_z"] = np.zeros(n), np.ones(n), np.zeros(n)
            anim["Pelvis_quat"] = np.tile(np.array([0.0, 1.0, 0.0, 0.0]), (n, 1))
        if lifting is not None:
            # stand-in for the CollaborativeLifting/* recordings: a human facing the robot who raises and lowers the far end of the board.
            # `lifting` = world position of the middle between the two hands at the first frame (where the board's grips are at a reset);
            # the pelvis track is shifted frame by frame so that the middle of the hands follows a smooth lift of 60-70 cm with a
            # little sideways sway; the posture itself is rigid and symmetric, so the hands stay level and half a metre apart
            anim["Pelvis_quat"] = np.tile(np.array([0.0, 1.0, 0.0, 0.0]), (n, 1))   # half a turn about the vertical: facing the robot, left hand at -y
            F = np.zeros((n, FRAME_DIM))
            F[:, 0], F[:, 1], F[:, 2] = anim["Pelvis_pos_x"], anim["Pelvis_pos_y"], anim["Pelvis_pos_z"]
            F[:, 3:7] = anim["Pelvis_quat"]
            for k, name in enumerate(order):
                F[:, 7 + k] = anim[name]
            lh, rh = hand_sites(F, info)
            u = t / t[-1]
            lift = lift_height * np.sin(np.pi * u) ** 2 * rng.uniform(0.85, 1.0)   # default 0.6-0.7 m: a robot that does not follow tips the board past min_balance
            sway = 0.04 * np.sin(2 * np.pi * u * rng.uniform(0.5, 1.5))
            want = np.asarray(lifting, float)[None, :] + np.stack([0.0 * u, sway, lift], 1)
            Rb = _quat_xyzw_to_mat(np.array([0.5, 0.5, 0.5, 0.5])) @ _quat_xyzw_to_mat(np.asarray(info["orientation_quat"], float))
            shift = (want - 0.5 * (lh + rh)) @ Rb          # world shift -> animation frame (p_world = Rb p_anim)
            anim["Pelvis_pos_x"] = anim["Pelvis_pos_x"] + shift[:, 0]
            anim["Pelvis_pos_y"] = anim["Pelvis_pos_y"] + shift[:, 1]
            anim["Pelvis_pos_z"] = anim["Pelvis_pos_z"] + shift[:, 2]
        clips.append((anim, info))
    return ClipSet(clips)


HAMMERING_YAW = np.pi * 0.75        # turn of the synthetic hammering human about the vertical (BVH y axis)
HAMMERING_HANDS = (1.05, 0.0, 1.05)  # middle between the hands while the board is presented: the nail area lies 0.3 - 0.7 m in front of the robot base, 0.25 m above the table


def hammering_relquat(anchors=((-0.1, 0.2, 0.0), (-0.5, -0.2, 0.0))):
    """Relative pose quaternion (w, x, y, z) of the right-hand weld of CollaborativeHammeringCart for which the SYNTHETIC human of
    `synthetic_clips(hammering=...)` holds the board level: q_mocap = q_board o q_rel with the mocap frame = right-hand body turned +90 deg about y
    (collaborative_hammering_cartesian_env.py:688-715) and the board yawed so that its two grips lie on the line between the hands.  The reference's
    value (0, 0, 0, 1) belongs to the recorded clips' hand orientations, which are absent."""
    order = _qpos_joint_order()
    F = np.zeros((1, FRAME_DIM))
    F[0, 1] = 1.0
    F[0, 3:7] = [0.0, np.sin(HAMMERING_YAW / 2), 0.0, np.cos(HAMMERING_YAW / 2)]
    for k, name in enumerate(order):
        F[0, 7 + k] = {"L_Shoulder_z": -1.43, "R_Shoulder_z": 1.43}.get(name, 0.0)
    info = {"position_offset": [0.0, 0.0, 1.2], "orientation_quat": [0.0, 0.0, 0.0, 1.0]}
    lh, rh, _, Rr = hand_sites(F, info, with_rot=True)
    v = (lh - rh)[0]
    a = np.asarray(anchors[0], float) - np.asarray(anchors[1], float)
    th = np.arctan2(v[1], v[0]) - np.arctan2(a[1], a[0])
    Rboard = np.array([[np.cos(th), -np.sin(th), 0.0], [np.sin(th), np.cos(th), 0.0], [0.0, 0.0, 1.0]])
    Ry = np.array([[0.0, 0.0, 1.0], [0.0, 1.0, 0.0], [-1.0, 0.0, 0.0]])       # +90 deg about y
    Rrel = Rboard.T @ (Rr[0] @ Ry)
    w = 0.5 * np.sqrt(max(1.0 + np.trace(Rrel), 1e-12))
    q = np.array([w, (Rrel[2, 1] - Rrel[1, 2]) / (4 * w), (Rrel[0, 2] - Rrel[2, 0]) / (4 * w), (Rrel[1, 0] - Rrel[0, 1]) / (4 * w)])
    return (q / np.linalg.norm(q)).tolist()


def lifting_hands_nominal(desc):
    """Where the board's two hand grips are at a reset of CollaborativeLiftingCart (middle between them): the argument `lifting` of
    `synthetic_clips`.  The board sits in the gripper with its robot-side edge `lift_grip_depth` past the grip site."""
    from .model import robot_fk_numpy
    q = list(desc.init_qpos[:]) + [0.0] * (CONST["HRG_NV"] - len(desc.init_qpos[:]))
    R, p = robot_fk_numpy(desc, q)
    k = CONST["HRG_NARM"] - 1
    ze, eef = R[k][:, 2], p[k] + R[k] @ np.asarray(desc.eef_pos[:])
    ax = 0.5 * (np.asarray(desc.lift_anchor[0][:]) + np.asarray(desc.lift_anchor[1][:]))
    return eef + ze * (desc.box_half[0] - desc.lift_grip_depth - ax[0])


def static_clip(n_frames=600, pelvis=(0.0, 1.0, 0.0)):
    """A T-pose human standing still (the `Static/tpose` clip of the shield demos)."""
    order = _qpos_joint_order()
    anim = {k: np.zeros(n_frames) for k in order}
    anim["Pelvis_pos_x"] = np.full(n_frames, pelvis[0])
    anim["Pelvis_pos_y"] = np.full(n_frames, pelvis[1])
    anim["Pelvis_pos_z"] = np.full(n_frames, pelvis[2])
    anim["Pelvis_quat"] = np.tile(np.array([0.0, 0.0, 0.0, 1.0]), (n_frames, 1))
    return ClipSet([(anim, None)])


def load_clips_npz(paths):
    """Load user-supplied clips stored as .npz (same keys as the reference pkl dicts) + optional _info.json."""
    clips = []
    for p in paths:
        with np.load(p, allow_pickle=False) as z:
            anim = {k: z[k] for k in z.files}
        info = None
        try:
            with open(p[: -len(".npz")] + "_info.json") as f:
                info = json.load(f)
        except FileNotFoundError:
            pass
        clips.append((anim, info))
    return ClipSet(clips)
