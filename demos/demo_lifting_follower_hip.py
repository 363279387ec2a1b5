#!/usr/bin/env python3
"""CollaborativeLiftingCart on the HIP stepper with a scripted follower: Cartesian actions (IKPositionDeltaWrapper front-end) keep the gripper level
with the middle of the human's hands while the human raises and lowers their end of the board.  Prints how the episodes end.  Needs an MI355X.

    python demos/demo_lifting_follower_hip.py [--n-envs 64] [--steps 150] [--rest]      (--rest: the robot does not move -> the board tips)
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from human_robot_gym_amd.mixed import task_clips  # noqa: E402
from human_robot_gym_amd.vec_env import HipVecEnv  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-envs", type=int, default=64)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--rest", action="store_true")
    args = ap.parse_args()
    n = args.n_envs
    clips = task_clips("CollaborativeLiftingCart", 4, min_frames=200, max_frames=260)          # 10-13 s at 20 Hz
    env = HipVecEnv(n, env_id="CollaborativeLiftingCart", env_kwargs=dict(seed=0, horizon=400), clips=clips,
                    obs_keys=["vec_eef_to_human_lh", "vec_eef_to_human_rh", "board_balance", "board_gripped"], ik_position_delta=dict(action_limit=0.15))
    obs = env.reset()
    wins = fails = 0
    held = []
    for t in range(args.steps):
        mid = 0.5 * (obs[:, 0:3] + obs[:, 3:6])                                                  # gripper -> middle of the hands
        a = np.zeros((n, 4))
        if not args.rest:
            a[:, 0] = np.clip(mid[:, 0] - 0.95, -0.15, 0.15)                                      # the board's grips are 0.95 m from its robot-side edge
            a[:, 1] = np.clip(mid[:, 1], -0.15, 0.15)
            a[:, 2] = np.clip(1.5 * mid[:, 2], -0.15, 0.15)
        obs, rew, done, infos = env.step(a)
        held.append(float(obs[:, 7].mean()))
        for i in np.nonzero(done)[0]:
            if infos[i]["n_goal_reached"] > 0:
                wins += 1
            else:
                fails += 1
        if t % 25 == 0:
            print(f"t={t:4d} balance min {obs[:, 6].min():.3f} gripped {obs[:, 7].mean():.2f} episodes ended: {wins} by success, {fails} by imbalance / lost grip / timeout")
    print(f"{'resting' if args.rest else 'following'} robot: {wins} successes, {fails} failures in {args.steps} steps x {n} envs; board gripped {np.mean(held[10:]):.0%} of the time")
    env.close()


if __name__ == "__main__":
    main()
