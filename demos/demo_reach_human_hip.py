#!/usr/bin/env python3
"""The reference's `demos/demo_reach_human_environment.py` loop on the HIP stepper: one ReachHuman env, SSM shield, random or scripted
joint-space actions, the same 4-tuple gym API.  Needs an MI355X (there is no CPU fallback).

    python demos/demo_reach_human_hip.py [--steps 200] [--env PickPlaceHumanCart|CollaborativeLiftingCart|...] [--cartesian]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from human_robot_gym_amd.vec_env import HipVecEnv  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    from human_robot_gym_amd.model import ENV_DEFAULTS
    ap.add_argument("--env", default="ReachHuman", choices=sorted(ENV_DEFAULTS))
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--n-envs", type=int, default=4)
    ap.add_argument("--cartesian", action="store_true", help="[dx, dy, dz, gripper] actions through the in-kernel IK (config/wrappers/safe_ik.yaml)")
    args = ap.parse_args()
    from human_robot_gym_amd.mixed import task_clips
    clips = task_clips(args.env, 4)   # synthetic clips carrying the animation info the task reads
    wrappers = dict(collision_prevention=dict(replace_type=0, n_resamples=20))
    if args.cartesian:
        wrappers["ik_position_delta"] = dict(action_limit=0.15)
    env = HipVecEnv(args.n_envs, env_id=args.env, env_kwargs=dict(shield_type="SSM", seed=0), clips=clips, **wrappers)
    obs = env.reset()
    rng = np.random.RandomState(0)
    ret = np.zeros(args.n_envs)
    for t in range(args.steps):
        a = rng.uniform(env.action_space.low, env.action_space.high, (args.n_envs,) + env.action_space.shape)
        obs, rew, done, infos = env.step(a)
        ret += rew
        if t % 20 == 0 or done.any():
            i = infos[0]
            print(f"t={t:4d} reward={rew[0]:+.3f} failsafe_interventions={i['failsafe_interventions']} collisions={i['n_collisions']} "
                  f"goals={i['n_goal_reached']} action_resamples={i['action_resamples']}" + ("  [episode end]" if done[0] else ""))
    print("returns:", np.round(ret, 2))
    env.close()


if __name__ == "__main__":
    main()
