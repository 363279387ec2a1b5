"""Throughput of a mixed-task batch (BASELINE.json configs[4] over the tasks built so far): n envs split evenly over
mixed.ICRA_TASKS, per-task kernels on their own streams (concurrent) vs one after the other (serial), and every task alone
at its share.  Prints one JSON line.  usage: python tools/bench_mixed.py [--envs 4096] [--steps 100] [--warmup 10]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(step, steps, warmup, torch):
    for k in range(warmup):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    args = ap.parse_args()
    import torch
    from human_robot_gym_amd import mixed
    n = args.envs
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1234)
    pool = [torch.rand((n, 7), generator=gen, device="cuda", dtype=torch.float64) * 2 - 1 for _ in range(16)]
    out = {"workload": f"{n} envs split evenly over {[t[0] for t in mixed.ICRA_TASKS]}, random joint-space actions, 13 synthetic clips per task",
           "n_envs": n, "steps": args.steps}
    for mode in ("concurrent", "serial"):
        M = mixed.make_mixed_batch(n, seed=1234, concurrent=mode == "concurrent")
        M.reset()
        dt = timed(lambda k: M.step(pool[k % 16]), args.steps, args.warmup, torch)
        out[mode] = {"ms_per_step": 1e3 * dt, "env_steps_per_s": n / dt}
        if mode == "serial":
            per = {}
            for b, eid, sl in zip(M.batches, M.env_ids, M.slices):
                acts = [p[sl].contiguous() for p in pool]
                dt1 = timed(lambda k: b.step(acts[k % 16]), args.steps, args.warmup, torch)
                per[eid] = {"n_envs": sl.stop - sl.start, "ms_per_step": 1e3 * dt1}
            out["alone"] = per
        M.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
