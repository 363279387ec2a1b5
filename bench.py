#!/usr/bin/env python3
"""bench.py — env (policy) steps/sec of the batched ReachHuman stepper, one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric): ReachHuman, sara-shield SSM, 4096 envs per GPU, control_freq 10 (25 shield
cycles of 4 ms per policy step), synthetic random actions U(-1,1)^7, synthetic human clips, auto-reset on.
One "step" = one `hrg_batch_step` launch over the rank's 4096 envs (+ one RCCL all-gather of the packed
outputs when N > 1).  Envs shard embarrassingly: rank r owns global env ids [r*4096, (r+1)*4096).

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     — HBM roofline of the step kernel: ALGORITHMIC bytes per launch / average kernel time measured
                 live with HIP events on the launch stream (hrg_batch_kernel_time).
  cpu_baseline — the CPU oracle (oracle/hrg_oracle.c, kind "port") timed on the host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes_per_env_step(C, state_bytes, n_cycles):
    """state block read + written once, one animation frame per cycle, action in, outputs out."""
    frame = C["HRG_FRAME_DIM"] * 8
    out = 2 * C["HRG_OBS_DIM"] * 4 + 4 + 1 + C["HRG_INFO_DIM"] * 4
    return 2 * state_bytes + n_cycles * frame + C["HRG_ACT_DIM"] * 8 + out


def cpu_baseline(env_kwargs, clips_seed, budget_s=12.0, max_threads=16, env_id="ReachHuman", n_envs=ENVS_PER_GPU, wrappers=None):
    """Oracle on the host cores: P threads (ctypes releases the GIL) x n/P envs each, barrier per vec-step —
    the shape of the reference's SubprocVecEnv (one worker per core, synchronised once per step)."""
    import numpy as np
    import human_robot_gym_amd as hrg
    from oracle.oracle import OracleBatch
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(avail, max_threads)  # a 1-GPU box is given a 16-core CPU share
    n = n_envs
    clips = _bench_clips(env_id, clips_seed)
    desc = hrg.build_model_desc(env_kwargs, n_clips=clips.n_clips, env_id=env_id, **(wrappers or {}))
    B = OracleBatch(desc, clips, n, 0)
    B.reset()
    rng = np.random.RandomState(1234)
    bounds = [(n * i // cores, n * (i + 1) // cores) for i in range(cores)]

    def vec_step(a):
        ts = [threading.Thread(target=B.step_range, args=(lo, hi, a)) for lo, hi in bounds]
        for t in ts:
            t.start()
        for t in ts:
            t.join()

    def draw():
        a = rng.uniform(-1, 1, (n, 7))
        if wrappers:
            a[:, :3] *= 0.15
        return a

    vec_step(draw())  # warm-up
    t0 = time.perf_counter()
    k = 0
    while True:
        vec_step(draw())
        k += 1
        el = time.perf_counter() - t0
        if el >= budget_s and k >= 2:
            break
    B.close()
    return {"value": n * k / el, "unit": "env steps/s", "cores": cores, "kind": "port",
            "sample": f"{n} envs x {k} vec-steps ({el:.1f} s), oracle/hrg_oracle.c on {cores} of {avail} host threads"}


OTHER_TASKS = {  # --env values beyond the two benchmark configurations: (env kwargs, kernel) per training/icra_2024_run_experiments.sh:4-9
    "HumanObjectInspectionCart": (dict(horizon=1000), "hrg_step_kernel_box"),
    "HumanRobotHandoverCart": (dict(horizon=1000, shield_type="PFL"), "hrg_step_kernel_ho"),
    "RobotHumanHandoverCart": (dict(horizon=1000, shield_type="PFL"), "hrg_step_kernel_ho"),
    "CollaborativeLiftingCart": (dict(horizon=5000), "hrg_step_kernel_lift"),
}


def _bench_clips(env_id, seed=0):
    """13 synthetic clips; the collaboration tasks get the animation info they read (mixed.task_clips)."""
    import human_robot_gym_amd as hrg
    if env_id in OTHER_TASKS:
        from human_robot_gym_amd.mixed import task_clips
        return task_clips(env_id, 13, seed=seed)
    return hrg.synthetic_clips(13, seed=seed)


class _stdout_to_stderr:
    """RCCL prints a version banner on stdout when its communicator comes up; the contract is ONE JSON line on stdout, so the file
    descriptor is pointed at stderr while the process group initialises and the warm-up steps run."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--shield", default="SSM", choices=["SSM", "OFF"])
    ap.add_argument("--env", default="ReachHuman", choices=["ReachHuman", "PickPlaceHumanCart", "mixed"] + sorted(OTHER_TASKS),
                    help="ReachHuman = the configuration BASELINE.json's metric is quoted on (default); PickPlaceHumanCart = its config 4 (8192 envs); "
                         "the other tasks of the ICRA suite at 4096 envs (--shield is overridden by the suite's shield type where it names one); "
                         "mixed = BASELINE configs[4]: 4096 envs per GPU split evenly over the suite's tasks that are built (mixed.ICRA_TASKS)")
    ap.add_argument("--ik", action="store_true", help="Cartesian actions [dx,dy,dz,gripper] through the in-kernel IK front-end "
                    "(config/wrappers/safe_ik.yaml: IKPositionDeltaWrapper + CollisionPreventionWrapper), as the reference trains pick-place")
    ap.add_argument("--envs-per-gpu", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "r01_pmc_traffic.json"))
    args = ap.parse_args()

    import numpy as np
    import torch
    import human_robot_gym_amd as hrg
    from human_robot_gym_amd._cstruct import CONST as C
    from human_robot_gym_amd._lib import HipBatch, load_library
    from human_robot_gym_amd.dist import OverlappedGather

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dist = None
    quiet = _stdout_to_stderr()
    quiet.__enter__()   # until the warm-up is over (RCCL banner)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # ReachHuman training configuration: training/config/environment/reach_human.yaml + human_reach_ppo_parallel.yaml
    pick_place = args.env != "ReachHuman"   # every other task carries the manipulation object's state block
    if args.env in OTHER_TASKS:
        env_kwargs = dict(shield_type=args.shield, control_freq=10, seed=1234)
        env_kwargs.update(OTHER_TASKS[args.env][0])
        args.shield = env_kwargs["shield_type"]
    elif pick_place:  # training/config/environment/pick_place_human_cart.yaml
        env_kwargs = dict(shield_type=args.shield, control_freq=10, horizon=1000, done_at_success=False, goal_dist=0.1,
                          reward_shaping=False, collision_reward=0, object_gripped_reward=-0.25, seed=1234)
    else:
        env_kwargs = dict(shield_type=args.shield, control_freq=10, horizon=100, done_at_success=True, goal_dist=0.1,
                          reward_shaping=True, collision_reward=0, safe_vel=0.01, seed=1234)
    n = args.envs_per_gpu or (8192 if args.env == "PickPlaceHumanCart" else ENVS_PER_GPU)
    wrappers = dict(ik_position_delta=dict(action_limit=0.15), collision_prevention=dict(replace_type=0, n_resamples=20)) if args.ik else {}
    mixed_tasks = None
    if args.env == "mixed":  # one HipBatch per task on its own stream, one packed output block (human_robot_gym_amd/mixed.py)
        from human_robot_gym_amd import mixed
        if args.ik:
            raise SystemExit("--ik: the mixed batch takes joint-space actions")
        mixed_tasks = [t[0] for t in mixed.ICRA_TASKS]
        G = mixed.make_mixed_batch(n, seed=1234, env_id0=rank * n, device=local_rank)
        desc = hrg.build_model_desc(dict(seed=1234), env_id="PickPlaceHumanCart")   # (substeps per step and the horizon field of the report only)
        args.shield = "per task"
    else:
        clips = _bench_clips(args.env, 0)
        desc = hrg.build_model_desc(env_kwargs, n_clips=clips.n_clips, env_id=args.env, **wrappers)
        G = HipBatch(desc, clips, n, env_id0=rank * n, device=local_rank)
    dev = G.device
    G.reset()
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    pool = [torch.rand((n, C["HRG_ACT_DIM"]), generator=gen, device=dev, dtype=torch.float64) * 2 - 1 for _ in range(32)]
    if args.ik:  # position deltas U(-0.15, 0.15)^3, gripper U(-1, 1)
        for a in pool:
            a[:, :3] *= 0.15
    fresh = [torch.empty_like(pool[0]) for _ in range(2)]  # the kernel rewrites action rows in place when wrappers are on
    # N > 1: every rank's packed outputs are published to all ranks with one RCCL all-gather per step on the compute stream.
    # HRG_BENCH_GATHER_MODE=overlap moves it to a side stream (dist.OverlappedGather); measured on one MI355X that is SLOWER: the
    # step kernel fills every workgroup slot of the chip (4096 = 256 CUs x 16), so a concurrent copy / RCCL kernel pushes part of it
    # into a second round (2.31 instead of 1.96 ms).  HRG_BENCH_FORCE_GATHER=1 rehearses the gather path on one GPU (world size 1)
    force = os.environ.get("HRG_BENCH_FORCE_GATHER") == "1"
    if force and world == 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
    serial = os.environ.get("HRG_BENCH_GATHER_MODE", "serial") != "overlap"
    gather = (True if serial else OverlappedGather(G.packed_head, world)) if (world > 1 or force) else None
    gathered = torch.empty(world * G.packed_head.numel(), dtype=torch.uint8, device=dev) if (gather is not None and serial) else None

    def one_step(k):
        if args.ik:
            a = fresh[k & 1]
            a.copy_(pool[k % len(pool)])
            G.step(a)
        else:
            G.step(pool[k % len(pool)])
        if gather is not None and serial:
            dist.all_gather_into_tensor(gathered, G.packed_head)  # obs, reward, info, done of every rank (1.0 MB per rank)
        elif gather is not None:
            gather.publish(G.packed_head, k)  # one fused RCCL all-gather of obs/reward/done/info on a side stream

    for k in range(args.warmup):
        one_step(k)
    if gather is not None and args.warmup == 0:
        one_step(0)      # the first collective brings the communicator up: never inside the timed region
    torch.cuda.synchronize()
    quiet.__exit__(None, None, None)
    G.kernel_time()  # arm + clear the HIP-event kernel timer
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(k)
    if gather is not None and not serial:
        gather.finish()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms, n_launch = G.kernel_time()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        state_bytes = load_library().hrg_state_bytes() + (load_library().hrg_box_bytes() if pick_place else 0)
        per_env = algorithmic_bytes_per_env_step(C, state_bytes, desc.n_cycles)
        if mixed_tasks:  # ReachHuman's share moves no object block; the kernels overlap, so the launch duration is the step's wall time
            n_reach = G.slices[G.env_ids.index("ReachHuman")].stop - G.slices[G.env_ids.index("ReachHuman")].start if "ReachHuman" in G.env_ids else 0
            per_env = (per_env * n - n_reach * 2 * load_library().hrg_box_bytes()) / n
            kernel_ms = 1e3 * elapsed / args.steps
        achieved = per_env * n / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        traffic = None
        try:
            with open(args.traffic_json) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        except Exception:
            pass
        if pick_place or n != ENVS_PER_GPU:
            traffic = None  # the committed PMC capture is of the default workload's kernel
        out = {
            "metric": "env steps/sec (whole node), ReachHuman+shield 4096 envs" if not pick_place else (
                f"env steps/sec (whole node), mixed ICRA task batch {n} envs/GPU" if mixed_tasks else f"env steps/sec (whole node), {args.env}+shield {n} envs"),
            "value": world * n * args.steps / elapsed,
            "unit": "env steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.env}, {n} envs/GPU, sara-shield {args.shield}, control_freq 10 (25 x 4 ms shield cycles per step), "
                                   + ("Cartesian random actions through the IK front-end + collision prevention, " if args.ik else "random actions U(-1,1)^7, ")
                                   + "13 synthetic human clips, auto-reset",
                       "envs_per_gpu": n, "shield_type": args.shield, "horizon": int(desc.horizon), "substeps_per_step": int(desc.n_cycles),
                       "parallelism": f"env-sharded x{world}" + (", 1 RCCL all-gather/step" + ("" if serial else " on a side stream") if gather is not None else "")},
            "substeps_per_s": world * n * args.steps * int(desc.n_cycles) / elapsed,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "all step kernels, concurrent (wall time per step)" if mixed_tasks else (
                             OTHER_TASKS[args.env][1] if args.env in OTHER_TASKS else ("hrg_step_kernel_box" if pick_place else "hrg_step_kernel")), "kernel_ms": kernel_ms, "launches": n_launch,
                         "algorithmic_bytes_per_launch": per_env * n},
        }
        if world == 1 and not args.no_cpu_baseline:
            if mixed_tasks:
                out["config"]["tasks"] = mixed_tasks
            else:
                out["cpu_baseline"] = cpu_baseline(env_kwargs, 0, max_threads=args.cpu_threads, env_id=args.env, n_envs=n, wrappers=wrappers)
        print(json.dumps(out), flush=True)
    G.close()
    if world > 1:
        dist.barrier()
    if world > 1 or force:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
