#!/usr/bin/env python3
"""bench.py — env (policy) steps/sec of the batched human-robot-gym stepper, one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python bench.py --gpus N ...          (no WORLD_SIZE in the env: this process spawns the N ranks itself and never touches a GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric): ReachHuman, sara-shield SSM, 4096 envs per GPU, control_freq 10 (25 shield
cycles of 4 ms per policy step), synthetic random actions U(-1,1)^7, synthetic human clips, auto-reset on.
One "step" = one `hrg_batch_step` launch over the rank's 4096 envs (+ one RCCL all-gather of the packed
outputs when N > 1).  Envs shard embarrassingly: rank r owns global env ids [r*4096, (r+1)*4096).

STEADY STATE.  Before the W warm-up steps the batch is rolled, untimed, for `preroll` steps (one horizon, at most 1000:
`config.preroll_steps`), and the TimeLimit phase of the envs is staggered over the horizon once after the reset
(`config.episode_phases_staggered`: ~n / horizon envs end their episode in every step instead of all of them every
`horizon` steps), whatever --warmup says; a batch timed right after reset is ~25 % faster than the one a training run
sees (fewer envs braking or in contact, no auto-resets).

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     — HBM roofline of the step kernel: ALGORITHMIC bytes per launch / average kernel time measured
                 live with HIP events on the launch stream (hrg_batch_kernel_time).  `traffic`, `valu_*` and `fp64_*`
                 are PMC figures of a committed capture of this same command (`traffic_source` names the file): a
                 counter pass slows the kernel and cannot run inside the timed region.
  cpu_baseline — the CPU oracle (oracle/hrg_oracle.c, kind "port") on the host: P worker threads, barrier per vec-step
                 (the SubprocVecEnv shape), envs of a vec-step drawn dynamically; several legs (all physical cores pinned,
                 16 threads, the cgroup's cpu quota), `value` = the best, each with worker-busy and throttling figures.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
FP64_VECTOR_PEAK_TF = 78.6  # MI355X FP64 vector peak
DEFAULT_PMC = os.path.join(ROOT, "profiles", "r03_pmc.json")
DEFAULT_PMC_PP = os.path.join(ROOT, "profiles", "r03pp_pmc.json")   # the committed capture of `--env PickPlaceHumanCart` (8192 envs, SSM)


def algorithmic_bytes_per_env_step(C, state_bytes, n_cycles):
    """state block read + written once, one animation frame per cycle, action in, outputs out."""
    frame = C["HRG_FRAME_DIM"] * 8
    out = 2 * C["HRG_OBS_DIM"] * 4 + 4 + 1 + C["HRG_INFO_DIM"] * 4
    return 2 * state_bytes + n_cycles * frame + C["HRG_ACT_DIM"] * 8 + out


def physical_cores():
    """(logical cpus of this process's affinity mask, one logical cpu per physical core of that mask)."""
    avail = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    seen, firsts = set(), []
    for c in avail:
        key = None
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as f:
                key = f.read().strip()
        except OSError:
            key = str(c)
        if key not in seen:
            seen.add(key)
            firsts.append(c)
    return avail, firsts


def cgroup_cpu_limit():
    """(cpus the cgroup may use per period or None, path read): cgroup v2 `cpu.max` / v1 `cpu.cfs_quota_us`.  The affinity mask of a container usually shows every
    logical cpu of the host while the CFS quota caps the cpu TIME it gets: threads beyond the quota do not add throughput, they are throttled."""
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, per = open(path).read().split()[:2]
            return (None if q == "max" else float(q) / float(per)), path
        except (OSError, ValueError):
            pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return (None if q <= 0 else q / per), "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"
    except (OSError, ValueError):
        return None, None


def _throttled_usec():
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            d = dict(l.split()[:2] for l in open(path).read().splitlines() if l.strip())
            if "throttled_usec" in d:
                return float(d["throttled_usec"])
            if "throttled_time" in d:
                return float(d["throttled_time"]) / 1e3
        except (OSError, ValueError):
            pass
    return None


def _sockets(cpus):
    seen = set()
    for c in cpus:
        try:
            seen.add(open(f"/sys/devices/system/cpu/cpu{c}/topology/physical_package_id").read().strip())
        except OSError:
            pass
    return len(seen) or None


def cpu_baseline(env_kwargs, clips_seed, budget_s=10.0, env_id="ReachHuman", n_envs=ENVS_PER_GPU, wrappers=None, preroll=100):
    """Oracle on the host cores, the shape of the reference's SubprocVecEnv (P workers, synchronised once per vec-step): P pthreads inside the oracle library,
    pthread barrier per vec-step.  Round 3: the envs of a vec-step are drawn by the workers in runs of 8 from a shared counter (the per-env cost depends on its
    state, and a pinned worker of a shared host may lose its core: with fixed env ranges every vec-step waited for its unluckiest worker -- round 2's 128 pinned
    threads were SLOWER than 16), and the legs are chosen with the cgroup's cpu quota in view.  Legs: every physical core of the affinity mask (pinned), 16 threads
    (the CPU share gpurun documents for a 1-GPU box), and the quota's own thread count when it differs.  `value` = the best leg; every leg reports its throughput,
    how busy its workers were (seconds inside env_step / wall seconds, mean and min over the workers) and the time the cgroup spent throttled during it."""
    import math
    import numpy as np
    import human_robot_gym_amd as hrg
    from oracle.oracle import OracleBatch
    avail, firsts = physical_cores()
    quota, quota_src = cgroup_cpu_limit()
    n = n_envs
    clips = _bench_clips(env_id, clips_seed)
    desc = hrg.build_model_desc(env_kwargs, n_clips=clips.n_clips, env_id=env_id, **(wrappers or {}))
    B = OracleBatch(desc, clips, n, 0)
    B.reset()
    rng = np.random.RandomState(1234)
    pool = rng.uniform(-1, 1, (16, n, 7))
    if wrappers:
        pool[:, :, :3] *= 0.15

    def timed(workers, cpus, budget, chunk=8):
        B.rollout_parallel2(pool, 2, workers, cpus, chunk)   # warm-up of the thread team / caches
        th0 = _throttled_usec()
        t0 = time.perf_counter()
        k, per_call, busy = 0, 8, np.zeros(workers)
        while True:
            busy += B.rollout_parallel2(pool, per_call, workers, cpus, chunk)
            k += per_call
            el = time.perf_counter() - t0
            if el >= budget:
                break
        th1 = _throttled_usec()
        return {"threads": workers, "pinned": cpus is not None, "value": n * k / el, "vec_steps": k, "seconds": round(el, 2),
                "worker_busy_mean": round(float(busy.mean() / el), 3), "worker_busy_min": round(float(busy.min() / el), 3),
                "cgroup_throttled_s": None if th0 is None or th1 is None else round((th1 - th0) / 1e6, 2)}

    # the same pre-roll as the GPU leg, so that both time the steady state (bounded: the CPU is ~20-50x slower)
    p16 = min(16, len(avail))
    B.rollout_parallel2(pool, min(preroll, 24), min(len(firsts), 32), None, 8)
    legs = {"all_physical_cores_pinned": timed(len(firsts), firsts, 0.5 * budget_s)}
    legs["16_threads"] = timed(p16, None, 0.5 * budget_s)
    if quota is not None and int(math.ceil(quota)) not in (len(firsts), p16):
        pq = max(1, min(len(avail), int(math.ceil(quota))))
        legs["cgroup_quota_threads"] = timed(pq, None, 0.5 * budget_s)
    legs["16_threads_fixed_env_ranges"] = timed(p16, None, 0.3 * budget_s, chunk=0)   # round 2's partition, for the comparison
    B.close()
    best = max((k for k in legs if k != "16_threads_fixed_env_ranges"), key=lambda k: legs[k]["value"])
    L = legs[best]
    return {"value": L["value"], "unit": "env steps/s", "cores": L["threads"], "kind": "port",
            "sample": f"{n} envs x {L['vec_steps']} vec-steps ({L['seconds']} s), oracle/hrg_oracle.c, leg '{best}': {L['threads']} pthreads"
                      f"{' pinned one per physical core' if L['pinned'] else ''}, envs of a vec-step drawn in runs of 8, barrier per vec-step",
            "host": {"logical_cpus_in_affinity_mask": len(avail), "physical_cores": len(firsts), "sockets": _sockets(avail), "cgroup_cpu_quota": quota,
                     "cgroup_cpu_quota_source": quota_src, "loadavg_1min": round(os.getloadavg()[0], 1)},
            "legs": legs,
            "value_16_threads": legs["16_threads"]["value"]}


OTHER_TASKS = {  # --env values beyond the two benchmark configurations: (env kwargs, kernel) per training/icra_2024_run_experiments.sh:4-9
    "HumanObjectInspectionCart": (dict(horizon=1000), "hrg_step_kernel_box"),
    "HumanRobotHandoverCart": (dict(horizon=1000, shield_type="PFL"), "hrg_step_kernel_ho"),
    "RobotHumanHandoverCart": (dict(horizon=1000, shield_type="PFL"), "hrg_step_kernel_ho"),
    "CollaborativeLiftingCart": (dict(horizon=5000), "hrg_step_kernel_lift"),
    "CollaborativeStackingCart": (dict(horizon=3000), "hrg_step_kernel_stack"),
    "CollaborativeHammeringCart": (dict(horizon=1000), "hrg_step_kernel_hammer"),
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


def make_gather(G, world, mode="serial"):
    """The N > 1 exchange of one step: every rank's packed head (obs, reward, info, done) to all ranks with ONE RCCL all-gather.
    Returns (publish(k), finish(), gathered tensor or None).  Factored out so that tests/test_api_gpu.py can run it in-process."""
    import torch
    import torch.distributed as dist
    from human_robot_gym_amd.dist import OverlappedGather
    if mode == "overlap":   # side stream; measured slower on one GPU: kept as an experiment
        og = OverlappedGather(G.packed_head, world)
        return (lambda k: og.publish(G.packed_head, k)), og.finish, None
    gathered = torch.empty(world * G.packed_head.numel(), dtype=torch.uint8, device=G.device)
    return (lambda k: dist.all_gather_into_tensor(gathered, G.packed_head)), (lambda: None), gathered


def spawn_ranks(n, argv, script=None, deadline_s=3000.0, poll_s=0.2):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in their env).
    This parent never initialises torch.cuda or HIP and never execs; rank 0's stdout (the one JSON line) is passed through, the other ranks'
    stdout is dropped (their stderr stays visible).  The children are polled: the first one that exits non-zero -- a HIP error, an import error, an
    assert before or inside init_process_group -- ends the run: its siblings, which would otherwise sit in the RCCL rendezvous or a collective holding
    their GPUs, are terminated (killed after 5 s), and that exit code is returned.  `deadline_s` bounds the whole run the same way (exit code 124).
    Nothing is restarted.  `script`: the program the ranks run (default: this file); tests pass a stub."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = os.path.abspath(script or __file__)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env, stdout=None if r == 0 else subprocess.DEVNULL))

    def stop_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    t_dead = time.monotonic() + deadline_s
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                r, c = bad[0]
                print(f"bench.py: rank {r} exited with code {c}; stopping the other ranks", file=sys.stderr)
                stop_all()
                return abs(c) if abs(c) < 256 else 1
            if all(c == 0 for c in codes):
                return 0
            if time.monotonic() > t_dead:
                print(f"bench.py: the {n} ranks did not finish within {deadline_s:.0f} s; stopping them", file=sys.stderr)
                stop_all()
                return 124
            time.sleep(poll_s)
    except BaseException:
        stop_all()
        raise


def bench_workload(env="ReachHuman", shield="SSM", ik=False, envs_per_gpu=None, robot_geometry="capsule", collision_prevention=False):
    """The benchmark's workloads by name (BASELINE.json configs): env kwargs as the reference's training configs set them, envs per GPU, wrappers.
    tests/test_bench_state_gpu.py builds its batches through this function and `make_bench_batch`, so what it compares with the oracle is what is timed."""
    pick_place = env != "ReachHuman"   # every other task carries the manipulation object's state block
    if env == "mixed":
        if ik:
            raise SystemExit("--ik: the mixed batch takes joint-space actions")
        env_kwargs, shield = dict(seed=1234), "per task"
    elif env in OTHER_TASKS:
        env_kwargs = dict(shield_type=shield, control_freq=10, seed=1234)
        env_kwargs.update(OTHER_TASKS[env][0])
        from human_robot_gym_amd.mixed import task_env_kwargs
        env_kwargs.update(task_env_kwargs(env))   # what goes with the task's synthetic clips (hammering: the weld pose that holds the board level)
        shield = env_kwargs["shield_type"]
    elif pick_place:  # training/config/environment/pick_place_human_cart.yaml
        env_kwargs = dict(shield_type=shield, control_freq=10, horizon=1000, done_at_success=False, goal_dist=0.1,
                          reward_shaping=False, collision_reward=0, object_gripped_reward=-0.25, seed=1234)
    else:             # training/config/environment/reach_human.yaml + human_reach_ppo_parallel.yaml
        env_kwargs = dict(shield_type=shield, control_freq=10, horizon=100, done_at_success=True, goal_dist=0.1,
                          reward_shaping=True, collision_reward=0, safe_vel=0.01, seed=1234)
    n = envs_per_gpu or (8192 if env == "PickPlaceHumanCart" else ENVS_PER_GPU)
    wrappers = dict(ik_position_delta=dict(action_limit=0.15), collision_prevention=dict(replace_type=0, n_resamples=20)) if ik else {}
    if collision_prevention and not ik:   # training/config/wrappers/safe.yaml (the `wrappers: safe` of human_reach_ppo_parallel.yaml): CollisionPreventionWrapper, replace_type 0, 20 resamples
        wrappers["collision_prevention"] = dict(replace_type=0, n_resamples=20)
    if robot_geometry != "capsule":   # arm links as the convex hulls of their meshes: passed on to build_model_desc like the wrappers
        wrappers["robot_geometry"] = robot_geometry
    return dict(env=env, shield=shield, ik=bool(ik), n=n, env_kwargs=env_kwargs, wrappers=wrappers, pick_place=pick_place, robot_geometry=robot_geometry)


def make_bench_batch(W, rank=0, local_rank=0, stagger=True):
    """(batch, model desc, mixed task list or None, staggered) of a `bench_workload`: reset, TimeLimit phases staggered over the horizon."""
    import human_robot_gym_amd as hrg
    from human_robot_gym_amd._lib import HipBatch
    n, env = W["n"], W["env"]
    mixed_tasks = None
    if env == "mixed":  # one HipBatch per task on its own stream, one packed output block (human_robot_gym_amd/mixed.py)
        from human_robot_gym_amd import mixed
        mixed_tasks = [t[0] for t in mixed.ICRA_TASKS]
        G = mixed.make_mixed_batch(n, seed=1234, env_id0=rank * n, device=local_rank)
        desc = hrg.build_model_desc(dict(seed=1234), env_id="PickPlaceHumanCart")   # (substeps per step and the horizon field of the report only)
    else:
        clips = _bench_clips(env, 0)
        desc = hrg.build_model_desc(W["env_kwargs"], n_clips=clips.n_clips, env_id=env, **W["wrappers"])
        G = HipBatch(desc, clips, n, env_id0=rank * n, device=local_rank)
    G.reset()
    staggered = False
    if not mixed_tasks and stagger:   # spread the TimeLimit phase: ~n / horizon envs time out in every step instead of all of them every `horizon` steps
        G.stagger_episode_phases(int(desc.horizon))
        staggered = True
    return G, desc, mixed_tasks, staggered


def bench_action_pool(n, dev, rank=0, ik=False, n_pool=32):
    """The benchmark's synthetic actions: a pool of U(-1, 1)^7 batches on the device, cycled step by step."""
    import torch
    from human_robot_gym_amd._cstruct import CONST as C
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    pool = [torch.rand((n, C["HRG_ACT_DIM"]), generator=gen, device=dev, dtype=torch.float64) * 2 - 1 for _ in range(n_pool)]
    if ik:  # position deltas U(-0.15, 0.15)^3, gripper U(-1, 1)
        for a in pool:
            a[:, :3] *= 0.15
    return pool


def bench_preroll_steps(desc, preroll=None):
    """Untimed steps before the warm-up: one horizon, at most 1000."""
    return preroll if preroll is not None else min(int(desc.horizon), 1000)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--preroll", type=int, default=None, help="untimed steps before the warm-up that spread the episode phases (default: one horizon, at most 1000)")
    ap.add_argument("--shield", default="SSM", choices=["SSM", "OFF"])
    ap.add_argument("--env", default="ReachHuman", choices=["ReachHuman", "PickPlaceHumanCart", "mixed"] + sorted(OTHER_TASKS),
                    help="ReachHuman = the configuration BASELINE.json's metric is quoted on (default); PickPlaceHumanCart = its config 4 (8192 envs); "
                         "the other tasks of the ICRA suite at 4096 envs (--shield is overridden by the suite's shield type where it names one); "
                         "mixed = BASELINE configs[4]: 4096 envs per GPU split evenly over the suite's tasks (mixed.ICRA_TASKS)")
    ap.add_argument("--ik", action="store_true", help="Cartesian actions [dx,dy,dz,gripper] through the in-kernel IK front-end "
                    "(config/wrappers/safe_ik.yaml: IKPositionDeltaWrapper + CollisionPreventionWrapper), as the reference trains pick-place")
    ap.add_argument("--envs-per-gpu", type=int, default=None)
    ap.add_argument("--collision-prevention", action="store_true", help="joint-space actions screened by the CollisionPreventionWrapper in the kernel prologue (config/wrappers/safe.yaml: "
                    "replace_type 0, 20 resamples), as human_reach_ppo_parallel.yaml trains; named in config.workload")
    ap.add_argument("--robot-geometry", default="capsule", choices=["capsule", "hull"], help="collision geometry of the arm links: bounding capsules (default) or the convex "
                    "hulls of the link meshes (ReachHuman only; DESIGN.md D3); named in config.robot_geometry")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=10.0, help="seconds of CPU work for the all-cores leg of cpu_baseline (the 16-thread leg gets 0.6 x)")
    ap.add_argument("--pmc-json", default=DEFAULT_PMC, help="committed PMC capture of this command (tools/profile_capture.sh): source of roofline.traffic / valu_* / fp64_*")
    ap.add_argument("--variant-lib", default=None, help="tuning experiments only: time another build of the library; echoed as `variant_lib` in the JSON line")
    ap.add_argument("--no-stagger", action="store_true", help="leave every env at episode step 0 after the reset (all of them then time out in the same steps)")
    ap.add_argument("--force-gather", action="store_true", help="rehearse the N > 1 all-gather on one GPU (world size 1)")
    ap.add_argument("--gather-mode", default="serial", choices=["serial", "overlap"])
    args = ap.parse_args()

    for var in ("HRG_PHASE_MASK", "HRG_LIB_PATH"):   # round-1 tuning switches: refuse rather than print a number that means something else
        if os.environ.get(var):
            print(f"bench.py: {var} is set; unset it (a variant build is timed with --variant-lib, and is reported as such)", file=sys.stderr)
            raise SystemExit(2)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    import numpy as np  # noqa: F401
    import torch
    import human_robot_gym_amd as hrg
    from human_robot_gym_amd import _lib
    from human_robot_gym_amd._cstruct import CONST as C
    if args.variant_lib:
        _lib.use_variant_library(args.variant_lib)
    from human_robot_gym_amd._lib import HipBatch, load_library

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dist = None
    quiet = _stdout_to_stderr()
    quiet.__enter__()   # until the warm-up is over (RCCL banner)
    force = args.force_gather or os.environ.get("HRG_BENCH_FORCE_GATHER") == "1"
    if world > 1 or force:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    W = bench_workload(args.env, args.shield, args.ik, args.envs_per_gpu, args.robot_geometry, args.collision_prevention)
    args.shield, n, env_kwargs, wrappers, pick_place = W["shield"], W["n"], W["env_kwargs"], W["wrappers"], W["pick_place"]
    G, desc, mixed_tasks, staggered = make_bench_batch(W, rank=rank, local_rank=local_rank, stagger=not args.no_stagger)
    dev = G.device
    pool = bench_action_pool(n, dev, rank, args.ik)
    fresh = [torch.empty_like(pool[0]) for _ in range(2)]  # the kernel rewrites action rows in place when wrappers are on
    # N > 1: every rank's packed outputs are published to all ranks with one RCCL all-gather per step on the compute stream.
    publish = finish = None
    if world > 1 or force:
        publish, finish, _ = make_gather(G, world, args.gather_mode)

    rewrites_actions = args.ik or args.collision_prevention
    def one_step(k, exchange=True):
        if rewrites_actions:
            a = fresh[k & 1]
            a.copy_(pool[k % len(pool)])
            G.step(a)
        else:
            G.step(pool[k % len(pool)])
        if publish is not None and exchange:
            publish(k)

    # pre-roll: spread the episode phases (a fresh batch has every env at timestep 0, nobody braking, no contacts yet)
    preroll = bench_preroll_steps(desc, args.preroll)
    for k in range(preroll):
        one_step(k, exchange=False)
    for k in range(args.warmup):
        one_step(preroll + k)
    if publish is not None and args.warmup == 0:
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
    if finish is not None:
        finish()
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
        lib = load_library()
        obj_bytes = lib.hrg_stack_bytes() if args.env == "CollaborativeStackingCart" else (lib.hrg_hammer_bytes() if args.env == "CollaborativeHammeringCart" else (lib.hrg_box_bytes() if pick_place else 0))   # the task's object block, streamed next to the env block
        state_bytes = lib.hrg_state_bytes() + obj_bytes
        per_env = algorithmic_bytes_per_env_step(C, state_bytes, desc.n_cycles)
        if mixed_tasks:  # each task streams its own object block; the kernels overlap, so the launch duration is the step's wall time
            tot = 0
            for env_id, sl in zip(G.env_ids, G.slices):
                ob = 0 if env_id == "ReachHuman" else (lib.hrg_stack_bytes() if env_id == "CollaborativeStackingCart" else (lib.hrg_hammer_bytes() if env_id == "CollaborativeHammeringCart" else lib.hrg_box_bytes()))
                tot += (sl.stop - sl.start) * algorithmic_bytes_per_env_step(C, lib.hrg_state_bytes() + ob, desc.n_cycles)
            per_env = tot / n
            kernel_ms = 1e3 * elapsed / args.steps
        achieved = per_env * n / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        kernel_name = "all step kernels, concurrent (wall time per step)" if mixed_tasks else (
            OTHER_TASKS[args.env][1] if args.env in OTHER_TASKS else ("hrg_step_kernel_box" if pick_place else ("hrg_step_kernel_hull" if args.robot_geometry == "hull" else "hrg_step_kernel")))
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "traffic_source": None, "kernel": kernel_name, "kernel_ms": kernel_ms, "launches": n_launch, "algorithmic_bytes_per_launch": per_env * n}
        # PMC figures: from the committed capture of the default workload's kernel (never from this run: a counter pass perturbs the timing)
        plain = args.shield == "SSM" and not args.ik and not args.variant_lib and args.robot_geometry == "capsule" and not args.collision_prevention
        default_workload = plain and args.env == "ReachHuman" and n == ENVS_PER_GPU
        pmc_path = args.pmc_json if default_workload else (DEFAULT_PMC_PP if plain and args.env == "PickPlaceHumanCart" and n == 8192 and args.pmc_json == DEFAULT_PMC else None)
        if pmc_path:
            try:
                with open(pmc_path) as f:
                    pmc = json.load(f)
                src = os.path.relpath(pmc_path, ROOT)
                roof["traffic"] = pmc.get("hbm_bytes_per_launch")
                roof["traffic_source"] = f"{src}: {pmc.get('command', 'rocprofv3 --pmc passes of this command')} (committed capture, per launch; not collected by this run)"
                for key in ("valu_insts_per_substep", "valu_busy_frac", "valu_lane_utilisation", "fp64_flops_per_launch"):
                    if key in pmc:
                        roof[key] = pmc[key]
                if "fp64_flops_per_launch" in pmc and kernel_ms > 0:
                    roof["fp64_tflops_est"] = pmc["fp64_flops_per_launch"] / (kernel_ms * 1e-3) / 1e12
                    roof["fp64_vector_peak_tflops"] = FP64_VECTOR_PEAK_TF
                    roof["fp64_frac_est"] = roof["fp64_tflops_est"] / FP64_VECTOR_PEAK_TF
            except Exception as ex:  # no capture committed yet: the fields stay null
                roof["traffic_source"] = f"unavailable ({type(ex).__name__})"
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
            "config": {"workload": f"{args.env}, {n} envs/GPU, shield {args.shield}, " + ("arm links as convex hulls, " if args.robot_geometry == "hull" else "") + "25 x 4 ms cycles/step, "
                                   + ("Cartesian random actions via IK + collision prevention" if args.ik else ("random actions U(-1,1)^7 screened by collision prevention (replace_type 0, 20 resamples)" if args.collision_prevention else "random actions U(-1,1)^7"))
                                   + f", 13 synthetic clips, auto-reset, steady state ({preroll}-step pre-roll)",
                       "envs_per_gpu": n, "shield_type": args.shield, "robot_geometry": args.robot_geometry, "horizon": int(desc.horizon), "substeps_per_step": int(desc.n_cycles),
                       "preroll_steps": preroll, "episode_phases_staggered": staggered,
                       "parallelism": f"env-sharded x{world}" + (", 1 RCCL all-gather/step" + ("" if args.gather_mode == "serial" else " on a side stream") if publish is not None else "")},
            "substeps_per_s": world * n * args.steps * int(desc.n_cycles) / elapsed,
            "roofline": roof,
        }
        if mixed_tasks:
            out["config"]["tasks"] = mixed_tasks
        if args.variant_lib:
            out["variant_lib"] = os.path.abspath(args.variant_lib)
        if world == 1 and not args.no_cpu_baseline and not mixed_tasks:
            out["cpu_baseline"] = cpu_baseline(env_kwargs, 0, budget_s=args.cpu_budget, env_id=args.env, n_envs=n, wrappers=wrappers, preroll=preroll)
        print(json.dumps(out), flush=True)
    G.close()
    if world > 1:
        dist.barrier()
    if world > 1 or force:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
