#!/usr/bin/env python3
"""Condense rocprofv3 CSVs under gpurun_out/<tag>_* into profiles/<tag>_*.{md,json} (the tracked, judged evidence)."""
import csv, glob, json, os, sys, collections
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
os.makedirs(pr, exist_ok=True)
what = sys.argv[2] if len(sys.argv) > 2 else "`python3 bench.py --steps N --warmup W --no-cpu-baseline` (1 x MI355X, ReachHuman, 4096 envs, SSM)"
lines = [f"# rocprofv3 summary `{tag}` — {what}", ""]
f = sorted(glob.glob(f"{go}/{tag}_trace/*/*_kernel_stats.csv"), key=os.path.getmtime, reverse=True)
if f:
    lines += ["## `--kernel-trace --stats` (kernel_stats.csv)", "", "| kernel | calls | total ns | average ns | % | min ns | max ns |", "|---|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(f[0])):
        name = r["Name"].split("(")[0][:60]
        lines.append(f"| `{name}` | {r['Calls']} | {r['TotalDurationNs']} | {float(r['AverageNs']):.1f} | {float(r['Percentage']):.3f} | {r['MinNs']} | {r['MaxNs']} |")
    lines.append("")
pmc = {}
meta = {}
for sub in ["fetch", "write", "sq", "sq2", "sq3"]:
    f = sorted(glob.glob(f"{go}/{tag}_{sub}/*/*_counter_collection.csv"), key=os.path.getmtime, reverse=True)
    if not f:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "hrg_step_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = dict(VGPR_Count=r["VGPR_Count"], Accum_VGPR_Count=r.get("Accum_VGPR_Count"), SGPR_Count=r["SGPR_Count"], LDS_Block_Size=r["LDS_Block_Size"],
                        Scratch_Size=r["Scratch_Size"], Grid_Size=r["Grid_Size"], Workgroup_Size=r["Workgroup_Size"])
    for k, v in acc.items():
        pmc[k] = sum(v) / len(v)
if pmc:
    lines += ["## `--pmc` passes (separate runs; per-launch average over the hrg_step_kernel dispatches)", "", f"dispatch: {meta}", "", "| counter | per launch |", "|---|---|"]
    lines += [f"| {k} | {v:.1f} |" for k, v in sorted(pmc.items())]
    lines.append("")
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        # MI355X_MICROARCH.md §HBM: counters are in KiB; FETCH_SIZE reads half the bytes of a wide coalesced stream (x2 is the
        # prescribed correction for 16-B/lane streams; our 8-B/lane state stream is uncalibrated, so both figures are reported).
        fetch_raw, write = pmc["FETCH_SIZE"] * 1024, pmc["WRITE_SIZE"] * 1024
        traffic = dict(fetch_bytes_raw=fetch_raw, fetch_bytes_x2=2 * fetch_raw, write_bytes=write, hbm_bytes_per_launch=2 * fetch_raw + write,
                       note="FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE exact")
        traffic["command"] = ("rocprofv3 --pmc <counter set> --kernel-trace -- python3 bench.py --env PickPlaceHumanCart --steps 20 --warmup 3 --preroll 300 --no-cpu-baseline (separate passes: FETCH_SIZE, WRITE_SIZE, SQ; tools/profile_capture_pp.sh)"
                              if tag.endswith("pp") else
                              "rocprofv3 --pmc <counter set> --kernel-trace -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 (separate passes: FETCH_SIZE, WRITE_SIZE, SQ x3; tools/profile_capture.sh)")
        # vector-pipe figures for bench.py's roofline object
        n_env, n_cyc = int(meta.get("Grid_Size", 0)) // 64, 25
        if "SQ_INSTS_VALU" in pmc and n_env:
            traffic["valu_insts_per_substep"] = pmc["SQ_INSTS_VALU"] / (n_env * n_cyc)
        if "SQ_ACTIVE_INST_VALU" in pmc and "SQ_WAVE_CYCLES" in pmc:
            # SQ_WAVE_CYCLES sums over the resident waves (4 per SIMD at this occupancy): SIMD-cycles = WAVE_CYCLES / waves per SIMD
            wps = 4
            traffic["valu_busy_frac"] = pmc["SQ_ACTIVE_INST_VALU"] / (pmc["SQ_WAVE_CYCLES"] / wps)
            traffic["valu_busy_note"] = f"SQ_ACTIVE_INST_VALU / (SQ_WAVE_CYCLES / {wps} waves per SIMD)"
        if "SQ_THREAD_CYCLES_VALU" in pmc and "SQ_ACTIVE_INST_VALU" in pmc:
            traffic["valu_lane_utilisation"] = pmc["SQ_THREAD_CYCLES_VALU"] / (64.0 * pmc["SQ_ACTIVE_INST_VALU"])
        f64 = [pmc.get(k) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")]
        if all(v is not None for v in f64):
            lanes = 64.0 * traffic.get("valu_lane_utilisation", 1.0)
            traffic["fp64_wave_insts_per_launch"] = dict(add=f64[0], mul=f64[1], fma=f64[2], trans=f64[3])
            traffic["fp64_flops_per_launch"] = (f64[0] + f64[1] + 2 * f64[2] + f64[3]) * lanes
            traffic["fp64_note"] = "(ADD + MUL + 2 FMA + TRANS wave-instructions) x 64 lanes x mean lane utilisation of all VALU instructions: an estimate"
        json.dump(traffic, open(os.path.join(pr, f"{tag}_pmc.json"), "w"), indent=1)
        lines += [f"HBM traffic per launch: fetch {fetch_raw/1e6:.1f} MB raw (x2 = {2*fetch_raw/1e6:.1f} MB), write {write/1e6:.1f} MB -> **{(2*fetch_raw+write)/1e6:.1f} MB** "
                  f"(algorithmic bytes per launch: see bench JSON `roofline.algorithmic_bytes_per_launch`).", ""]
for name in ["bench", "bench_off"]:
    p = f"{go}/{tag}_{name}.json"
    if os.path.exists(p) and os.path.getsize(p):
        d = json.loads(open(p).read().strip().splitlines()[-1])
        json.dump(d, open(os.path.join(pr, f"{tag}_{name}.json"), "w"), indent=1)
        lines += [f"## bench.py ({name}): {d['value']:.0f} {d['unit']}, {d['ms_per_step']:.3f} ms/step, kernel {d['roofline']['kernel_ms']:.3f} ms, "
                  f"roofline frac {d['roofline']['frac']:.5f}" + (f", cpu_baseline {d['cpu_baseline']['value']:.0f} steps/s on {d['cpu_baseline']['cores']} threads" if 'cpu_baseline' in d else ""), ""]
open(os.path.join(pr, f"{tag}_rocprof_summary.md"), "w").write("\n".join(lines))
print("\n".join(lines))
