#!/usr/bin/env python3
"""On the GPU box, after tools/profile_capture_tasks_pmc.sh <tag>: condense the raw FETCH_SIZE / WRITE_SIZE counter CSVs (too large for gpurun's return limit) into
gpurun_out/summ/<tag>_tasks_pmc.json = {task: {FETCH_SIZE, WRITE_SIZE (KiB per launch, mean over the step-kernel dispatches), vgpr, scratch}}."""
import csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02t"
go = "gpurun_out"
out = {}
for d in sorted(glob.glob(f"{go}/{tag}_*_fetch")):
    t = os.path.basename(d)[len(tag) + 1:-len("_fetch")]
    v = {}
    for sub, cn in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        g = sorted(glob.glob(f"{go}/{tag}_{t}_{sub}/*/*_counter_collection.csv"), key=os.path.getmtime)
        if not g:
            continue
        rows = [r for r in csv.DictReader(open(g[-1])) if "hrg_step_kernel" in r["Kernel_Name"] and r["Counter_Name"] == cn]
        if rows:
            v[cn] = sum(float(r["Counter_Value"]) for r in rows) / len(rows)
            v["scratch"], v["vgpr"], v["kernel"] = rows[0]["Scratch_Size"], int(rows[0]["VGPR_Count"]) + int(rows[0].get("Accum_VGPR_Count") or 0), rows[0]["Kernel_Name"].split("(")[0]
    out[t] = v
os.makedirs(f"{go}/summ", exist_ok=True)
json.dump(out, open(f"{go}/summ/{tag}_tasks_pmc.json", "w"), indent=1)
print(json.dumps(out))
