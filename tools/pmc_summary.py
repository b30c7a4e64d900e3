"""Per-kernel, per-launch means of the rocprofv3 CSVs tools/profile_round.sh wrote under DIR (kernel trace + PMC passes)."""
import collections
import csv
import glob
import json
import os
import sys

d = sys.argv[1]


def short(name):
    for k in ("nerf_mlp_ob16_kernel", "depthnet_ob16_kernel", "nerf_mlp_kernel", "depthnet_kernel", "raw2outputs_kernel",
              "place_z", "get_rays_kernel", "points_kernel"):
        if k in name:
            return k
    return name[:40]


out = {}
for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"stats  {short(r['Name']):24s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e6:9.4f} ms  {r['Percentage']} %")
for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_trace.csv"), recursive=True):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        per[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for k, v in per.items():
        v.sort()
        tail = [x[1] for x in v[-5:]]       # the timed launches (the last 5 of 2 warm-up + 5)
        print(f"trace  {k:24s} last-5 mean {sum(tail) / len(tail) / 1e6:9.4f} ms")
        out.setdefault(k, {})["trace_last5_ms"] = sum(tail) / len(tail) / 1e6
for p in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_inst", "pmc_sq_f16"):
    for f in glob.glob(os.path.join(d, p, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            key = k + ("[f16]" if p.endswith("_f16") else "")
            for c, v in cs.items():
                m = sum(v) / len(v)
                print(f"{p:9s} {key:24s} {c:32s} {m:.6g}  (n={len(v)})")
                out.setdefault(key, {})[c] = m
for k, c in out.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        print(f"derived {k}: MFMA-pipe busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (c['GRBM_GUI_ACTIVE'] / 8):.4f} of cycles")
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        print(f"derived {k}: HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE = {(2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024:.6g}")
json.dump(out, open(os.path.join(d, "per_launch_means.json"), "w"), indent=1)
