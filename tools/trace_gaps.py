"""Per-frame timeline from a rocprofv3 --kernel-trace CSV: kernel durations and the idle gaps between them."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    for k in ("nerf_mlp", "depthnet", "raw2outputs", "place_z", "get_rays", "copyBuffer", "elementwise", "fill", "importance_z", "coarse_z", "sample_pdf", "sort_rows", "points"):
        if k in n: return k
    return n[:24]
# frames start at get_rays
frames, cur = [], []
for r in rows:
    if "get_rays" in r["Kernel_Name"] and cur:
        frames.append(cur); cur = []
    cur.append(r)
frames.append(cur)
for f in frames[3:6]:
    t0 = int(f[0]["Start_Timestamp"]); prev_end = t0
    print("frame:")
    for r in f:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"  +{(s-t0)/1e3:9.1f} us  gap {(s-prev_end)/1e3:8.1f}  dur {(e-s)/1e3:9.1f}  {short(r['Kernel_Name'])}")
        prev_end = e
starts = [int(f[0]["Start_Timestamp"]) for f in frames]
print("frame-to-frame ms:", [round((b-a)/1e6, 3) for a, b in zip(starts, starts[1:])])
