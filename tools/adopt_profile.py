"""Copy the summaries of one tools/profile_round.sh run into profiles/ under a prefix (the judged, committed copies):
    python tools/adopt_profile.py r03u r03c"""
import glob
import shutil
import sys

tag, pre = sys.argv[1:3]
src, dst = f"gpurun_out/{tag}", "profiles"
one = lambda pat: sorted(glob.glob(f"{src}/{pat}"))[0]  # noqa: E731
shutil.copy(f"{src}/traffic_nerf_mlp.json", f"{dst}/{pre}_traffic_nerf_mlp.json")
shutil.copy(f"{src}/per_launch_means.json", f"{dst}/{pre}_pmc_per_launch_means.json")
shutil.copy(f"{src}/summary.txt", f"{dst}/{pre}_summary.txt")
shutil.copy(f"{src}/bench_default.json", f"{dst}/{pre}_bench_bf16_default.json")
shutil.copy(f"{src}/bench_under_rocprof.json", f"{dst}/{pre}_bench_bf16_under_rocprof.json")
shutil.copy(one("stats/*/*kernel_stats.csv"), f"{dst}/{pre}_bench_bf16_kernel_stats.csv")
for p in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_inst", "pmc_sq_f16"):
    shutil.copy(one(f"{p}/*/*counter_collection.csv"), f"{dst}/{pre}_{p}_counter_collection.csv")
for t in ("eager", "graph"):
    shutil.copy(one(f"train_{t}/*/*kernel_stats.csv"), f"{dst}/{pre}_train_step_{t}_kernel_stats.csv")
print("adopted", tag, "as", pre)
