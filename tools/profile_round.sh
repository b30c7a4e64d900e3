# tools/profile_round.sh [TAG]: the round's numbers of record on one box -- the default bench line, a rocprofv3 kernel
# trace of the same command, and the PMC passes (each in its own run: counters never share a run with a trace domain
# other than --kernel-trace).  Outputs under gpurun_out/TAG/; copy the summaries to profiles/.
set -e
tag=${1:-r04}
root=$PWD
mkdir -p gpurun_out/$tag
python bench.py > gpurun_out/$tag/bench_default.json 2> gpurun_out/$tag/bench_default.err
tail -1 gpurun_out/$tag/bench_default.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
export NS_BENCH_NOCHECK=1   # profiled runs: no accuracy leg, so every launch in the CSVs is a full-frame launch
B="python $root/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$tag/stats -- $B --steps 5 --warmup 2 > $root/gpurun_out/$tag/bench_under_rocprof.json 2>$root/gpurun_out/$tag/stats.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $root/gpurun_out/$tag/pmc_sq -- $B --steps 2 --warmup 1 > /dev/null 2>$root/gpurun_out/$tag/pmc_sq.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $root/gpurun_out/$tag/pmc_fetch -- $B --steps 2 --warmup 1 > /dev/null 2>$root/gpurun_out/$tag/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $root/gpurun_out/$tag/pmc_write -- $B --steps 2 --warmup 1 > /dev/null 2>$root/gpurun_out/$tag/pmc_write.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $root/gpurun_out/$tag/pmc_inst -- $B --steps 2 --warmup 1 > /dev/null 2>$root/gpurun_out/$tag/pmc_inst.err
# the same SQ pass on the f16 kernel (same instruction stream as bf16: is its extra time cycles or clock?)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $root/gpurun_out/$tag/pmc_sq_f16 -- $B --dtype f16 --steps 2 --warmup 1 > /dev/null 2>$root/gpurun_out/$tag/pmc_sq_f16.err
# the DepthNet training step (1024 rays), eager and as one hipGraph replay per step
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$tag/train_eager -- python $root/tools/bench_train_step.py bf16 > $root/gpurun_out/$tag/train_eager.json 2>$root/gpurun_out/$tag/train_eager.err
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$tag/train_graph -- python $root/tools/bench_train_step.py bf16 --graph > $root/gpurun_out/$tag/train_graph.json 2>$root/gpurun_out/$tag/train_graph.err
cd $root
python tools/bench_train_step.py bf16 | tail -n 1 > gpurun_out/$tag/train_step_eager_unprofiled.json
python tools/bench_train_step.py bf16 --graph | tail -n 1 > gpurun_out/$tag/train_step_graph_unprofiled.json
python tools/pmc_summary.py gpurun_out/$tag | tee gpurun_out/$tag/summary.txt
# the HBM-traffic record bench.py quotes in `roofline.traffic`, tied to the kernel sources it was measured on
python - "$tag" <<'PY'
import csv, glob, json, sys
sys.path.insert(0, ".")
import bench
tag = sys.argv[1]
m = json.load(open(f"gpurun_out/{tag}/per_launch_means.json"))
line = json.loads(open(f"gpurun_out/{tag}/bench_under_rocprof.json").read().strip().splitlines()[-1])
rec = {"workload": "800x800, 64 samples/ray, one frame per step (fitted scene), bf16",
       "renderer": line["config"]["renderer"],
       "kernel_sources_sha256": bench.kernel_sources_sha256(),
       "note": "separate --pmc passes (pmc_fetch / pmc_write counter_collection.csv of the same run, every launch a full frame); "
               "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of streaming-read bytes).  Algorithmic bytes of the "
               "one-kernel renderer's MLP launch: 40 B in (o, d, viewdirs, DepthNet depth) + 16 B out (rgb, disp) per ray = 36 MB "
               "per 640 000-ray frame + the 1.05 MiB weight stream; z and raw never reach HBM."}
hbm = lambda c: int(round((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024))
for k, name in (("nerf_mlp_ob16_kernel", None), ("depthnet_ob16_kernel", "depthnet_ob16_kernel")):
    d = {"FETCH_SIZE_KB": m[k]["FETCH_SIZE"], "WRITE_SIZE_KB": m[k]["WRITE_SIZE"], "hbm_bytes_per_launch": hbm(m[k])}
    if name:
        rec[name] = d
    else:
        rec.update(d)
        rec["kernel"] = "nerf_mlp_ob16_kernel (production program, the launch the headline bench times)"
# every kernel of a frame: launches per frame from the kernel trace (calls / frames), bytes from the PMC passes
stats = sorted(glob.glob(f"gpurun_out/{tag}/stats/*/*kernel_stats.csv"))[0]
frames = line["steps"] + line["warmup"]
per_frame, total = {}, 0
for r in csv.DictReader(open(stats)):
    key = next((k for k in m if k in r["Name"] and "FETCH_SIZE" in m[k] and "WRITE_SIZE" in m[k]), None)
    if key is None:
        continue
    n = int(r["Calls"]) / frames
    per_frame[key] = {"launches_per_frame": n, "avg_ms": float(r["AverageNs"]) / 1e6, "hbm_bytes_per_launch": hbm(m[key])}
    total += n * hbm(m[key])
rec["kernels_per_frame"] = per_frame
rec["hbm_bytes_per_frame_all_kernels"] = int(round(total))
json.dump(rec, open(f"gpurun_out/{tag}/traffic_nerf_mlp.json", "w"), indent=1)
print("traffic record:", rec["hbm_bytes_per_launch"], "B per MLP launch,", rec["hbm_bytes_per_frame_all_kernels"], "B per frame over",
      sorted(per_frame), "sources", rec["kernel_sources_sha256"][:16])
PY
