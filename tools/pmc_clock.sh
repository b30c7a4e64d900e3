#!/bin/bash
# Cycles vs clock of the NeRF-MLP kernel for library variants on one box (is a saving cycles or clock?):
#   tools/pmc_clock.sh NAME ...   (NAME = head | generic (head with NS_OB16_GENERIC=1) | variant of tools/build_*variant.sh)
root=$PWD
export NS_BENCH_NOCHECK=1
cd /tmp && export TMPDIR=/tmp
for name in "$@"; do
  unset NS_LIB_PATH NS_OB16_GENERIC
  [ "$name" = generic ] && export NS_OB16_GENERIC=1
  [ "$name" != head ] && [ "$name" != generic ] && export NS_LIB_PATH=$root/gpurun_ab_$name.so
  out=$root/gpurun_out/pmcclk_$name
  rm -rf $out; mkdir -p $out
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $out -- python $root/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs --no-api-path ${PMC_BENCH_ARGS} > $out/run.log 2>&1 || { tail -5 $out/run.log; echo "$name failed"; continue; }
  python - "$name" "$out" <<'PY'
import csv, glob, collections, sys
name, out = sys.argv[1:]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "nerf_mlp_ob16" in r["Kernel_Name"] or "nerf_mlp_x3" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = []
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "nerf_mlp_ob16" in r["Kernel_Name"] or "nerf_mlp_x3" in r["Kernel_Name"]:
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
top = lambda v: sorted(v)[len(v) // 2:]          # full-frame launches only
m = {k: sum(top(v)) / len(top(v)) for k, v in acc.items()}
d = sum(top(dur)) / len(top(dur)) / 1e6
cyc = m["GRBM_GUI_ACTIVE"] / 8
w = m["SQ_WAVE_CYCLES"]
print(f"{name:10s} {d:7.3f} ms  {cyc / 1e6:7.2f} Mcycles  {cyc / d / 1e6:5.3f} GHz  mfma_busy {m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc:5.3f}  "
      f"wave: active {m['SQ_ACTIVE_INST_ANY'] / w:5.3f} issue-stall {m['SQ_WAIT_INST_ANY'] / w:5.3f} parked {m['SQ_WAIT_ANY'] / w:5.3f} lds-stall {m['SQ_WAIT_INST_LDS'] / w:5.3f}")
PY
done
