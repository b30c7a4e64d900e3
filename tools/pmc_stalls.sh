#!/bin/bash
# Stall breakdown of the NeRF-MLP kernel via SQ counters, for one library variant:
#   tools/pmc_stalls.sh NAME   (NAME = head | variant built by tools/build_variant.sh)
set -e
name=$1
root=$PWD
export NS_BENCH_NOCHECK=1
[ "$name" != head ] && export NS_LIB_PATH=$root/gpurun_ab_$name.so
mkdir -p $root/gpurun_out/pmc_$name
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $root/gpurun_out/pmc_$name/p$i -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $root/gpurun_out/pmc_$name/p$i.log 2>&1 || { tail -5 $root/gpurun_out/pmc_$name/p$i.log; echo "pass $i failed"; }
done
cd $root
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_$name/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "nerf_mlp" in r["Kernel_Name"] and "MmaF32" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = sorted(v)[len(v)//2:]   # full-frame launches only (drop the small band launches)
        print(f"$name {k} {sum(v)/len(v):.4g}")
PY
