# tools/profile_round.sh [TAG]: the round's numbers of record on one box -- the default bench line, a rocprofv3 kernel
# trace of the same command, and the PMC passes (each in its own run: counters never share a run with a trace domain
# other than --kernel-trace).  Outputs under gpurun_out/TAG/; copy the summaries to profiles/.
set -e
tag=${1:-r03}
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
