set -e
root=$PWD
mkdir -p gpurun_out/r01e
python bench.py > gpurun_out/r01e/bench_default.json 2> gpurun_out/r01e/bench_default.err
tail -1 gpurun_out/r01e/bench_default.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
export NS_BENCH_NOCHECK=1   # profiled runs: no PSNR leg, so every launch in the CSVs is a full-frame launch
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r01e/stats -- python $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $root/gpurun_out/r01e/bench_under_rocprof.json 2>$root/gpurun_out/r01e/stats.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $root/gpurun_out/r01e/pmc_sq -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>$root/gpurun_out/r01e/pmc_sq.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $root/gpurun_out/r01e/pmc_fetch -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>$root/gpurun_out/r01e/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $root/gpurun_out/r01e/pmc_write -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>$root/gpurun_out/r01e/pmc_write.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $root/gpurun_out/r01e/pmc_inst -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>$root/gpurun_out/r01e/pmc_inst.err
cd $root; find gpurun_out/r01e -name "*.csv" | head -20
