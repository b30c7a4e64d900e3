#!/bin/bash
# Same-box interleaved A/B of library variants (device-to-device spread is 5-8 %, so only same-box deltas count):
#   tools/ab.sh ROUNDS head slab32 ...     ("head" = the in-tree library; NAME = gpurun_ab_NAME.so)
# Each variant first has to pass the kernel + render parity tests.
set -e
rounds=$1; shift
export NS_BENCH_NOCHECK=1
mkdir -p gpurun_out
for v in "$@"; do
  [ "$v" = head ] && continue
  [ -n "$AB_SKIP_TESTS" ] && continue   # diagnostic variants that are wrong on purpose
  NS_LIB_PATH=$PWD/gpurun_ab_$v.so timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_render.py -x -q -m gpu ${AB_PYTEST_ARGS} > gpurun_out/ab_$v.log 2>&1 || { tail -30 gpurun_out/ab_$v.log; echo "variant $v FAILED parity"; exit 1; }
  echo "$v: $(tail -1 gpurun_out/ab_$v.log)"
done
for i in $(seq $rounds); do
  for v in "$@"; do
    if [ "$v" = head ]; then unset NS_LIB_PATH; else export NS_LIB_PATH=$PWD/gpurun_ab_$v.so; fi
    echo "$v $(timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline ${AB_BENCH_ARGS} | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],3), round(d["roofline"]["achieved"],1))')"
  done
done
