#!/bin/bash
# Same-box check + A/B of generated-stream variants (tools/build_asm_variant.sh NAME ...), to be run on a GPU box:
#   tools/ab_asm.sh NAME ...
# Each variant's raw output on seeded inputs (tools/raw_dump.py) must be bit-identical to the in-tree build's -- the
# variant libraries hold only the production bf16 kernel, so this replaces tools/ab.sh's pytest step -- then the
# variants and the in-tree build ("head") are timed interleaved, three rounds (tools/ab.sh).
set -e
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
python tools/raw_dump.py /tmp/raw_head.npy
for v in "$@"; do
  NS_LIB_PATH=$PWD/gpurun_ab_$v.so python tools/raw_dump.py /tmp/raw_$v.npy
  cmp /tmp/raw_head.npy /tmp/raw_$v.npy && echo "$v bit-identical"
done
AB_SKIP_TESTS=1 AB_BENCH_ARGS="--no-other-configs --no-api-path" bash tools/ab.sh 3 head "$@" 2>&1 | grep -v amdgpu.ids
