set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03q
python tools/raw_dump.py /tmp/raw_head.npy
for v in "$@"; do
  NS_LIB_PATH=$PWD/gpurun_ab_$v.so python tools/raw_dump.py /tmp/raw_$v.npy
  cmp /tmp/raw_head.npy /tmp/raw_$v.npy && echo "$v bit-identical"
done
AB_SKIP_TESTS=1 AB_BENCH_ARGS="--no-other-configs --no-api-path" bash tools/ab.sh 3 head "$@" 2>&1 | grep -v amdgpu.ids
