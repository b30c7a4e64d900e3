#!/bin/bash
# Build a tuning variant of the library for same-box A/B runs:
#   tools/build_variant.sh NAME [-DNS_SLAB_CHUNKS=32 ...]   ->  gpurun_ab_NAME.so (select with NS_LIB_PATH)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=${NS_VARIANT_SRC:-$root/nerf_sampling_amd/csrc}   # NS_VARIANT_SRC: build the kernels of another checkout (A/B against an older commit)
bld=/tmp/ns_variant_$name
mkdir -p $bld
if [ -n "$NS_ABLATIONS" ]; then   # timing-only NS_EXP_* ablations live in a patch, not in the production sources (tools/ablations/README.md)
  rm -rf $bld/src && mkdir -p $bld/src/nerf_sampling_amd $bld/src/include
  cp -r $src $bld/src/nerf_sampling_amd/csrc && cp $root/include/*.h $bld/src/include/
  (cd $bld/src && patch -p1 < $root/tools/ablations/ns_exp_ablations.patch)
  src=$bld/src/nerf_sampling_amd/csrc
fi
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=off $*"
pids=()
for f in ns_core.cpp ns_render.cpp; do /opt/rocm/bin/hipcc $flags -x hip -c $src/$f -o $bld/${f%.*}.o & pids+=($!); done
for f in ns_rays ns_composite ns_pack ns_nerf_mlp ns_depthnet ns_train; do /opt/rocm/bin/hipcc $flags -c $src/$f.hip -o $bld/$f.o & pids+=($!); done
for f in ns_nerf_mlp_ob16 ns_nerf_mlp_x3 ns_depthnet_ob16; do /opt/rocm/bin/hipcc $flags -mllvm -amdgpu-mfma-vgpr-form -c $src/$f.hip -o $bld/$f.o & pids+=($!); done
/opt/rocm/bin/hipcc $flags -mllvm -amdgpu-mfma-vgpr-form -DNS_OB16_TU_T5 -c $src/ns_nerf_mlp_ob16.hip -o $bld/ns_nerf_mlp_ob16_t5.o & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $bld/*.o -o $root/gpurun_ab_$name.so
echo built $root/gpurun_ab_$name.so
