#!/bin/bash
# Build a variant of the hand-scheduled layer streams for a same-box A/B (tools/ab.sh):
#   tools/build_asm_variant.sh NAME [generator flags ...]   ->  gpurun_ab_NAME.so
# Only the production bf16 kernel is instantiated in the variant (NS_OB16_VARIANT_BUILD); every other object is the
# in-tree build's.  The DepthNet, compositing etc. are shared, so the bench's frame differs in the NeRF-MLP kernel only.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
bld=/tmp/ns_asmvar_$name
mkdir -p $bld
python $root/tools/gen_ob16_asm.py -o $bld/ns_ob16_asm.inc "$@" 2> $bld/gen.log
src=$root/nerf_sampling_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form \
  -DNS_OB16_VARIANT_BUILD -DNS_OB16_ASM_INC="\"$bld/ns_ob16_asm.inc\"" ${NS_VARIANT_DEFS} -c $src/ns_nerf_mlp_ob16.hip -o $bld/ns_nerf_mlp_ob16.o
objs=$(ls $src/build/*.o | grep -v ns_nerf_mlp_ob16.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $bld/ns_nerf_mlp_ob16.o -o $root/gpurun_ab_$name.so
echo "built gpurun_ab_$name.so: $(grep 'bf16 A->V:' $bld/gen.log)"
