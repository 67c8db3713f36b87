#!/bin/bash
# Whole library with one source file recompiled under extra -D flags (the other objects are the shipped ones):
#   tools/ablate/build_lib_variant.sh fused.hip k2w1 -DK2_WAVES=1   ->  runia_core_amd/librunia_k2w1.so
# The variant travels to the GPU box with the snapshot; select it with RUNIA_LIB=$PWD/runia_core_amd/librunia_<tag>.so
set -e
cd "$(dirname "$0")/../../runia_core_amd/csrc"
src=$1; tag=$2; shift 2
make -s -j8
form="-mllvm -amdgpu-mfma-vgpr-form"; [ "$src" = knn_bf16.hip ] && form=""   # as the Makefile
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $form "$@" -c $src -o /tmp/variant_$tag.o
objs=""
for f in *.o; do if [ "$f" = "${src%.hip}.o" ]; then objs="$objs /tmp/variant_$tag.o"; else objs="$objs $f"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../librunia_$tag.so
echo "built runia_core_amd/librunia_$tag.so"
