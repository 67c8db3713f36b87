#!/bin/bash
# usage: tools/ablate/build.sh <file.hip> <prefix> <flag values...>
set -e
cd "$(dirname "$0")"
src=$1; prefix=$2; shift 2
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -shared -D${MACRO:-RUNIA_ABLATE}=$v \
     -I../../runia_core_amd/csrc ../../runia_core_amd/csrc/$src -o lib${prefix}_$v.so &
done
wait
