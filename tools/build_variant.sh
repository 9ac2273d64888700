#!/bin/bash
# Build a library variant from the working-tree sources into build_ab/<name>.so without touching csrc/liborbhip.so
# (for A/B runs of several libraries in one GPU call, tools/ab_libs.sh):   tools/build_variant.sh name [extra hipcc flags]
set -e
NAME=$1; shift
CS=vi-orb-slam-icra2018_amd/csrc
OBJ=/tmp/orbhip_variant_$NAME
mkdir -p $OBJ build_ab
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950:sramecc+ -ffp-contract=off -Wall -Wno-unused-function"
pids=()
for f in $CS/*.hip; do
  b=$(basename $f .hip)
  extra=""
  { [ $b = k_blur ] || [ $b = k_hamming ]; } && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
  if [ ! -f $OBJ/$b.o ] || [ $f -nt $OBJ/$b.o ] || [ -n "$(find $CS -name '*.h' -newer $OBJ/$b.o 2>/dev/null)" ]; then
    /opt/rocm/bin/hipcc $FLAGS $extra "$@" -c $f -o $OBJ/$b.o &
    pids+=($!)
    if [ ${#pids[@]} -ge 6 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950:sramecc+ -shared -fPIC -o build_ab/$NAME.so $OBJ/*.o -ldl
echo built build_ab/$NAME.so
