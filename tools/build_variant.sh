#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...]  ->  gradient-sdf_amd/csrc/variants/libgsdf_NAME.so  (kernel-variant experiments;
# loaded with GSDF_LIB=... or binding.load(path); *.so is git-ignored but travels to the GPU box)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/gradient-sdf_amd/csrc
out=$src/variants
tmp=$(mktemp -d)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -mllvm -amdgpu-set-wave-priority"
for f in gsdf_kernels gsdf_ba gsdf_capi gsdf_merge gsdf_sort; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $src/$f.hip -o $tmp/$f.o &
done
wait
g++ -shared -o $out/libgsdf_$name.so $tmp/gsdf_kernels.o $tmp/gsdf_ba.o $tmp/gsdf_capi.o $tmp/gsdf_merge.o $tmp/gsdf_sort.o -lstdc++ -ldl -lm
rm -rf $tmp
echo built $out/libgsdf_$name.so
