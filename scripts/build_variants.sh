#!/bin/bash
# A/B builds of the library for one gpurun call: scripts/build_variants.sh name "flags" [name "flags" ...]
#   -> build_ab/lib<name>.so   (load with MMHN_LIB=build_ab/lib<name>.so; build_ab/ travels to the GPU box, not into git)
cd "$(dirname "$0")/.."
mkdir -p build_ab
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -Wno-comment $flags -o build_ab/lib$name.so metmhn_amd/csrc/engine.hip &
done
wait
ls -la build_ab/
