#!/bin/bash
# A/B builds of the same source with extra -D flags: tools/build_variant.sh NAME -DCHEM_HZ=2 ...
# -> chemlab_amd/csrc/variants/libchem_NAME.so (git-ignored, travels with gpurun); select with CHEM_MI355_LIB=<path>
set -e
name=$1; shift
cd "$(dirname "$0")/../chemlab_amd/csrc"
mkdir -p variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result "$@" -shared -o variants/libchem_$name.so chem_api.hip
echo built variants/libchem_$name.so
