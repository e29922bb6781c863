#!/bin/bash
# usage: tools/build_variant.sh <name> "<extra hipcc flags>" [source, default sparse_conv]   e.g.  tools/build_variant.sh ra4 "-DSEEVCN_RS3_RA=4"
# Builds see-vcn_amd/lib/variants/libseevcn_hip_<name>.so: <source>.hip recompiled with the flags, every other object as built by `make`.
# A/B runs on one box: SEEVCN_LIB=see-vcn_amd/lib/variants/libseevcn_hip_<name>.so python bench.py ...
set -e
cd "$(dirname "$0")/../see-vcn_amd/csrc"
make -j8 >/dev/null
mkdir -p ../lib/variants ../build/variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics -ffp-contract=off -Wall -Wno-unused-function"
SRC=${3:-sparse_conv}
/opt/rocm/bin/hipcc $FLAGS $2 -c $SRC.hip -o ../build/variants/${SRC}_$1.o
objs=$(ls ../build/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libseevcn_hip_$1.so $objs ../build/variants/${SRC}_$1.o
echo built ../lib/variants/libseevcn_hip_$1.so
