#!/bin/bash
# Build a variant of libbcehip.so with extra -D flags into bce_amd/lib/var_NAME.so (for A/B runs on the GPU box:
# BCE_HIP_LIB=bce_amd/lib/var_NAME.so python bench.py ...).   tools/build_variant.sh NAME -DFOO=1 ...
set -e
NAME=$1; shift
cd "$(dirname "$0")/../bce_amd/csrc"
make -s -j8 OBJDIR=build_$NAME LIBDIR=../lib/var_$NAME BINDIR=../lib/var_$NAME HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wextra -Wno-unused-parameter -ffp-contract=off $*" ../lib/var_$NAME/libbcehip.so
mv ../lib/var_$NAME/libbcehip.so ../lib/var_$NAME.so && rmdir ../lib/var_$NAME && rm -rf build_$NAME
echo built bce_amd/lib/var_$NAME.so
