#!/bin/bash
# Build the library under another name with extra compiler flags, for A/B timing on one GPU box:
#   tools/exp/build_variant.sh <tag> [-DNAME=value ...]   ->  build/libzj_<tag>.so
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
tag=$1; shift
C=$ROOT/jpeg-decoder_amd/csrc
mkdir -p $ROOT/build
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wno-unused-value "$@" -shared \
  $C/zj_kernels.hip $C/zj_api.hip $C/zj_host.cpp $C/zj_decoder.cpp -o $ROOT/build/libzj_$tag.so
echo built build/libzj_$tag.so
