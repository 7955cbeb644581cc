#!/bin/bash
# Developer builds of libtdsa_hip.so with extra -D flags, next to the production library (git-ignored, shipped by gpurun):
#   tools/build_variants.sh name "flags" [name "flags" ...]   ->  topdogspectrumanalyser_amd/libtdsa_<name>.so
# Developer switches need -DTDSA_DEV as well (e.g. dev "-DTDSA_DEV -DTDSA_TIMELINE"); such a library is never loaded by the
# package unless TDSA_HIP_LIB points at it.  tools/c5_ab_libs.py alternates several of them inside one process.
set -e
cd "$(dirname "$0")/../topdogspectrumanalyser_amd/csrc"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  make -s -j"$(nproc)" B=build_$name OUT=../libtdsa_$name.so EXTRA="-fno-slp-vectorize $flags" >/dev/null
  echo "built libtdsa_$name.so  ($flags)"
done
