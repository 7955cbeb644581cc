#!/bin/bash
# same-box alternating A/B of variant libraries against the production one: tools/ab_quick.sh name [name ...]
OUT=gpurun_out/abq
rm -rf $OUT && mkdir -p $OUT && export TMPDIR=/tmp
L=$PWD/topdogspectrumanalyser_amd
for rep in 1 2 3; do
for lib in hip "$@"; do
  TDSA_HIP_LIB=$L/libtdsa_$lib.so timeout 120 python tools/devbench.py --steps 4000 --warmup 1500 >> $OUT/ab.txt 2>&1
  TDSA_HIP_LIB=$L/libtdsa_$lib.so timeout 120 python tools/devbench.py --steps 4000 --warmup 1504 --batch 8 >> $OUT/ab.txt 2>&1
done; done
cut -c1-62,105-150 $OUT/ab.txt
