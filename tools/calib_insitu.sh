#!/bin/bash
OUT=gpurun_out/calib
rm -rf $OUT && mkdir -p $OUT && export TMPDIR=/tmp
L=$PWD/topdogspectrumanalyser_amd
for rep in 1 2; do
for lib in hip abl6 abl7 abl39 abl71 abl519 abl8199 abl4103; do
  TDSA_HIP_LIB=$L/libtdsa_$lib.so timeout 120 python tools/devbench.py --steps 3000 --warmup 1000 --batch 8 >> $OUT/calib.txt 2>&1
done; done
cut -c1-160 $OUT/calib.txt
