#!/bin/bash
OUT=gpurun_out/r3k
rm -rf $OUT && mkdir -p $OUT && export TMPDIR=/tmp
L=$PWD/topdogspectrumanalyser_amd
for rep in 1 2 3; do
for lib in hip abl1024; do
  TDSA_HIP_LIB=$L/libtdsa_$lib.so python tools/devbench.py --steps 6000 --warmup 1500 >> $OUT/ab.txt 2>&1
  TDSA_HIP_LIB=$L/libtdsa_$lib.so python tools/devbench.py --steps 6000 --warmup 1504 --batch 8 >> $OUT/ab.txt 2>&1
done; done
cut -c1-150 $OUT/ab.txt
