#!/bin/bash
# in-situ issue rate of the frame kernel's VALU stream: timing-only builds (tools/build_variants.sh ablN "-DTDSA_ABLATE=N")
# named on the command line, production library first, alternating, 8-second launches
OUT=gpurun_out/calib
rm -rf $OUT && mkdir -p $OUT && export TMPDIR=/tmp
L=$PWD/topdogspectrumanalyser_amd
for rep in 1 2; do
for lib in hip "$@"; do
  TDSA_HIP_LIB=$L/libtdsa_$lib.so timeout 120 python tools/devbench.py --steps 3000 --warmup 1000 --batch 8 >> $OUT/calib.txt 2>&1
done; done
cut -c1-160 $OUT/calib.txt
