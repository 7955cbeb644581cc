#!/bin/bash
OUT=gpurun_out/r3e
rm -rf $OUT && mkdir -p $OUT && export TMPDIR=/tmp
L=$PWD/topdogspectrumanalyser_amd
( TDSA_HIP_LIB=$L/libtdsa_hip.so timeout 900 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
for rep in 1 2 3; do
for lib in nowin hip; do
  TDSA_HIP_LIB=$L/libtdsa_$lib.so python tools/devbench.py --steps 6000 --warmup 1500 >> $OUT/ab.txt 2>&1
  TDSA_HIP_LIB=$L/libtdsa_$lib.so python tools/devbench.py --steps 6000 --warmup 1504 --batch 8 >> $OUT/ab.txt 2>&1
done; done
for lib in nowin hip; do
  TDSA_HIP_LIB=$L/libtdsa_$lib.so python tools/devbench.py --steps 6000 --warmup 1500 --nfft 8192 --hop 8192 --frames 8192 >> $OUT/ab.txt 2>&1
  TDSA_HIP_LIB=$L/libtdsa_$lib.so python tools/devbench.py --steps 6000 --warmup 1500 --nfft 4096 --hop 4096 --frames 4096 --mode pow >> $OUT/ab.txt 2>&1
  TDSA_HIP_LIB=$L/libtdsa_$lib.so python tools/devbench.py --steps 6000 --warmup 1500 --nfft 1024 --hop 1024 --frames 16384 >> $OUT/ab.txt 2>&1
done
cut -c1-150 $OUT/ab.txt
TDSA_HIP_LIB=$L/libtdsa_hip.so python tools/parity_soak.py > $OUT/soak.txt 2>&1; tail -15 $OUT/soak_fuse.txt
