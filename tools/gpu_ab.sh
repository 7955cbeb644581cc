#!/bin/bash
# A/B of the working tree's library against libtdsa_prev.so (the previous commit), same box, alternating
OUT=gpurun_out/ab
rm -rf $OUT && mkdir -p $OUT && export TMPDIR=/tmp
L=$PWD/topdogspectrumanalyser_amd
( timeout 900 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
for rep in 1 2 3; do
for lib in prev hip; do
  TDSA_HIP_LIB=$L/libtdsa_$lib.so python tools/devbench.py --steps 6000 --warmup 1500 >> $OUT/ab.txt 2>&1
  TDSA_HIP_LIB=$L/libtdsa_$lib.so python tools/devbench.py --steps 6000 --warmup 1504 --batch 8 >> $OUT/ab.txt 2>&1
done; done
for lib in prev hip; do
  for cfg in "--nfft 8192 --hop 8192 --frames 8192" "--nfft 4096 --hop 4096 --frames 4096 --mode pow" "--nfft 2048 --hop 1024 --frames 16384"; do
    TDSA_HIP_LIB=$L/libtdsa_$lib.so python tools/devbench.py --steps 4000 --warmup 1000 $cfg >> $OUT/ab.txt 2>&1
  done
done
cut -c1-150 $OUT/ab.txt
