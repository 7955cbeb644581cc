#!/bin/bash
OUT=gpurun_out/r3f
rm -rf $OUT && mkdir -p $OUT && export TMPDIR=/tmp
L=$PWD/topdogspectrumanalyser_amd
( timeout 900 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
for rep in 1 2; do
for lib in noinl hip; do
  for cfg in "--nfft 8192 --hop 8192 --frames 8192" "--nfft 4096 --hop 4096 --frames 4096 --mode pow" "--nfft 2048 --hop 1024 --frames 16384" "--nfft 256 --hop 256 --frames 65536" "--nfft 128 --hop 128 --frames 65536"; do
    TDSA_HIP_LIB=$L/libtdsa_$lib.so python tools/devbench.py --steps 4000 --warmup 1000 $cfg >> $OUT/ab.txt 2>&1
  done
done; done
cut -c1-150 $OUT/ab.txt
python tools/parity_soak.py > $OUT/soak.txt 2>&1; tail -5 $OUT/soak.txt
