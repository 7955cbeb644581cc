#!/bin/bash
# re-collects only the HBM counter passes of prof_round3.sh into gpurun_out/prof_r03 (other results there are kept)
set -x
OUT=gpurun_out/prof_r03
mkdir -p $OUT && export TMPDIR=/tmp
pmc() { local name=$1; shift; local ctr=$1; shift; rm -rf $OUT/pmc_$name; rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$name -- "$@" > $OUT/pmc_$name.log 2>&1; }
DEV="python tools/devbench.py --steps 9 --warmup 2 --hold 1"
pmc c3_rd FETCH_SIZE $DEV
pmc c3_wr WRITE_SIZE $DEV
pmc c3b_rd FETCH_SIZE $DEV --steps 24 --warmup 8 --batch 8
pmc c3b_wr WRITE_SIZE $DEV --steps 24 --warmup 8 --batch 8
pmc c2_rd FETCH_SIZE $DEV --nfft 4096 --hop 4096 --frames 4096 --mode pow
pmc c2_wr WRITE_SIZE $DEV --nfft 4096 --hop 4096 --frames 4096 --mode pow
pmc c4_rd FETCH_SIZE $DEV --nfft 8192 --hop 8192 --frames 8192
pmc c4_wr WRITE_SIZE $DEV --nfft 8192 --hop 8192 --frames 8192
