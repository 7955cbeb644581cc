#!/bin/bash
# Round 6: HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, no trace domains) of the shipped kernels,
# incl. the C3 frame kernel with the per-frame scalars on (c3s).  tools/prof_collect5.py gpurun_out/prof_r06 r06 turns them into profiles/r06_*_pmc.json.
set -x
OUT=gpurun_out/prof_r06
mkdir -p $OUT && export TMPDIR=/tmp
pmc() { local name=$1; shift; local ctr=$1; shift; rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$name -- "$@" > $OUT/pmc_$name.log 2>&1; }
DEV="python tools/devbench.py --steps 9 --warmup 2 --hold 1"
pmc c3_rd FETCH_SIZE $DEV
pmc c3_wr WRITE_SIZE $DEV
pmc c3s_rd FETCH_SIZE $DEV --stats 4096:12288
pmc c3s_wr WRITE_SIZE $DEV --stats 4096:12288
pmc c3b_rd FETCH_SIZE $DEV --steps 24 --warmup 8 --batch 8
pmc c3b_wr WRITE_SIZE $DEV --steps 24 --warmup 8 --batch 8
pmc c2_rd FETCH_SIZE $DEV --nfft 4096 --hop 4096 --frames 4096 --mode pow
pmc c2_wr WRITE_SIZE $DEV --nfft 4096 --hop 4096 --frames 4096 --mode pow
pmc c4_rd FETCH_SIZE $DEV --nfft 8192 --hop 8192 --frames 65536 --steps 5
pmc c4_wr WRITE_SIZE $DEV --nfft 8192 --hop 8192 --frames 65536 --steps 5
C5="python bench.py --config c5 --steps 2 --warmup 1 --reps 1 --min-region-s 0.05 --preroll-seconds 0 --no-cpu-baseline --no-parity"
pmc c5_rd FETCH_SIZE $C5
pmc c5_wr WRITE_SIZE $C5
find $OUT -name "*.db" -delete
