#!/bin/bash
# Collects the round's committed evidence on the GPU box (run through gpurun from the repo root):
# bench lines for every config, rocprofv3 kernel stats of the same bench command, HBM counters (separate
# --pmc passes: FETCH_SIZE and WRITE_SIZE do not fit one pass), SQ wait/issue counters for the C3 kernel.
set -x
OUT=gpurun_out/prof_r02
rm -rf $OUT && mkdir -p $OUT && export TMPDIR=/tmp
python bench.py > $OUT/bench_c3.json 2> $OUT/bench.err
for c in c2 c4 c5; do python bench.py --config $c > $OUT/bench_$c.json 2>> $OUT/bench.err; done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3 -- python bench.py --streams 1 --steps 300 --warmup 50 --reps 3 --no-cpu-baseline > $OUT/stats_c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -- python bench.py --config c5 --steps 30 --warmup 5 --reps 3 --no-cpu-baseline > $OUT/stats_c5.log 2>&1
pmc() {  # name counters... -- command
  local name=$1; shift; local ctr=$1; shift
  rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$name -- "$@" > $OUT/pmc_$name.log 2>&1
}
pmc c3_rd FETCH_SIZE python tools/devbench.py --steps 3 --warmup 1 --hold 1
pmc c3_wr WRITE_SIZE python tools/devbench.py --steps 3 --warmup 1 --hold 1
pmc c2_rd FETCH_SIZE python tools/devbench.py --nfft 4096 --hop 4096 --frames 4096 --steps 3 --warmup 1 --hold 1 --mode pow
pmc c2_wr WRITE_SIZE python tools/devbench.py --nfft 4096 --hop 4096 --frames 4096 --steps 3 --warmup 1 --hold 1 --mode pow
pmc c4_rd FETCH_SIZE python tools/devbench.py --nfft 8192 --hop 8192 --frames 8192 --steps 3 --warmup 1 --hold 1
pmc c4_wr WRITE_SIZE python tools/devbench.py --nfft 8192 --hop 8192 --frames 8192 --steps 3 --warmup 1 --hold 1
pmc c5_rd FETCH_SIZE python bench.py --config c5 --steps 2 --warmup 1 --reps 1 --preroll-seconds 0 --no-cpu-baseline
pmc c5_wr WRITE_SIZE python bench.py --config c5 --steps 2 --warmup 1 --reps 1 --preroll-seconds 0 --no-cpu-baseline
pmc c3_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS" python tools/devbench.py --steps 3 --warmup 1 --hold 1
python tools/pipebench.py > $OUT/pipebench.txt 2>&1
find $OUT -name "*.csv" | head -40
