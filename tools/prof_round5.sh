#!/bin/bash
# Round 5: collects the committed evidence on the GPU box (run through gpurun from the repo root); tools/prof_collect5.py
# gpurun_out/prof_r05 turns it into profiles/r05_*.  The C2 / C3 / C4 frame-kernel instruction streams are those of
# round 3 (tests/test_isa_frozen.py): the VALU-issue inputs (profiles/r03_c3_valu_issue.json) are not re-collected.
set -x
OUT=gpurun_out/prof_r05
rm -rf $OUT && mkdir -p $OUT && export TMPDIR=/tmp
( time python bench.py > $OUT/bench_c3.json 2> $OUT/bench.err ) 2> $OUT/bench_c3_time.txt
for c in c2 c4 c5; do python bench.py --config $c > $OUT/bench_$c.json 2>> $OUT/bench.err; done
python bench.py --config c5 --gpus 2 --c5-combine host > $OUT/bench_c5_2ranks.json 2>> $OUT/bench.err
python bench.py --config c5 --gpus 2 --c5-combine peer > $OUT/bench_c5_2ranks_peer.json 2>> $OUT/bench.err
for n in 3 4 8; do   # plumbing + parity of the peer exchange at more ranks (all on this one GPU)
  python bench.py --gpus $n --config c5 --c5-combine peer --reps 2 --min-region-s 0.1 --no-cpu-baseline > $OUT/bench_c5_${n}ranks_peer.json 2>> $OUT/bench.err
done
python bench.py --config c5 --gpus 2 --c5-shard captures > $OUT/bench_c5_2ranks_captures.json 2>> $OUT/bench.err
python bench.py --gpus 2 > $OUT/bench_c3_2ranks.json 2>> $OUT/bench.err
python bench.py --config c4 --gpus 2 > $OUT/bench_c4_2ranks.json 2>> $OUT/bench.err
python tools/c5_scaling.py --steps 150 --reps 2 --ks 8,16,32,64 > $OUT/c5_scaling.txt 2>&1
FAST="--reps 3 --min-region-s 0.15 --no-cpu-baseline --no-parity --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3_serial -- python bench.py --legs serial --streams 1 --batch 1 --steps 800 --warmup 100 $FAST > $OUT/stats_c3_serial.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3_batch -- python bench.py --legs value --streams 1 --batch 8 --steps 800 --warmup 100 $FAST > $OUT/stats_c3_batch.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3_value -- python bench.py --legs value --steps 800 --warmup 100 $FAST > $OUT/stats_c3_value.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c2 -- python bench.py --config c2 --legs value --streams 1 --batch 8 --steps 400 --warmup 50 $FAST > $OUT/stats_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c4 -- python bench.py --config c4 --steps 60 --warmup 10 $FAST > $OUT/stats_c4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -- python bench.py --config c5 --steps 30 --warmup 5 --reps 3 --min-region-s 0.05 --no-cpu-baseline --no-parity > $OUT/stats_c5.log 2>&1
pmc() {  # name "counters" command...
  local name=$1; shift; local ctr=$1; shift
  rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$name -- "$@" > $OUT/pmc_$name.log 2>&1
}
DEV="python tools/devbench.py --steps 9 --warmup 2 --hold 1"
pmc c3_rd FETCH_SIZE $DEV
pmc c3_wr WRITE_SIZE $DEV
pmc c3b_rd FETCH_SIZE $DEV --steps 24 --warmup 8 --batch 8
pmc c3b_wr WRITE_SIZE $DEV --steps 24 --warmup 8 --batch 8
pmc c2_rd FETCH_SIZE $DEV --nfft 4096 --hop 4096 --frames 4096 --mode pow
pmc c2_wr WRITE_SIZE $DEV --nfft 4096 --hop 4096 --frames 4096 --mode pow
pmc c4_rd FETCH_SIZE $DEV --nfft 8192 --hop 8192 --frames 65536 --steps 5
pmc c4_wr WRITE_SIZE $DEV --nfft 8192 --hop 8192 --frames 65536 --steps 5
C5="python bench.py --config c5 --steps 2 --warmup 1 --reps 1 --min-region-s 0.05 --preroll-seconds 0 --no-cpu-baseline --no-parity"
pmc c5_rd FETCH_SIZE $C5
pmc c5_wr WRITE_SIZE $C5
# sizes that are not a power of two: kernel stats and traffic (mixed radix; chirp-z: one launch, long-frame kernels, split plan)
CH="python tools/devbench.py --hold 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_chirp_n1000 -- $CH --nfft 1000 --hop 1000 --frames 4096 --steps 200 --warmup 20 > $OUT/stats_chirp_n1000.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_chirp_n1021 -- $CH --nfft 1021 --hop 1021 --frames 4096 --steps 200 --warmup 20 > $OUT/stats_chirp_n1021.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_chirp_n20000 -- $CH --nfft 20000 --hop 20000 --frames 512 --steps 100 --warmup 10 > $OUT/stats_chirp_n20000.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_chirp_n1000000 -- $CH --nfft 1000000 --hop 1000000 --frames 10 --steps 50 --warmup 5 > $OUT/stats_chirp_n1000000.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_chirp_n20011 -- $CH --nfft 20011 --hop 20011 --frames 512 --steps 100 --warmup 10 > $OUT/stats_chirp_n20011.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_chirp_n999983 -- $CH --nfft 999983 --hop 999983 --frames 10 --steps 50 --warmup 5 > $OUT/stats_chirp_n999983.log 2>&1
pmc chirp_n1000_rd FETCH_SIZE $CH --nfft 1000 --hop 1000 --frames 4096 --steps 5 --warmup 2
pmc chirp_n1000_wr WRITE_SIZE $CH --nfft 1000 --hop 1000 --frames 4096 --steps 5 --warmup 2
pmc chirp_n1021_rd FETCH_SIZE $CH --nfft 1021 --hop 1021 --frames 4096 --steps 5 --warmup 2
pmc chirp_n1021_wr WRITE_SIZE $CH --nfft 1021 --hop 1021 --frames 4096 --steps 5 --warmup 2
pmc chirp_n20011_rd FETCH_SIZE $CH --nfft 20011 --hop 20011 --frames 512 --steps 5 --warmup 2
pmc chirp_n20011_wr WRITE_SIZE $CH --nfft 20011 --hop 20011 --frames 512 --steps 5 --warmup 2
pmc chirp_n20000_rd FETCH_SIZE $CH --nfft 20000 --hop 20000 --frames 512 --steps 5 --warmup 2
pmc chirp_n20000_wr WRITE_SIZE $CH --nfft 20000 --hop 20000 --frames 512 --steps 5 --warmup 2
find $OUT -name "*kernel_stats.csv" | head -10
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +20M -delete
