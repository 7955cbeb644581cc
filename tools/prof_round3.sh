#!/bin/bash
# Round 3: collects the committed evidence on the GPU box (run through gpurun from the repo root).
#   bench lines of every configuration (+ C5 sharded over two ranks on the one GPU)
#   rocprofv3 --kernel-trace --stats of bench.py in each submission mode of C3 (one launch shape per run: --legs)
#   HBM counters (separate --pmc passes: FETCH_SIZE and WRITE_SIZE do not fit one pass) for C2..C5
#   SQ issue / wait counters and the per-class VALU instruction counters of the C3 frame kernel
#   per-class VALU issue cost (tools/ubench/valu_rate2) and the timing-only ablation builds of the frame kernel
# Before sending it: tools/build_variants.sh abl2 "-DTDSA_ABLATE=2" abl4 "-DTDSA_ABLATE=4" abl6 "-DTDSA_ABLATE=6"
# (the ablation libraries are git-ignored and travel with the snapshot); afterwards tools/prof_collect3.py and
# tools/valu_issue.py turn gpurun_out/prof_r03 into profiles/r03_*.
set -x
OUT=gpurun_out/prof_r03
rm -rf $OUT && mkdir -p $OUT && export TMPDIR=/tmp
( time python bench.py > $OUT/bench_c3.json 2> $OUT/bench.err ) 2> $OUT/bench_c3_time.txt
for c in c2 c4 c5; do python bench.py --config $c > $OUT/bench_$c.json 2>> $OUT/bench.err; done
python bench.py --config c5 --gpus 2 > $OUT/bench_c5_2ranks.json 2>> $OUT/bench.err
FAST="--reps 3 --min-region-s 0.15 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3_serial -- python bench.py --legs serial --streams 1 --batch 1 --steps 800 --warmup 100 $FAST > $OUT/stats_c3_serial.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3_batch -- python bench.py --legs value --streams 1 --batch 8 --steps 800 --warmup 100 $FAST > $OUT/stats_c3_batch.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3_value -- python bench.py --legs value --steps 800 --warmup 100 $FAST > $OUT/stats_c3_value.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3_streams -- python bench.py --legs streams --steps 800 --warmup 100 $FAST > $OUT/stats_c3_streams.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -- python bench.py --config c5 --steps 30 --warmup 5 --reps 3 --min-region-s 0.05 --no-cpu-baseline > $OUT/stats_c5.log 2>&1
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
pmc c4_rd FETCH_SIZE $DEV --nfft 8192 --hop 8192 --frames 8192
pmc c4_wr WRITE_SIZE $DEV --nfft 8192 --hop 8192 --frames 8192
C5="python bench.py --config c5 --steps 2 --warmup 1 --reps 1 --min-region-s 0.05 --preroll-seconds 0 --no-cpu-baseline"
pmc c5_rd FETCH_SIZE $C5
pmc c5_wr WRITE_SIZE $C5
pmc c3_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS" $DEV
pmc c3_cls "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_LDS SQ_INSTS_VMEM" $DEV
pmc c3_cls2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE GRBM_GUI_ACTIVE" $DEV
./tools/ubench/valu_rate2 > $OUT/ubench_valu_rate2.txt 2>&1
# steady-state ablations of the frame kernel (timing only; cache-resident devbench shape, one box, back to back)
for lib in hip abl2 abl4 abl6; do
  [ -f $PWD/topdogspectrumanalyser_amd/libtdsa_$lib.so ] || continue
  TDSA_HIP_LIB=$PWD/topdogspectrumanalyser_amd/libtdsa_$lib.so python tools/devbench.py --steps 6000 --warmup 1500 >> $OUT/ablation.txt 2>&1
done
for lib in hip abl6; do
  [ -f $PWD/topdogspectrumanalyser_amd/libtdsa_$lib.so ] || continue
  TDSA_HIP_LIB=$PWD/topdogspectrumanalyser_amd/libtdsa_$lib.so python tools/devbench.py --steps 6000 --warmup 1504 --batch 8 >> $OUT/ablation.txt 2>&1
done
cat $OUT/ablation.txt
find $OUT -name "*kernel_stats.csv" | head -10
