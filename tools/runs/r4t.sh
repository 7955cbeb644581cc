set -x
OUT=gpurun_out/prof_r04
mkdir -p $OUT && export TMPDIR=/tmp
rm -rf $OUT/stats_c5 $OUT/pmc_c5_rd $OUT/pmc_c5_wr
python bench.py --config c5 > $OUT/bench_c5.json 2>> $OUT/bench.err
python bench.py --config c5 --gpus 2 > $OUT/bench_c5_2ranks.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -- python bench.py --config c5 --steps 30 --warmup 5 --reps 3 --min-region-s 0.05 --no-cpu-baseline > $OUT/stats_c5.log 2>&1
C5="python bench.py --config c5 --steps 2 --warmup 1 --reps 1 --min-region-s 0.05 --preroll-seconds 0 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_c5_rd -- $C5 > $OUT/pmc_c5_rd.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_c5_wr -- $C5 > $OUT/pmc_c5_wr.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
