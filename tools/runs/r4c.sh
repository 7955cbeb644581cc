mkdir -p gpurun_out/r4c
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r4c/pytest.log
cat gpurun_out/r4c/pytest.log
R=$PWD; OUT=$R/gpurun_out/r4c; cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do for old in 0 1; do
  TDSA_BIG_ROWS_OLD=$old rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rows_old${old}_$rep -- python $R/bench.py --config c5 --steps 30 --warmup 5 --reps 3 --min-region-s 0.05 --no-cpu-baseline > $OUT/rows_old${old}_$rep.json 2> $OUT/rows_old${old}_$rep.err
  f=$(ls $OUT/rows_old${old}_$rep/*/*kernel_stats.csv | head -1)
  echo "== old=$old rep=$rep $(grep -o '"ms_per_step": [0-9.]*' $OUT/rows_old${old}_$rep.json | head -1)"; head -4 $f | cut -d, -f1-4 | cut -c1-120
done; done
