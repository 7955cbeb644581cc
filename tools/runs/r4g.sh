mkdir -p gpurun_out/r4g
R=$PWD; OUT=$R/gpurun_out/r4g
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "averag or avg or golden_batch or workgroup_chunks" 2>&1 | tail -4
cd /tmp; export TMPDIR=/tmp
for mode in "exp 4" "lin 16"; do
  tag=$(echo $mode | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/avg_$tag -- python $R/tools/avgbench.py --avg $mode --steps 1000 --warmup 100 > $OUT/avg_$tag.txt 2>&1
  tail -1 $OUT/avg_$tag.txt
  f=$(ls $OUT/avg_$tag/*/*kernel_stats.csv | head -1); head -7 $f | cut -d, -f1-4 | cut -c1-150
done
TDSA_AVG_OLD=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/avg_old -- python $R/tools/avgbench.py --avg exp 4 --steps 1000 --warmup 100 > $OUT/avg_old.txt 2>&1
tail -1 $OUT/avg_old.txt
f=$(ls $OUT/avg_old/*/*kernel_stats.csv | head -1); head -7 $f | cut -d, -f1-4 | cut -c1-150
