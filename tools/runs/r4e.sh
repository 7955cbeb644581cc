bash tools/c5_groups.sh "64" hip c3 c4 2>&1 | tail -8 > gpurun_out/r4e_cols.txt
cat gpurun_out/r4e_cols.txt
mkdir -p gpurun_out/r4e
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r4e/pytest.log; cat gpurun_out/r4e/pytest.log
for old in 1 0; do for mode in "exp 4" "lin 16"; do
  echo "== TDSA_AVG_OLD=$old avg $mode" >> gpurun_out/r4e/avg.txt
  if [ $old = 1 ]; then export TDSA_AVG_OLD=1; else unset TDSA_AVG_OLD; fi
  timeout 300 python tools/avgbench.py --avg $mode 2>&1 | tail -3 >> gpurun_out/r4e/avg.txt
done; done
cat gpurun_out/r4e/avg.txt
