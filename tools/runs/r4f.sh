bash tools/c5_groups.sh "64" hip c5 c6 c7 2>&1 | tail -10 > gpurun_out/r4f_cols.txt
cat gpurun_out/r4f_cols.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c5 or long or welch or big" 2>&1 | tail -3
