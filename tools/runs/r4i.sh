timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4i.log 2>&1
tail -6 gpurun_out/r4i.log | cut -c1-200
