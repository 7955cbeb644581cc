#!/bin/bash
OUT=gpurun_out/r3k
rm -rf $OUT && mkdir -p $OUT && export TMPDIR=/tmp
L=$PWD/topdogspectrumanalyser_amd
for rep in 1 2 3; do for lib in hip early; do
  TDSA_HIP_LIB=$L/libtdsa_$lib.so python bench.py --config c5 --no-cpu-baseline > $OUT/b.json 2>/dev/null; echo -n "$lib "; python -c "import json; print(json.load(open('$OUT/b.json'))['ms_per_step'])"
done; done
TDSA_HIP_LIB=$L/libtdsa_early.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "c5 or long or big or welch" 2>&1 | tail -2
