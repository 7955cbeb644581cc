#!/bin/bash
# round 5, run 1: parity of the long-frame paths with the in-kernel cosine window + lazy Welch mean; K-scaling with per-kernel
# traces; table window against cosine window, alternating processes
set -x
OUT=gpurun_out/a1; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "c5 or long or big or chirp or welch or any_size" > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
for rep in 1 2; do
  python tools/c5_scaling.py --steps 100 --reps 1 --ks 8,16,32,64 > $OUT/scale_cos_$rep.txt 2>&1
  TDSA_HIP_LIB=$PWD/topdogspectrumanalyser_amd/libtdsa_dev.so TDSA_BIG_WIN_TABLE=1 python tools/c5_scaling.py --steps 100 --reps 1 --ks 8,16,32,64 > $OUT/scale_tab_$rep.txt 2>&1
done
for K in 8 64; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_cos_$K -- python tools/c5_scaling.py --steps 60 --reps 1 --ks $K > $OUT/trace_cos_$K.log 2>&1
  python tools/c5_trace.py $OUT/trace_cos_$K > $OUT/trace_cos_$K.txt 2>&1
  TDSA_HIP_LIB=$PWD/topdogspectrumanalyser_amd/libtdsa_dev.so TDSA_BIG_WIN_TABLE=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_tab_$K -- python tools/c5_scaling.py --steps 60 --reps 1 --ks $K > $OUT/trace_tab_$K.log 2>&1
  python tools/c5_trace.py $OUT/trace_tab_$K > $OUT/trace_tab_$K.txt 2>&1
done
cat $OUT/scale_*.txt $OUT/trace_*.txt
find $OUT -name "*.csv" -size +3M -delete; find $OUT -name "*.db" -delete
