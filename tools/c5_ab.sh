#!/bin/bash
# same-box alternating A/B of the long-frame chain's kernels over library variants: tools/c5_ab.sh <reps> lib1 lib2 ...
# (rocprofv3 --kernel-trace --stats around bench.py --config c5; prints per variant the per-kernel averages of every repetition)
REPS=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/c5ab
rm -rf $OUT && mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
for rep in $(seq 1 $REPS); do for lib in "$@"; do
  TDSA_HIP_LIB=$GRAFT_REPO_ROOT/topdogspectrumanalyser_amd/libtdsa_$lib.so rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${lib}_$rep -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 30 --warmup 5 --reps 3 --min-region-s 0.05 --no-cpu-baseline > $OUT/${lib}_$rep.log 2>&1
done; done
cd $GRAFT_REPO_ROOT && python - "$@" <<'P'
import csv, glob, sys
for lib in sys.argv[1:]:
    out = []
    for d in sorted(glob.glob(f"gpurun_out/c5ab/{lib}_*/")):
        f = glob.glob(d + "*/*kernel_stats.csv")[0]
        t = {}
        for r in list(csv.DictReader(open(f)))[:3]:
            n = r["Name"]
            t["cols" if "big_cols" in n else ("rows" if ("big_rows" in n or "spectrum_kernel" in n) else "gather")] = float(r["AverageNs"]) / 1e3
        out.append(f"rows {t.get('rows', 0):6.1f} cols {t.get('cols', 0):6.1f}")
    print(f"{lib:10s}", " | ".join(out))
P
