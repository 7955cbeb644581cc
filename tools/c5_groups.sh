#!/bin/bash
# per-kernel times of the long-frame chain (C5) against segments per round and library variant:
#   tools/c5_groups.sh "64 16" hip colsA ...     (rocprofv3 --kernel-trace --stats around bench.py --config c5)
OUT=$GRAFT_REPO_ROOT/gpurun_out/c5grp
rm -rf $OUT && mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
GROUPS_=$1; shift
for lib in "$@"; do for g in $GROUPS_; do
  TDSA_HIP_LIB=$GRAFT_REPO_ROOT/topdogspectrumanalyser_amd/libtdsa_$lib.so TDSA_BIG_GROUP=$g rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${lib}_g$g -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 30 --warmup 5 --reps 3 --min-region-s 0.05 --no-cpu-baseline > $OUT/${lib}_g$g.log 2>&1
  f=$(ls $OUT/${lib}_g$g/*/*kernel_stats.csv | head -1)
  echo "== $lib, $g segments per round: $(grep -o '"ms_per_step": [0-9.]*' $OUT/${lib}_g$g.log | head -1)  parity $(grep -o '"pass": [a-z]*' $OUT/${lib}_g$g.log | tail -1)"

done; done
cd $GRAFT_REPO_ROOT && python - <<'P'
import csv, glob
for d in sorted(glob.glob("gpurun_out/c5grp/*_g*/")):
    f = glob.glob(d + "*/*kernel_stats.csv")[0]
    out = []
    for r in list(csv.DictReader(open(f)))[:3]:
        n = r["Name"]
        short = "cols" if "big_cols" in n else ("rows" if "spectrum_kernel" in n else ("gather" if "gather" in n else n[:20]))
        out.append(f"{short} {int(r['Calls'])} x {float(r['AverageNs'])/1e3:.1f} us")
    print(d.split("/")[-2], " | ".join(out))
P
