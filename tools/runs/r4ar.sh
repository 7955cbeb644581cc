#!/bin/bash
mkdir -p gpurun_out/r4ar; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4ar/prof -- python $GRAFT_REPO_ROOT/tools/dcbench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python - <<'P'
import csv,glob
f=glob.glob('gpurun_out/r4ar/prof/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3)
P
