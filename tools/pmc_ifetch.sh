#!/bin/bash
# instruction-fetch side of the C3 frame kernel: production library against the timing-only VALU builds
OUT=gpurun_out/ifetch
rm -rf $OUT && mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
L=$R/topdogspectrumanalyser_amd
for lib in hip abl6 abl7; do
  for set in "SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_VALU" \
             "SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQC_TC_INST_REQ SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    tag=$(echo $set | cut -c1-12 | tr ' ' '_')
    TDSA_HIP_LIB=$L/libtdsa_$lib.so rocprofv3 --pmc $set --output-format csv -d $R/$OUT/${lib}_$tag -- python $R/tools/devbench.py --steps 12 --warmup 4 --batch 8 > $R/$OUT/${lib}_$tag.log 2>&1
  done
done
cd $R
python - <<'P'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/ifetch/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "spectrum_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(d, {k: sorted(v)[len(v)//2] for k, v in acc.items()})
P
