#!/bin/bash
OUT=gpurun_out/fifo
rm -rf $OUT && mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
L=$R/topdogspectrumanalyser_amd
for set in "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
    tag=$(echo $set | cut -c1-14 | tr ' ' '_')
    rocprofv3 --pmc $set --output-format csv -d $R/$OUT/$tag -- python $R/tools/devbench.py --steps 12 --warmup 4 --batch 8 > $R/$OUT/$tag.log 2>&1
done
cd $R
python - <<'P'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/fifo/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "spectrum_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(d, {k: sorted(v)[len(v)//2] for k, v in acc.items()})
P
bash tools/ab_quick.sh stag stag2
