#!/usr/bin/env python3
"""Turns what tools/prof_round.sh left under gpurun_out/prof_r02 into the committed summaries under
profiles/r02_*: bench lines, rocprofv3 kernel stats, per-launch HBM bytes (FETCH_SIZE / WRITE_SIZE, separate
--pmc passes) and the SQ wait / issue counters of the C3 frame kernel."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

SRC = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r02"
DST = "profiles"
ALGO = {"c3": 2440 * 81920, "c2": 4096 * 24576, "c4": 8192 * 49152, "c5": 64 * 2162688}


def counters(name):
    """-> ({kernel short name: {counter: mean per dispatch}}, {kernel: dispatches})"""
    acc = defaultdict(lambda: defaultdict(list))
    files = glob.glob(os.path.join(SRC, f"pmc_{name}", "**", "*counter_collection.csv"), recursive=True)
    for f in sorted(files, key=os.path.getmtime)[-1:]:       # the newest pass only (gpurun merges old ones back in)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            short = "frame" if "spectrum_kernel" in k else ("cols" if "big_cols" in k else ("gather" if "big_gather" in k else None))
            if short:
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return ({k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()},
            {k: len(next(iter(d.values()))) for k, d in acc.items()})


def main():
    os.makedirs(DST, exist_ok=True)
    for c in ("c2", "c3", "c4", "c5"):
        shutil.copy(os.path.join(SRC, f"bench_{c}.json"), os.path.join(DST, f"r02_{c}_bench.json"))
    for c in ("c3", "c5"):
        f = max(glob.glob(os.path.join(SRC, f"stats_{c}", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
        shutil.copy(f, os.path.join(DST, f"r02_{c}_kernel_stats.csv"))
    lines = ["# rocprofv3 --pmc passes, round 2 (tools/prof_round.sh; FETCH_SIZE and WRITE_SIZE in separate passes)",
             "# FETCH_SIZE is reported in KB and counts 64 B per 128-B request on gfx950 for wide coalesced reads",
             "# (MI355X_MICROARCH.md): the upper bound doubles it; WRITE_SIZE (KB) is 1:1 (calibrated in round 1)."]
    for c in ("c2", "c3", "c4", "c5"):
        rd, nrd = counters(f"{c}_rd")
        wr, nwr = counters(f"{c}_wr")
        per_step = {}
        if c == "c5":       # two column launches + two row launches + one gather per step (64 segments)
            launches = {"cols": 2, "frame": 2, "gather": 1}
            fetch = sum(rd.get(k, {}).get("FETCH_SIZE", 0.0) * n for k, n in launches.items()) * 1024
            write = sum(wr.get(k, {}).get("WRITE_SIZE", 0.0) * n for k, n in launches.items()) * 1024
            per_step = {k: {"fetch_kb_per_launch": rd.get(k, {}).get("FETCH_SIZE"),
                            "write_kb_per_launch": wr.get(k, {}).get("WRITE_SIZE"), "launches_per_step": n}
                        for k, n in launches.items()}
        else:
            fetch = rd["frame"]["FETCH_SIZE"] * 1024
            write = wr["frame"]["WRITE_SIZE"] * 1024
        out = {"config": c, "kernel": "spectrum_kernel" if c != "c5" else "cols + rows + gather (one 64-segment step)",
               "fetch_bytes_raw": fetch, "fetch_bytes_upper": 2 * fetch, "write_bytes": write,
               "algorithmic_bytes": ALGO[c], "traffic_over_algorithmic": (2 * fetch + write) / ALGO[c],
               "dispatches_averaged": {"read_pass": nrd, "write_pass": nwr}, "per_kernel": per_step,
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/prof_round.sh"}
        json.dump(out, open(os.path.join(DST, f"r02_{c}_pmc.json"), "w"), indent=1)
        lines.append(f"{c}: FETCH_SIZE {fetch/1e6:8.1f} MB raw (<= {2*fetch/1e6:8.1f} MB)  WRITE_SIZE {write/1e6:8.1f} MB  "
                     f"algorithmic {ALGO[c]/1e6:8.1f} MB  -> traffic / algorithmic <= {(2*fetch+write)/ALGO[c]:.2f}")
    sq, nsq = counters("c3_sq")
    if "frame" in sq:
        s = sq["frame"]
        wc = s.get("SQ_WAVE_CYCLES", 1.0)
        lines.append("")
        lines.append("# SQ counters of spectrum_kernel<14,false,1> (C3 shape, 2440 frames / launch), means per launch, quad-cycles")
        for k, v in sorted(s.items()):
            lines.append(f"{k:24s} {v:14.4g}   {v / wc * 100:6.1f} % of SQ_WAVE_CYCLES")
        if "SQ_INSTS_VALU" in s:
            lines.append(f"SQ_INSTS_VALU per wave per frame = {s['SQ_INSTS_VALU'] / (2440 * 16):.0f}")
    open(os.path.join(DST, "r02_pmc.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
