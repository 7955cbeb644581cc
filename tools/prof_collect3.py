#!/usr/bin/env python3
"""Turns what tools/prof_round3.sh left under gpurun_out/prof_r03 into the committed summaries under profiles/r03_*:
bench lines, rocprofv3 kernel stats per submission mode, concurrency of the overlapped modes (from the kernel trace),
per-launch HBM bytes (FETCH_SIZE / WRITE_SIZE, separate --pmc passes), the SQ wait / issue counters and the per-class
VALU counts of the C3 frame kernel, the ubench issue costs and the ablation timings (inputs of the `valu_issue` block
bench.py carries in its roofline: tools/valu_issue.py turns them into profiles/r03_c3_valu_issue.json)."""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

SRC = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r03"
DST = "profiles"
TAG = sys.argv[2] if len(sys.argv) > 2 else "r03"
ALGO = {"c3": 2440 * 81920, "c3b": 8 * 2440 * 81920, "c2": 4096 * 24576, "c4": 8192 * 49152, "c5": 64 * 2162688}


def newest(pattern):
    files = glob.glob(os.path.join(SRC, pattern), recursive=True)
    return max(files, key=os.path.getmtime) if files else None


def counters(name):
    """-> ({kernel short name: {counter: mean per dispatch}}, {kernel: dispatches})"""
    acc = defaultdict(lambda: defaultdict(list))
    f = newest(f"pmc_{name}/**/*counter_collection.csv")
    if f:
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            short = "frame" if "spectrum_kernel" in k else ("cols" if "big_cols" in k else ("gather" if "big_gather" in k else
                                                             ("rows" if "big_rows" in k else None)))
            if short:
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    # steady state: the MEDIAN dispatch (the first launch after a hold reset also writes the 16384-bin hold trace with
    # up to 4.2 M atomics: 163.7 instead of 159.9 MB at C3)
    def med(v):
        v = sorted(v)
        return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])
    return ({k: {c: med(v) for c, v in d.items()} for k, d in acc.items()},
            {k: len(next(iter(d.values()))) for k, d in acc.items()})


def trace_summary(name):
    """kernel_trace.csv of one bench run -> durations and concurrency of the frame kernel's dispatches"""
    f = newest(f"stats_{name}/**/*kernel_trace.csv")
    if not f:
        return None
    rows = [r for r in csv.DictReader(open(f)) if "spectrum_kernel" in r["Kernel_Name"]]
    if not rows:
        return None
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
    ev = ev[len(ev) // 5:]                      # steady state: drop the first fifth (pre-roll, warm-up, calibration)
    dur = [e - s for s, e in ev]
    # concurrency: integral of (number of dispatches in flight) over time / time with at least one in flight
    pts = sorted([(s, 1) for s, _ in ev] + [(e, -1) for _, e in ev])
    busy = area = 0
    level, last = 0, pts[0][0]
    hist = defaultdict(int)
    for t, d in pts:
        if level > 0:
            busy += t - last
            area += level * (t - last)
        hist[level] += t - last
        level += d
        last = t
    span = ev[-1][1] - ev[0][0]
    return {"dispatches": len(ev), "avg_duration_us": sum(dur) / len(dur) / 1e3, "min_us": min(dur) / 1e3, "max_us": max(dur) / 1e3,
            "avg_in_flight_while_busy": area / busy if busy else 0.0, "gpu_busy_fraction_of_span": busy / span if span else 0.0,
            "time_share_by_kernels_in_flight": {str(k): v / span for k, v in sorted(hist.items()) if span and v / span > 1e-4},
            "grid": rows[-1].get("Grid_Size"), "workgroup": rows[-1].get("Workgroup_Size"),
            "dispatches_per_second_of_span": len(ev) / (span / 1e9) if span else 0.0}


def main():
    os.makedirs(DST, exist_ok=True)
    for c in ("c2", "c3", "c4", "c5", "c5_2ranks", "c3_2ranks", "c4_2ranks"):
        src = os.path.join(SRC, f"bench_{c}.json")
        if os.path.exists(src):
            shutil.copy(src, os.path.join(DST, f"{TAG}_{c}_bench.json"))
    names = {"c3_serial": f"{TAG}_c3_kernel_stats.csv", "c3_batch": f"{TAG}_c3_batch8_kernel_stats.csv",
             "c3_value": f"{TAG}_c3_value_kernel_stats.csv", "c3_streams": f"{TAG}_c3_streams3_kernel_stats.csv",
             "c5": f"{TAG}_c5_kernel_stats.csv"}
    for k, dst in names.items():
        f = newest(f"stats_{k}/**/*kernel_stats.csv")
        if f:
            shutil.copy(f, os.path.join(DST, dst))
    conc = {k: trace_summary(k) for k in ("c3_serial", "c3_batch", "c3_value", "c3_streams")}
    json.dump({"what": "frame-kernel dispatches of `rocprofv3 --kernel-trace -- python bench.py --legs <mode> ...` (tools/prof_round3.sh), "
                       "steady-state part of each run: duration per dispatch and how many dispatches are in flight at once",
               "c3_serial": conc["c3_serial"], "c3_batch8_serial": conc["c3_batch"], "c3_value_batch8_3streams": conc["c3_value"],
               "c3_streams3_one_step_per_launch": conc["c3_streams"]},
              open(os.path.join(DST, f"{TAG}_c3_concurrency.json"), "w"), indent=1)
    lines = [f"# rocprofv3 --pmc passes, {TAG} (tools/prof_round3.sh / prof_round4.sh; FETCH_SIZE and WRITE_SIZE in separate passes)",
             "# FETCH_SIZE is reported in KB and counts 64 B per 128-B request on gfx950 for wide coalesced reads",
             "# (MI355X_MICROARCH.md): the upper bound doubles it; WRITE_SIZE (KB) is 1:1 (calibrated in round 1)."]
    for c in ("c2", "c3", "c3b", "c4", "c5"):
        rd, nrd = counters(f"{c}_rd")
        wr, nwr = counters(f"{c}_wr")
        if not rd or not wr:
            continue
        per_step = {}
        if c == "c5":
            ng = max(1, nrd.get("gather", 1))
            launches = {"cols": nrd.get("cols", 0) / ng, "frame": nrd.get("frame", 0) / ng, "rows": nrd.get("rows", 0) / ng, "gather": 1}
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
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/prof_round3.sh"}
        json.dump(out, open(os.path.join(DST, f"{TAG}_{c}_pmc.json"), "w"), indent=1)
        lines.append(f"{c}: FETCH_SIZE {fetch/1e6:8.1f} MB raw (<= {2*fetch/1e6:8.1f} MB)  WRITE_SIZE {write/1e6:8.1f} MB  "
                     f"algorithmic {ALGO[c]/1e6:8.1f} MB  -> traffic / algorithmic <= {(2*fetch+write)/ALGO[c]:.2f}")
    allsq = {}
    for nm in ("c3_sq", "c3_cls", "c3_cls2"):
        sq, _ = counters(nm)
        allsq.update(sq.get("frame", {}))
    wf = 2440 * 16                                  # wave-frames per launch
    if allsq:
        wc = allsq.get("SQ_WAVE_CYCLES", 1.0)
        lines += ["", "# SQ counters of spectrum_kernel<14,false,1> (C3 shape, 2440 frames / launch), means per launch; cycle counters in quad-cycles"]
        for k, v in sorted(allsq.items()):
            lines.append(f"{k:26s} {v:14.5g}   {v / wc * 100:6.1f} % of SQ_WAVE_CYCLES   {v / wf:9.1f} per wave and frame")
    open(os.path.join(DST, f"{TAG}_pmc.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    for src, dst in (("ubench_valu_rate2.txt", f"{TAG}_ubench_valu.txt"), ("ablation.txt", f"{TAG}_c3_ablation.txt")):
        if os.path.exists(os.path.join(SRC, src)):
            shutil.copy(os.path.join(SRC, src), os.path.join(DST, dst))

    # ---- inputs of the valu_issue block of bench.py's roofline ---------------------------------------------------
    cost = {}
    ub = os.path.join(SRC, "ubench_valu_rate2.txt")
    if os.path.exists(ub):
        for ln in open(ub):
            m = re.match(r"(.+?)\s+([\d.]+) ms\s+([\d.]+) ns per wave-instr per SIMD", ln)
            if m:
                cost[m.group(1).strip()] = float(m.group(3))
    abl = {}
    ab = os.path.join(SRC, "ablation.txt")
    if os.path.exists(ab):
        for ln in open(ab):
            m = re.search(r"lib=libtdsa_(\w+)\.so .* batch=(\d+) .* step=([\d.]+) us", ln)
            if m:
                abl[f"{m.group(1)}_batch{m.group(2)}"] = float(m.group(3))
    if allsq.get("SQ_INSTS_VALU") and cost:
        per_wf = {k: allsq.get(k, 0.0) / wf for k in sorted(allsq) if k.startswith("SQ_INSTS")}
        json.dump({"per_wave_frame": per_wf, "ubench_ns_per_wave_instr_per_simd": cost, "ablation_us_per_step": abl,
                   "sq_cycles": {k: allsq[k] for k in allsq if not k.startswith("SQ_INSTS")}},
                  open(os.path.join(DST, f"{TAG}_c3_valu_inputs.json"), "w"), indent=1)
        print(json.dumps(per_wf, indent=1))
        print(json.dumps(abl))


if __name__ == "__main__":
    main()
