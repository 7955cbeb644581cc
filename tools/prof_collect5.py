#!/usr/bin/env python3
"""Turns what tools/prof_round5.sh left under gpurun_out/prof_r05 into the committed summaries under profiles/r05_*: bench
lines, rocprofv3 kernel stats per configuration / submission mode, concurrency of the overlapped C3 mode, per-launch HBM bytes
(FETCH_SIZE / WRITE_SIZE, separate --pmc passes; every r05_*_pmc.json names the script and the command that produced it),
the emulated strong-scaling table of C5."""
import csv
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import prof_collect3 as pc3  # noqa: E402

SRC = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r05"
TAG = sys.argv[2] if len(sys.argv) > 2 else "r05"
DST = "profiles"
pc3.SRC, pc3.TAG = SRC, TAG
ALGO = {"c3": 2440 * 81920, "c3b": 8 * 2440 * 81920, "c2": 4096 * 24576, "c4": 65536 * 49152, "c5": 64 * 2162688}
CMD = {"c3": "tools/devbench.py --steps 9 --warmup 2 --hold 1", "c3b": "tools/devbench.py --steps 24 --warmup 8 --batch 8 --hold 1",
       "c2": "tools/devbench.py --nfft 4096 --hop 4096 --frames 4096 --mode pow --hold 1",
       "c4": "tools/devbench.py --nfft 8192 --hop 8192 --frames 65536 --steps 5 --hold 1",
       "c5": "bench.py --config c5 --steps 2 --warmup 1 --reps 1 --min-region-s 0.05 --preroll-seconds 0 --no-cpu-baseline --no-parity"}


def main():
    for c in ("c2", "c3", "c4", "c5", "c5_2ranks", "c5_2ranks_peer", "c5_2ranks_captures", "c3_2ranks", "c4_2ranks"):
        src = os.path.join(SRC, f"bench_{c}.json")
        if os.path.exists(src) and os.path.getsize(src) > 10:
            shutil.copy(src, os.path.join(DST, f"{TAG}_{c}_bench.json"))
    names = {"c3_serial": f"{TAG}_c3_kernel_stats.csv", "c3_batch": f"{TAG}_c3_batch8_kernel_stats.csv",
             "c3_value": f"{TAG}_c3_value_kernel_stats.csv", "c2": f"{TAG}_c2_batch8_kernel_stats.csv",
             "c4": f"{TAG}_c4_kernel_stats.csv", "c5": f"{TAG}_c5_kernel_stats.csv"}
    for k, dst in names.items():
        f = pc3.newest(f"stats_{k}/**/*kernel_stats.csv")
        if f:
            shutil.copy(f, os.path.join(DST, dst))
    conc = {k: pc3.trace_summary(k) for k in ("c3_serial", "c3_batch", "c3_value")}
    json.dump({"what": "frame-kernel dispatches of `rocprofv3 --kernel-trace -- python bench.py --legs <mode> ...` (tools/prof_round5.sh), "
                       "steady-state part of each run: duration per dispatch and how many dispatches are in flight at once",
               "c3_serial": conc["c3_serial"], "c3_batch8_serial": conc["c3_batch"], "c3_value_batch8_3streams": conc["c3_value"]},
              open(os.path.join(DST, f"{TAG}_c3_concurrency.json"), "w"), indent=1)
    lines = [f"# rocprofv3 --pmc passes, {TAG} (tools/prof_round5.sh; FETCH_SIZE and WRITE_SIZE in separate passes)",
             "# FETCH_SIZE is reported in KB and counts 64 B per 128-B request on gfx950 for wide coalesced reads",
             "# (MI355X_MICROARCH.md): the upper bound doubles it; WRITE_SIZE (KB) is 1:1 (calibrated in round 1)."]
    for c in ("c2", "c3", "c3b", "c4", "c5"):
        rd, nrd = pc3.counters(f"{c}_rd")
        wr, nwr = pc3.counters(f"{c}_wr")
        if not rd or not wr:
            continue
        per_step = {}
        if c == "c5":
            ng = max(1, nrd.get("gather", 1))
            launches = {"cols": nrd.get("cols", 0) / ng, "rows": nrd.get("rows", 0) / ng, "gather": 1}
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
               "collected_by": "tools/prof_round5.sh + tools/prof_collect5.py",
               "command": f"rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python {CMD[c]}",
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), median dispatch"}
        json.dump(out, open(os.path.join(DST, f"{TAG}_{c}_pmc.json"), "w"), indent=1)
        lines.append(f"{c}: FETCH_SIZE {fetch/1e6:8.1f} MB raw (<= {2*fetch/1e6:8.1f} MB)  WRITE_SIZE {write/1e6:8.1f} MB  "
                     f"algorithmic {ALGO[c]/1e6:8.1f} MB  -> traffic / algorithmic <= {(2*fetch+write)/ALGO[c]:.2f}")
    open(os.path.join(DST, f"{TAG}_pmc.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    # ---- C5 strong scaling: the K-segment captures a rank of a W-GPU run gets, measured on one GPU, + the combine's cost
    sc = os.path.join(SRC, "c5_scaling.txt")
    if os.path.exists(sc):
        best = {}
        for ln in open(sc):
            if ln.startswith("K="):
                k = int(ln.split(":")[0][2:])
                us = float(ln.split(":")[1].split("us")[0])
                best[k] = min(best.get(k, 1e9), us)
        comb = None
        two = os.path.join(SRC, "bench_c5_2ranks.json")
        if os.path.exists(two) and os.path.getsize(two) > 10:
            comb = json.load(open(two))["welch"]
        out = ["# C5 (2^20 points, Welch average of 64 segments): the strong-scaling curve emulated on ONE MI355X",
               "# tools/c5_scaling.py (gpurun_out/prof_r05/c5_scaling.txt): a capture of K segments is what a rank of a W = 64 / K GPU run",
               "# averages when the segments of one capture are sharded (bench.py --config c5 --c5-shard segments)", ""]
        out += [ln.rstrip() for ln in open(sc)]
        out.append("")
        if 64 in best:
            out.append("W GPUs | segments per rank | compute per capture | speed-up (compute only) | efficiency")
            for w in (1, 2, 4, 8):
                k = 64 // w
                if k in best:
                    out.append(f"{w:6d} | {k:17d} | {best[k]:16.1f} us | {best[64] / best[k]:22.2f}x | {100 * best[64] / best[k] / w:8.0f} %")
            if comb:
                c_ms = comb["combine_ms"]
                out += ["", f"Cross-rank combine measured with 2 ranks on this one GPU (profiles/{TAG}_c5_2ranks_bench.json): {c_ms * 1e3:.0f} us per step",
                        f"(rank 0: wait for the partials {comb['rank0_wait_for_partials_ms'] * 1e3:.0f} us, upload of {2} x 4 MiB + combine kernel "
                        f"{comb['rank0_upload_combine_ms'] * 1e3:.0f} us; every rank: export of its 4 MiB float32 partial mean).  Model per step at W ranks:",
                        "  t(W) = compute(64 / W) + export (4 MiB device -> pinned host, ~90 us) + upload on rank 0 (W x 4 MiB host -> device at ~50 GB/s:"
                        " ~85 us x W) + combine kernel (~10 us)"]
                out.append("W GPUs | end to end per capture (model) | speed-up end to end")
                for w in (2, 4, 8):
                    k = 64 // w
                    if k in best:
                        t = best[k] + 90.0 + 85.0 * w + 10.0
                        out.append(f"{w:6d} | {t:27.0f} us | {best[64] / t:18.2f}x")
                out += ["-> sharding the segments of ONE capture never wins end to end over host memory: the partial means (4 MiB per rank and",
                        "   capture) cost more to move than the capture costs to compute.  --c5-shard captures (every rank whole captures,",
                        "   nothing to combine) scales like C3: W x 1 GPU's rate."]
            peer = os.path.join(SRC, "bench_c5_2ranks_peer.json")
            if comb and os.path.exists(peer) and os.path.getsize(peer) > 10:
                pj = json.load(open(peer))
                pw = pj["welch"]
                out += ["", "## peer exchange (bench.py --c5-combine auto | peer; sharding.WelchPeerSlab, tdsa_peer_*, tdsa_welch_export_dev / _combine_dev)",
                        "The partial means stay in device buffers of the ranks' own GPUs; rank 0 maps them through HIP IPC handles and its combine kernel",
                        "reads them in place (other GPUs': over xGMI, each over its own link).  Two ranks on this one GPU, same box as the host-exchange",
                        f"line above (profiles/{TAG}_c5_2ranks_peer_bench.json):",
                        f"  peer: combine {pw['combine_ms']:.3f} ms per step (rank 0: wait {pw['rank0_wait_for_partials_ms']:.3f}, combine kernel "
                        f"{pw['rank0_upload_combine_ms']:.3f}), end to end {pj['ms_per_step']:.3f} ms per capture = {pj['value'] / 1e3:.0f} k segments/s, "
                        f"parity.pass {pj['parity'].get('pass')}",
                        f"  host: combine {comb['combine_ms']:.3f} ms per step (rank 0: wait {comb['rank0_wait_for_partials_ms']:.3f}, upload + kernel "
                        f"{comb['rank0_upload_combine_ms']:.3f})",
                        f"  compute only: {pw['ms_per_step_compute_only']:.3f} ms per capture",
                        "Model per step at W ranks with the peer exchange: compute(64 / W) + the measured export + combine of the same-GPU case, with the",
                        "xGMI reads ASSUMED to add 10-15 us ((W - 1) x 4 MiB over W - 1 links at once; the links' 153 GB/s peak would allow 27 us in all;",
                        "not measured - one GPU per box here):",
                        "W GPUs | end to end per capture (model) | speed-up end to end"]
                ex = pw["combine_ms"] * 1e3 + 12.0
                for w in (2, 4, 8):
                    k = 64 // w
                    if k in best:
                        t = best[k] + ex
                        out.append(f"{w:6d} | {t:27.0f} us | {best[64] / t:18.2f}x")
                for n in (3, 4, 8):
                    f = os.path.join(SRC, f"bench_c5_{n}ranks_peer.json")
                    if os.path.exists(f) and os.path.getsize(f) > 10:
                        j = json.load(open(f))
                        out.append(f"plumbing, {n} ranks sharing this GPU: segments {j['welch']['segments_per_rank']}, exchange {j['welch']['exchange']}, "
                                   f"parity.pass {j['parity'].get('pass')} (max dB error {j['parity'].get('max_db_err_top100dB'):.2e}), combine "
                                   f"{j['welch']['combine_ms']:.3f} ms per step (processes time-slicing one GPU: not a scaling figure)")
        open(os.path.join(DST, f"{TAG}_c5_strong_scaling.txt"), "w").write("\n".join(out) + "\n")
        print("\n".join(out[-30:]))
    chirp_stats()


def chirp_stats():
    """profiles/<TAG>_chirp_kernel_stats.txt: per-kernel time per call and HBM traffic of three chirp-z plans"""
    import collections
    cfg = {"n1000": "N = 1000 = 2^3 5^3, 4096 frames per call (mixed-radix transform of N points, tdsa_smooth.hip)",
           "n1021": "N = 1021 (a prime), 4096 frames per call (chirp-z, M = 2048: the whole convolution in one launch)",
           "n20000": "N = 20 000 = 2^5 5^4, 512 frames per call (mixed radix in two passes: 125-point columns, 160-point rows)",
           "n20011": "N = 20 011 (a prime), 512 frames per call (chirp-z, M = 65536: columns, rows of both transforms in one kernel, columns out)",
           "n1000000": "N = 1 000 000 = 2^6 5^6, 10 frames per call (mixed radix in two passes: 125-point columns, 8000-point rows)",
           "n999983": "N = 999 983 (a prime), 10 frames per call (chirp-z split plan: four half-length sub-convolutions of 2^20 points)"}
    out = ["# rocprofv3 --kernel-trace --stats around tools/devbench.py (tools/prof_round5.sh), per-kernel average duration per call;",
           "# one devbench step = one tdsa_process_dev call with --hold 1 (max-hold trace folded)"]
    for n in cfg:
        f = pc3.newest(f"stats_chirp_{n}/**/*kernel_stats.csv")
        if not f:
            continue
        out.append(f"\n## {cfg[n]}")
        rows = [r for r in csv.DictReader(open(f)) if "tdsa" in r["Name"] and "fill_kernel" not in r["Name"]]
        steps = min(int(r["Calls"]) for r in rows)
        tot = 0.0
        for r in rows:
            per = float(r["TotalDurationNs"]) / steps / 1e3
            tot += per
            out.append(f"  {r['Name'].split('(')[0][:62]:62s} {int(r['Calls']) // steps} x {float(r['AverageNs']) / 1e3:7.1f} us = {per:7.1f} us per call")
        out.append(f"  sum of kernels {tot:7.1f} us per call")
    out.append("\n## HBM traffic per call (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; KB units; FETCH_SIZE counts 64 B per 128-B "
               "request for wide reads: the upper bound doubles it)")
    for n, algo in (("n1000", 4096 * 6 * 1000), ("n1021", 4096 * 6 * 1021), ("n20000", 512 * 6 * 20000), ("n20011", 512 * 6 * 20011)):
        tot = {}
        for kind, ctr in (("rd", "FETCH_SIZE"), ("wr", "WRITE_SIZE")):
            f = pc3.newest(f"pmc_chirp_{n}_{kind}/**/*counter_collection.csv")
            if not f:
                continue
            per = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == ctr:
                    per[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
            s = 0.0
            for k, v in per.items():
                if "fill" in k or "rocclr" in k:
                    continue
                v = sorted(v)
                s += v[len(v) // 2]
            tot[kind] = s * 1024
        if len(tot) == 2:
            out.append(f"  {n}: FETCH_SIZE {tot['rd'] / 1e6:7.1f} MB raw (<= {2 * tot['rd'] / 1e6:7.1f})  WRITE_SIZE {tot['wr'] / 1e6:7.1f} MB   "
                       f"algorithmic (2 N in + 4 N out per frame) {algo / 1e6:7.1f} MB")
    if len(out) > 3:
        open(os.path.join(DST, f"{TAG}_chirp_kernel_stats.txt"), "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
