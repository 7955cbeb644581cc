#!/usr/bin/env python3
"""Per-kernel durations and the gaps between the kernels of the long-frame chain, from a rocprofv3 --kernel-trace csv:
tools/c5_trace.py <dir with *kernel_trace.csv> [skip_first_steps]
A step = column pass -> row pass -> gather + finish; prints the averages over the steps of the trace."""
import csv
import glob
import sys

import numpy as np


def main():
    files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            kind = "cols" if "big_cols" in n else "rows" if "big_rows" in n else "gather" if "big_gather" in n else None
            if kind:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind))
    rows.sort()
    steps, cur = [], []
    for s, e, k in rows:
        if k == "cols" and cur:
            steps.append(cur)
            cur = []
        cur.append((s, e, k))
    if cur:
        steps.append(cur)
    steps = [st for st in steps if [k for _, _, k in st] == ["cols", "rows", "gather"]][skip:]
    if not steps:
        print("no complete steps")
        return
    d = {k: np.array([st[i][1] - st[i][0] for st in steps]) / 1e3 for i, k in enumerate(("cols", "rows", "gather"))}
    g1 = np.array([st[1][0] - st[0][1] for st in steps]) / 1e3
    g2 = np.array([st[2][0] - st[1][1] for st in steps]) / 1e3
    period = np.diff(np.array([st[0][0] for st in steps])) / 1e3
    g3 = np.array([steps[i + 1][0][0] - steps[i][2][1] for i in range(len(steps) - 1)]) / 1e3
    med = np.median
    print(f"{len(steps)} steps: cols {med(d['cols']):.1f}  gap {med(g1):.1f}  rows {med(d['rows']):.1f}  gap {med(g2):.1f}  "
          f"gather {med(d['gather']):.1f}  gap to next {med(g3):.1f}  | kernels {med(d['cols']) + med(d['rows']) + med(d['gather']):.1f}  "
          f"period {med(period):.1f} us (medians)")


if __name__ == "__main__":
    main()
