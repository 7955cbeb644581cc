#!/usr/bin/env python3
"""Instruction histogram of one kernel in a hipcc -save-temps .s file (developer tool).

usage: isa_hist.py file.s mangled_kernel_name [--loop]
--loop: restrict to the largest backward-branch loop body (the persistent frame loop)."""
import collections
import re
import sys


def kernel_lines(path, name):
    out, on = [], False
    for ln in open(path):
        if ln.startswith(name + ":"):
            on = True
            continue
        if on:
            if ln.startswith(".Lfunc_end"):
                break
            out.append(ln.rstrip("\n"))
    return out


def main():
    path, name = sys.argv[1], sys.argv[2]
    lines = kernel_lines(path, name)
    if "--loop" in sys.argv:
        labels = {}
        for i, ln in enumerate(lines):
            m = re.match(r"^(\.LBB\d+_\d+):", ln)
            if m:
                labels[m.group(1)] = i
        best = (0, 0, 0)
        for i, ln in enumerate(lines):
            m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", ln) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", ln)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                if i - labels[m.group(1)] > best[0]:
                    best = (i - labels[m.group(1)], labels[m.group(1)], i)
        lines = lines[best[1]:best[2] + 1]
        print(f"loop body: lines {best[1]}..{best[2]}")
    hist = collections.Counter()
    for ln in lines:
        s = ln.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        hist[op] += 1
    tot = sum(hist.values())
    valu = sum(c for o, c in hist.items() if o.startswith("v_"))
    print(f"total {tot}  valu {valu}  salu {sum(c for o, c in hist.items() if o.startswith('s_'))} "
          f"ds {sum(c for o, c in hist.items() if o.startswith('ds_'))} "
          f"vmem {sum(c for o, c in hist.items() if o.startswith(('buffer_', 'global_', 'scratch_', 'flat_')))}")
    for o, c in hist.most_common():
        print(f"{c:6d}  {o}")


if __name__ == "__main__":
    main()
