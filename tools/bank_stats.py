#!/usr/bin/env python3
"""VGPR bank conflicts among the source operands of the VALU instructions of one kernel's loop (developer tool).
usage: bank_stats.py file.s mangled_kernel_name      (bank of vN = N mod 4)

What costs on gfx950 (tools/ubench/valu_bank.hip): two sources in one bank are free for v_add / v_mul and for VOP3 v_fma;
THREE sources of a v_fma in one bank, or src0 and src1 of a v_fmac in one bank, double the instruction's issue time.  The
"penalised" column counts those two patterns; "conflicts" counts any two distinct sources in one bank."""
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


def loop_body(lines):
    labels = {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            labels[m.group(1)] = i
    best = (0, 0, 0)
    for i, ln in enumerate(lines):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", ln) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[0]:
            best = (i - labels[m.group(1)], labels[m.group(1)], i)
    return lines[best[1]:best[2] + 1]


def sources(op, args):
    """VGPR numbers read by the instruction (single registers only)"""
    regs = []
    for a in args:
        m = re.fullmatch(r"[-|]*v(\d+)\|?", a.strip())
        regs.append(int(m.group(1)) if m else None)
    if not regs:
        return []
    dst, src = regs[0], regs[1:]
    if op.startswith(("v_fmac", "v_mac")):
        src = src + [dst]
    return [r for r in src if r is not None]


def main():
    lines = loop_body(kernel_lines(sys.argv[1], sys.argv[2]))
    stat = collections.defaultdict(lambda: [0, 0, 0, 0])
    for ln in lines:
        s = ln.split(";")[0].strip()
        if not s.startswith("v_") or s.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_permlane")):
            continue
        op, _, rest = s.partition(" ")
        rest = re.sub(r"\b(quad_perm|row_\w+|bank_mask|row_mask|bound_ctrl|op_sel\w*|neg_\w+|clamp|mul:\d|div:\d):?\S*", "", rest)
        args = [a for a in rest.split(",") if a.strip()]
        src = sources(op, args)
        banks = collections.Counter(r % 4 for r in set(src))
        conflict = any(c > 1 for c in banks.values())
        repeat = len(src) != len(set(src))
        pen = (op.startswith("v_fma_f32") and len(set(src)) == 3 and len(banks) == 1) or \
              (op.startswith("v_fmac") and len(src) >= 2 and src[0] != src[1] and src[0] % 4 == src[1] % 4)
        key = op
        stat[key][0] += 1
        stat[key][1] += conflict
        stat[key][2] += repeat
        stat[key][3] += pen
    tot = [sum(v[i] for v in stat.values()) for i in range(4)]
    print(f"VALU instructions in the loop body: {tot[0]}, source-bank conflicts: {tot[1]} ({100*tot[1]/tot[0]:.1f} %), repeated source register: {tot[2]}, penalised patterns: {tot[3]}")
    for k, v in sorted(stat.items(), key=lambda kv: -kv[1][0]):
        if v[0] >= 8:
            print(f"  {k:28s} {v[0]:5d}  conflicts {v[1]:4d} ({100*v[1]/v[0]:5.1f} %)  repeats {v[2]}  penalised {v[3]}")


if __name__ == "__main__":
    main()
