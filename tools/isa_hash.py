#!/usr/bin/env python3
"""Opcode-sequence fingerprint of kernels in a gfx950 assembly listing (hipcc -S --cuda-device-only):
tools/isa_hash.py file.s [name-substring]  ->  one line per kernel: instructions, sha256[:16] of the mnemonic sequence.
Operands, labels, comments and directives are left out: register allocation details and kernarg offsets may move, the
instruction stream may not (tests/test_isa_frozen.py pins the BASELINE instantiations of the frame kernel with it)."""
import hashlib
import re
import sys


def kernels(path):
    out, name, ops = {}, None, []
    for ln in open(path):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, ops = m.group(1), []
            continue
        if name is None:
            continue
        if ln.startswith(".Lfunc_end"):
            out[name] = ops
            name = None
            continue
        t = ln.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        ops.append(t.split()[0])
    return out


def fingerprint(ops):
    return len(ops), hashlib.sha256("\n".join(ops).encode()).hexdigest()[:16]


if __name__ == "__main__":
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    for k, ops in kernels(sys.argv[1]).items():
        if sub in k:
            n, h = fingerprint(ops)
            print(f"{k} {n} {h}")
