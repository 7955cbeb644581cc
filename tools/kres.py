#!/usr/bin/env python3
"""Register / scratch report of every frame-kernel instantiation (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kres.py [log2n ...] [-D...]   -> one line per kernel: size, format, hold, VGPRs, scratch, occupancy"""
import concurrent.futures
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CSRC = os.path.join(ROOT, "topdogspectrumanalyser_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def report(log2n, extra=()):
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", *extra,
               f"-DTDSA_LOG2N={log2n}", "-Rpass-analysis=kernel-resource-usage", "-c",
               os.path.join(CSRC, "tdsa_spectrum_inst.hip"), "-o", os.path.join(tmp, "x.o")]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-3000:])
    kernels, cur = {}, None
    for ln in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+?):\s+(\S+)\s+\[-Rpass", ln)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    return kernels


def main():
    sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or list(range(6, 15))
    extra = [a for a in sys.argv[1:] if a.startswith("-")]
    with concurrent.futures.ThreadPoolExecutor(min(len(sizes), 8)) as ex:
        reps = dict(zip(sizes, ex.map(lambda k: report(k, extra), sizes)))
    for k in sizes:
        for name, r in sorted(reps[k].items()):
            m = re.search(r"spectrum_kernelILi(\d+)ELb(\d)ELi(\d)E()", name)
            if not m:
                continue
            print(f"N=2^{m.group(1):>2} c64={m.group(2)} hold={m.group(3)} acc={m.group(4)}  VGPRs {r['VGPRs']:>3}  "
                  f"scratch {r['ScratchSize [bytes/lane]']:>3}  occ {r['Occupancy [waves/SIMD]']}  "
                  f"spillV {r['VGPRs Spill']} spillS {r['SGPRs Spill']}")


if __name__ == "__main__":
    main()
