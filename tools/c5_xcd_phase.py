#!/usr/bin/env python3
"""Developer experiment (round-4 verdict, lever c): the column pass of the long-frame chain runs in one of two "modes of a
process" (116 or 122 us per 64 segments, the row pass trading the other way).  Is it the XCD a workgroup lands on?  The
dispatcher hands workgroups to the eight XCDs round robin and every launch of the chain has a multiple of eight of them,
so the phase a process starts with stays.  Here k = 0 .. 8 empty workgroups are launched ahead of every column pass
(tdsa_debug_knob big_pre_wgs) inside ONE process, alternating; the row-pass time comes from the plan's profiling events.
Needs a developer build of the library (tools/build_variants.sh dev "-DTDSA_DEV", TDSA_HIP_LIB=...): the shipped one has no
such knob.
python tools/c5_xcd_phase.py [--steps 120] [--rounds 3]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from topdogspectrumanalyser_amd import SpectrumEngine, _native as nat  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    n, K = 1 << 20, 64
    iq = np.random.default_rng(0).integers(-100, 100, size=2 * n * K, dtype=np.int8)
    di, do = C.c_void_p(), C.c_void_p()
    nat.check(nat.lib.tdsa_dev_alloc(0, iq.nbytes * 2, C.byref(di)))
    nat.check(nat.lib.tdsa_dev_alloc(0, n * 4 * 2, C.byref(do)))
    for r in range(2):
        nat.check(nat.lib.tdsa_memcpy_h2d(0, C.c_void_p(di.value + r * iq.nbytes), iq.ctypes.data_as(C.c_void_p), iq.nbytes))
    e = SpectrumEngine(n, max_frames=K)
    e.set_window(np.hanning(n).astype(np.float32))
    e.configure(db_mode="pow", power_scale=1.0, log_floor=1e-12, dc_alpha=-1.0, avg=("lin", K), cal_offset_db=-0.8087)

    def run(steps):
        for i in range(steps):
            r = i % 2
            e.reset(nat.RESET_AVG)
            e.process_device(nat.IN_I8, di.value + r * iq.nbytes, n * K, n, K, do.value + r * n * 4)
        e.synchronize()
    run(30)
    for rnd in range(a.rounds):
        for k in (0, 1, 2, 3, 4, 5, 6, 7, 8):
            e.debug_knob("big_pre_wgs", k)
            run(10)
            t0 = time.perf_counter()
            run(a.steps)
            total = (time.perf_counter() - t0) / a.steps * 1e6
            e.profile_enable(True)
            run(40)
            launches, ms = e.profile_read()
            e.profile_enable(False)
            print(f"round {rnd} pre_wgs {k}: capture {total:6.1f} us  row pass {ms * 1e3 / max(1, launches):6.1f} us", flush=True)


if __name__ == "__main__":
    main()
