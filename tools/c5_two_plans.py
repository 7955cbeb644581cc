#!/usr/bin/env python3
"""Developer experiment: two independent long-frame plans (C5 shape: 64 segments of 2^20 points, Welch) on one GPU, each on
its own stream - does the column pass of one capture (write dominated) run beside the row pass of the other (read dominated)
better than the two in series?  Prints us per capture for 1 plan and for 2 plans alternating.
python tools/c5_two_plans.py [--steps 300]"""
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
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--plans", type=int, default=2)
    ap.add_argument("--masks", default="", help="developer build only (TDSA_HIP_LIB=libtdsa_dev.so): tdsa_debug_knob cu_mask per "
                                                "plan, comma separated: 1 / 2 = mask words 0-3 / 4-7, 3 / 4 = every second CU")
    ap.add_argument("--num-cu", type=int, default=0, help="tdsa_debug_knob num_cu on every plan (persistent grids)")
    ap.add_argument("--knob", action="append", default=[], help="name=value for tdsa_debug_knob on every plan (repeatable)")
    a = ap.parse_args()
    masks = a.masks.split(",") if a.masks else []
    n, K = 1 << 20, 64
    ns = n * K
    iq = np.random.default_rng(0).integers(-100, 100, size=2 * ns, dtype=np.int8)
    ring = 4
    di, do = C.c_void_p(), C.c_void_p()
    nat.check(nat.lib.tdsa_dev_alloc(0, iq.nbytes * ring, C.byref(di)))
    nat.check(nat.lib.tdsa_dev_alloc(0, n * 4 * ring * a.plans, C.byref(do)))
    for r in range(ring):
        nat.check(nat.lib.tdsa_memcpy_h2d(0, C.c_void_p(di.value + r * iq.nbytes), iq.ctypes.data_as(C.c_void_p), iq.nbytes))
    engs = []
    for k in range(a.plans):
        e = SpectrumEngine(n, max_frames=K)
        for kv in a.knob:
            e.debug_knob(kv.split('=')[0], int(kv.split('=')[1]))
        if masks:
            e.debug_knob("cu_mask", int(masks[k % len(masks)]))
        if a.num_cu:
            e.debug_knob("num_cu", a.num_cu)
        e.set_window(np.hanning(n).astype(np.float32))
        e.configure(db_mode="pow", power_scale=1.0, log_floor=1e-12, dc_alpha=-1.0, avg=("lin", K), cal_offset_db=-0.8087)
        engs.append(e)

    def step(e, i, slot):
        r = i % ring
        e.reset(nat.RESET_AVG)
        e.process_device(nat.IN_I8, di.value + r * iq.nbytes, ns, n, K, do.value + (slot * ring + r) * n * 4)

    def run(active, steps):
        for e in engs:
            e.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            for s, e in enumerate(engs[:active]):
                step(e, i, s)
        for e in engs[:active]:
            e.synchronize()
        return (time.perf_counter() - t0) / (steps * active) * 1e6

    for active in range(1, a.plans + 1):
        run(active, a.warmup)
    for rep in range(3):
        for active in range(1, a.plans + 1):
            print(f"{active} plan(s) in flight: {run(active, a.steps):.1f} us per capture of 64 segments", flush=True)


if __name__ == "__main__":
    main()
