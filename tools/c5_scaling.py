#!/usr/bin/env python3
"""Developer experiment: the long-frame chain (2^20-point Welch) over K segments per capture, K = 8 ... 128 - what of a C5 step
is per-segment work and what is per-launch fill / drain (the K / W segments a rank of a W-GPU strong-scaling run gets).
Prints us per capture, a least-squares line and the strong-scaling speed-ups that follow for one capture of --capture segments.
python tools/c5_scaling.py [--steps 200] [--ks 8,16,32,64]"""
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
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ks", default="8,16,32,64,128")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--capture", type=int, default=64, help="segments of the capture whose strong-scaling curve is predicted")
    ap.add_argument("--knob", action="append", default=[], help="name=value for tdsa_debug_knob on every plan (repeatable)")
    a = ap.parse_args()
    ks = [int(k) for k in a.ks.split(",")]
    n, kmax = 1 << 20, max(ks)
    iq = np.random.default_rng(0).integers(-100, 100, size=2 * n * kmax, dtype=np.int8)
    ring = 2
    di, do = C.c_void_p(), C.c_void_p()
    nat.check(nat.lib.tdsa_dev_alloc(0, iq.nbytes * ring, C.byref(di)))
    nat.check(nat.lib.tdsa_dev_alloc(0, n * 4 * ring, C.byref(do)))
    for r in range(ring):
        nat.check(nat.lib.tdsa_memcpy_h2d(0, C.c_void_p(di.value + r * iq.nbytes), iq.ctypes.data_as(C.c_void_p), iq.nbytes))
    pts = []
    for rep in range(a.reps):
        for K in ks:
            e = SpectrumEngine(n, max_frames=K)
            for kv in a.knob:
                e.debug_knob(kv.split('=')[0], int(kv.split('=')[1]))
            e.set_window(np.hanning(n).astype(np.float32))
            e.configure(db_mode="pow", power_scale=1.0, log_floor=1e-12, dc_alpha=-1.0, avg=("lin", K), cal_offset_db=-0.8087)

            def step(i):
                r = i % ring
                e.reset(nat.RESET_AVG)
                e.process_device(nat.IN_I8, di.value + r * iq.nbytes, n * K, n, K, do.value + r * n * 4)
            for i in range(a.warmup):
                step(i)
            e.synchronize()
            steps = max(20, a.steps * 64 // K)
            t0 = time.perf_counter()
            for i in range(steps):
                step(i)
            e.synchronize()
            us = (time.perf_counter() - t0) / steps * 1e6
            pts.append((K, us))
            print(f"K={K:4d}: {us:8.1f} us per capture, {us / K:.3f} us per segment", flush=True)
            e.close()
    ks_a = np.array([p[0] for p in pts], float)
    ts = np.array([p[1] for p in pts], float)
    b, c = np.polyfit(ks_a, ts, 1)
    print(f"fit: {c:.1f} us per capture + {b:.3f} us per segment  (K = 64 -> {c + 64 * b:.1f})")
    best = {K: min(t for k, t in pts if k == K) for K in ks}
    if a.capture in best:
        t1 = best[a.capture]
        for w in (2, 4, 8):
            kw = a.capture // w
            if kw in best:
                print(f"strong scaling of one {a.capture}-segment capture, compute only: {w} GPUs x {kw} segments = {best[kw]:.1f} us "
                      f"-> {t1 / best[kw]:.2f}x ({100 * t1 / best[kw] / w:.0f} % efficiency)")


if __name__ == "__main__":
    main()
