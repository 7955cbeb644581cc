#!/usr/bin/env python3
"""Developer benchmark of the batched TraceAverager path (utils/signal_processing.py:35-61 of the reference, run over a
whole capture on the device): C3-shaped steps (2440 frames of 16384 points, hop N/2, int8 IQ resident in HBM) with
averaging on - frame kernel (linear rows + chunk aggregates) -> chain -> re-scan + dB rows.

python tools/avgbench.py [--avg exp 4] [--steps 1200] [--warmup 200] [--nfft 16384] [--hop 8192] [--frames 2440] [--hold 0]
(rocprofv3 --kernel-trace --stats around it gives the per-kernel split: profiles/r04_c3_avg_kernel_stats.csv)"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from topdogspectrumanalyser_amd import SpectrumEngine, _native as nat  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--avg", nargs=2, default=["exp", "4"])
    ap.add_argument("--steps", type=int, default=1200)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--nfft", type=int, default=16384)
    ap.add_argument("--hop", type=int, default=8192)
    ap.add_argument("--frames", type=int, default=2440)
    ap.add_argument("--hold", type=int, default=0)
    ap.add_argument("--state-only", action="store_true", help="no dB rows wanted (out = NULL): only the averager's state")
    ap.add_argument("--ring", type=int, default=4, help="distinct input / output buffers cycled through")
    ap.add_argument("--f64-chunks", action="store_true", help="tdsa_debug_knob avg_f64_chunks: fixed 64-frame chunks, float64 aggregates")
    a = ap.parse_args()
    n, hop, F = a.nfft, a.hop, a.frames
    ns = hop * (F - 1) + n
    iq = np.random.default_rng(0).integers(-100, 100, size=2 * ns, dtype=np.int8)
    di, do = C.c_void_p(), C.c_void_p()
    nat.check(nat.lib.tdsa_dev_alloc(0, iq.nbytes * a.ring, C.byref(di)))
    nat.check(nat.lib.tdsa_dev_alloc(0, F * n * 4 * a.ring, C.byref(do)))
    for r in range(a.ring):
        nat.check(nat.lib.tdsa_memcpy_h2d(0, C.c_void_p(di.value + r * iq.nbytes), iq.ctypes.data_as(C.c_void_p), iq.nbytes))
    e = SpectrumEngine(n, max_frames=F)
    e.set_window(np.hanning(n).astype(np.float32))
    avg = (a.avg[0], int(a.avg[1]))
    e.configure(db_mode="pow", power_scale=1.0, log_floor=1e-10, dc_alpha=1.0, avg=avg, hold_max=bool(a.hold & 1),
                hold_min=bool(a.hold & 2))
    if a.f64_chunks:
        e.debug_knob("avg_f64_chunks", 1)

    def step(i):
        r = i % a.ring
        e.process_device(nat.IN_I8, di.value + r * iq.nbytes, ns, hop, F, None if a.state_only else do.value + r * F * n * 4)
    for i in range(a.warmup):
        step(i)
    e.synchronize()
    e.timer_begin()
    for i in range(a.steps):
        step(i)
    ms = e.timer_end()
    us = ms / a.steps * 1e3
    algo = F * (2 * hop + 4 * n)
    print(f"avg={avg} N={n} hop={hop} F={F} hold={a.hold} state_only={int(a.state_only)} f64_chunks={int(a.f64_chunks)}  step {us:.1f} us  "
          f"{algo / us / 1e6:.3f} TB/s algorithmic ({algo / us / 1e6 / 8 * 100:.1f} % of 8 TB/s)")


if __name__ == "__main__":
    main()
