#!/usr/bin/env python3
"""PCIe-inclusive throughput of the C3 shape through the pinned host pipeline (tdsa_pipe_*):
host IQ -> H2D -> frame kernel -> (D2H dB rows) with the three legs overlapped.  Not the contract bench
(bench.py keeps inputs resident in HBM); this is the number DESIGN.md quotes for host-buffer callers."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from topdogspectrumanalyser_amd import SpectrumEngine  # noqa: E402
from topdogspectrumanalyser_amd.utils.synthetic import synth_iq_int8  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--slots", type=int, default=3)
    ap.add_argument("--streams", type=int, default=3)
    a = ap.parse_args()
    n, hop, ns = 16384, 8192, 20_000_000
    F = (ns - n) // hop + 1
    iq = synth_iq_int8(ns, n, seed=3)
    w = np.hanning(n).astype(np.float32)
    w /= np.sqrt(np.mean(w ** 2))
    for rows in (True, "u8", False):
        with SpectrumEngine(n, max_frames=F) as e:
            e.set_window(w)
            e.configure(db_mode="mag", log_floor=1e-12, dc_alpha=1.0, hold_max=True)
            e.set_overlap(a.streams)
            with e.pipe(ns, n_slots=a.slots, rows=rows) as q:
                for fill in (False, True):      # first pass: producer cost excluded (slots pre-filled)
                    for _ in range(a.slots):
                        q.acquire()[:] = iq
                        q.submit(ns, hop, F)
                    collect = q.collect_u8 if rows == "u8" else q.collect
                    while q.pending:
                        collect()
                    t0 = time.perf_counter()
                    for i in range(a.steps):
                        if q.pending == a.slots:
                            collect()
                        s = q.acquire()
                        if fill:
                            s[:] = iq            # producer memcpy into the pinned slot (one host core)
                        q.submit(ns, hop, F)
                    while q.pending:
                        collect()
                    dt = (time.perf_counter() - t0) / a.steps
                    gb = (2 * ns + ({True: 4, "u8": 1, False: 0}[rows] * F * n)) / 1e9
                    print(f"rows={ {True: 'float32', 'u8': 'uint8 levels', False: 'no (hold trace only)'}[rows]:22s}  producer memcpy={'yes' if fill else 'no '}: "
                          f"{dt*1e3:7.3f} ms/step  {F/dt/1e6:6.3f} Mframes/s  {gb/dt:6.1f} GB/s over PCIe")


if __name__ == "__main__":
    main()
