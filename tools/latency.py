#!/usr/bin/env python3
"""Developer tool: what one displayed frame costs through the host entry point (what the live sources do on every
GUI tick): total per call, and the same frame with the samples already on the device."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from topdogspectrumanalyser_amd import SpectrumEngine, _native as nat  # noqa: E402


def main():
    for n in (1000, 1024, 2048, 4096, 8192, 16384, 1 << 15, 1 << 17, 1 << 20):
        iq = np.random.default_rng(0).integers(-100, 100, size=2 * n, dtype=np.int8)
        x = (iq[0::2] + 1j * iq[1::2]).astype(np.complex64) / 128
        with SpectrumEngine(n, max_frames=1) as e:
            e.set_window(np.hanning(n).astype(np.float32))
            e.configure(db_mode="mag", log_floor=1e-12, dc_alpha=1.0)
            for name, data in (("int8", iq), ("complex64", x)):
                for _ in range(200):
                    e.process(data, hop=n, n_frames=1)
                t0 = time.perf_counter()
                for _ in range(2000):
                    e.process(data, hop=n, n_frames=1)
                host = (time.perf_counter() - t0) / 2000 * 1e6
                print(f"N={n:6d} {name:9s}: {host:6.1f} us per host call", end="")
                if name == "int8":
                    d_in, d_out = C.c_void_p(), C.c_void_p()
                    nat.check(nat.lib.tdsa_dev_alloc(0, iq.nbytes, C.byref(d_in)))
                    nat.check(nat.lib.tdsa_dev_alloc(0, n * 4, C.byref(d_out)))
                    nat.check(nat.lib.tdsa_memcpy_h2d(0, d_in, iq.ctypes.data_as(C.c_void_p), iq.nbytes))
                    for _ in range(200):
                        e.process_device(nat.IN_I8, d_in.value, n, n, 1, d_out.value)
                    e.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(2000):
                        e.process_device(nat.IN_I8, d_in.value, n, n, 1, d_out.value)
                        e.synchronize()
                    dev = (time.perf_counter() - t0) / 2000 * 1e6
                    print(f"   device-resident + synchronize: {dev:6.1f} us", end="")
                print()

    # the display processor's share of a tick: calibration offset, tare, both holds in one launch (TraceState.update)
    from topdogspectrumanalyser_amd.engine import TraceState
    for n in (1024, 4096, 16384):
        row = np.random.default_rng(1).normal(-80, 5, n).astype(np.float32)
        ts = TraceState(n)
        for _ in range(200):
            ts.update(row, cal_offset_db=-0.8, hold_max=True, hold_min=True)
        t0 = time.perf_counter()
        for _ in range(2000):
            ts.update(row, cal_offset_db=-0.8, hold_max=True, hold_min=True)
        print(f"N={n:6d} trace update (cal + both holds, three rows back): {(time.perf_counter() - t0) / 2000 * 1e6:6.1f} us")
        ts.close()
    # TraceAverager.process on a host row (float64 in, float64 state out)
    from topdogspectrumanalyser_amd import TraceAverager
    for n in (1024, 16384):
        ta = TraceAverager()
        ta.set_mode("exp", 8)
        row = np.random.default_rng(3).uniform(1e-9, 1e-6, n)
        for _ in range(200):
            ta.process(row)
        t0 = time.perf_counter()
        for _ in range(2000):
            ta.process(row)
        print(f"N={n:6d} TraceAverager.process (host row):                  {(time.perf_counter() - t0) / 2000 * 1e6:6.1f} us")

    # one tick of the microphone source: stereo float32 block -> one-sided dB trace(s)
    for n in (1024, 4096):
        st = np.random.default_rng(2).normal(0, 0.1, (n, 2)).astype(np.float32)
        with SpectrumEngine(n, max_frames=1) as e:
            e.set_window(np.hanning(n).astype(np.float32))
            e.configure(db_mode="pow", log_floor=1e-10, dc_alpha=1.0)
            for chan in ("mono", "stereo"):
                for _ in range(200):
                    e.process_real2(st, chan)
                t0 = time.perf_counter()
                for _ in range(2000):
                    e.process_real2(st, chan)
                print(f"N={n:6d} audio tick ({chan:6s}): {(time.perf_counter() - t0) / 2000 * 1e6:6.1f} us")

    # the displays' share of a tick: one host row into the density histogram / the waterfall ring, the marker search and the
    # peak list of one row on the device
    from topdogspectrumanalyser_amd import analytics as an
    for n in (1024, 4096, 16384):
        row = np.random.default_rng(4).normal(-80, 5, n).astype(np.float32)
        row[n // 3] = -20.0
        with SpectrumEngine(n, max_frames=1) as e, an.DensityHistogram(n, 0.96) as dh, an.WaterfallRing(600, n, -120.0) as wf:
            d_row = C.c_void_p()
            nat.check(nat.lib.tdsa_dev_alloc(0, n * 4, C.byref(d_row)))
            nat.check(nat.lib.tdsa_memcpy_h2d(0, d_row, row.ctypes.data_as(C.c_void_p), n * 4))
            rows2 = [row, row + np.float32(0.5)]
            calls = (("density update (host row)", lambda i: dh.update(rows2[i & 1])),
                     ("waterfall push (host row)", lambda i: wf.push(rows2[i & 1])),
                     ("waterfall view_u8 (600 lines back)", lambda i: wf.view_u8(-110.0, -20.0)),
                     ("top-5 peaks of one device row", lambda i: an.rows_top_peaks(e, d_row.value, 1)),
                     ("marker search of one device row", lambda i: an.rows_marker_peaks(e, d_row.value, 1, peak_threshold=-60.0,
                                                                                        peak_excursion=6.0)),
                     ("peak / argmax / band of one device row", lambda i: an.rows_stats(e, d_row.value, 1)))
            for name, fn in calls:
                reps = 200 if "view" in name else 2000
                for i in range(50):
                    fn(i)
                t0 = time.perf_counter()
                for i in range(reps):
                    fn(i)
                print(f"N={n:6d} {name:40s}: {(time.perf_counter() - t0) / reps * 1e6:7.1f} us")
            nat.lib.tdsa_dev_free(0, d_row)

if __name__ == "__main__":
    main()
