#!/usr/bin/env python3
"""Timing of the device-side analytics on one C3 second of spectra (2440 rows x 16384 bins, resident in HBM)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from topdogspectrumanalyser_amd import SpectrumEngine, _native as nat, analytics as an  # noqa: E402
from topdogspectrumanalyser_amd.utils.synthetic import synth_iq_int8  # noqa: E402


def timed(f, reps=5):
    f()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    return (time.perf_counter() - t0) / reps


def main():
    n, hop, ns = 16384, 8192, 20_000_000
    F = (ns - n) // hop + 1
    iq = synth_iq_int8(ns, n, seed=3)
    d_in, d_out = C.c_void_p(), C.c_void_p()
    nat.check(nat.lib.tdsa_dev_alloc(0, iq.nbytes, C.byref(d_in)))
    nat.check(nat.lib.tdsa_dev_alloc(0, F * n * 4, C.byref(d_out)))
    nat.check(nat.lib.tdsa_memcpy_h2d(0, d_in, iq.ctypes.data_as(C.c_void_p), iq.nbytes))
    w = np.hanning(n).astype(np.float32)
    w /= np.sqrt(np.mean(w ** 2))
    fb = np.fft.fftshift(np.fft.fftfreq(n, 1 / 20e6)) + 2.45e9
    with SpectrumEngine(n, max_frames=F) as e:
        e.set_window(w)
        e.configure(db_mode="mag", log_floor=1e-12, dc_alpha=1.0)
        e.process_device(nat.IN_I8, d_in.value, ns, hop, F, d_out.value)
        e.synchronize()
        mb = F * n * 4 / 1e6
        t = timed(lambda: an.rows_stats(e, d_out.value, F, freq_bins=fb, band=(2.449e9, 2.451e9)))
        print(f"rows_stats (peak, argmax, band power)   {t*1e3:8.3f} ms  ({mb/t/1e6:.2f} TB/s of rows)")
        for exc in (10.0, 6.0):
            t = timed(lambda: an.rows_top_peaks(e, d_out.value, F, min_excursion_db=exc), reps=2)
            print(f"rows_top_peaks n=5 excursion {exc:4.1f} dB      {t*1e3:8.3f} ms")
        with an.DensityHistogram(n, 0.96) as dh:
            t = timed(lambda: dh.update_rows(e, d_out.value, F), reps=3)
            print(f"density histogram, {F} rows             {t*1e3:8.3f} ms")
            t = timed(lambda: dh.image(), reps=3)
            print(f"density image read-back (32 MiB)        {t*1e3:8.3f} ms")
        with an.WaterfallRing(2000, n, -120.0) as wf:
            t = timed(lambda: wf.push_rows(e, d_out.value, F), reps=3)
            print(f"waterfall ring push, {F} rows (H=2000)   {t*1e3:8.3f} ms")
        rows = np.empty((F, n), dtype=np.float32)
        t = timed(lambda: nat.check(nat.lib.tdsa_memcpy_d2h(0, rows.ctypes.data_as(C.c_void_p), d_out, rows.nbytes)), reps=2)
        print(f"for scale: reading the {mb:.0f} MB of rows back  {t*1e3:8.3f} ms (pageable)")
        t0 = time.perf_counter()
        for r in rows[:20]:
            float(np.max(r)); int(np.argmax(r))
        print(f"for scale: numpy max+argmax per row        {(time.perf_counter()-t0)/20*F*1e3:8.1f} ms per second of IQ (one core)")


if __name__ == "__main__":
    main()
