#!/usr/bin/env python3
"""End-to-end example: an int8 IQ capture -> spectra, peak-hold trace, per-frame peaks, top-5 peak lists,
density histogram and waterfall ring, with only scalars and two small images leaving the GPU.

    python examples/batch_capture.py [capture.i8]      (synthetic capture when no file is given)

What the reference does one 20 ms tick at a time (HackrfSamplesDataSource.get_power_levels ->
DataProcessor -> DutyCycleAnalyser / _find_top_peaks / DensityDisplay / Waterfall), for a whole capture:
the producer fills pinned slots, the copy of the next second overlaps the spectra of the current one, and
the dB rows never leave the device.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from topdogspectrumanalyser_amd import SpectrumEngine, analytics as an  # noqa: E402
from topdogspectrumanalyser_amd.utils.synthetic import synth_iq_int8  # noqa: E402

FS, FC, NFFT, HOP = 20e6, 2.45e9, 16384, 8192
SECOND = 20_000_000                       # samples per slot of the pinned ring


def seconds_of_iq(path):
    if path:
        raw = np.memmap(path, dtype=np.int8, mode="r")
        for k in range(raw.size // (2 * SECOND)):
            yield raw[2 * SECOND * k: 2 * SECOND * (k + 1)]
    else:
        for k in range(4):
            yield synth_iq_int8(SECOND, NFFT, seed=10 + k)


def main():
    frames = (SECOND - NFFT) // HOP + 1
    freq = np.fft.fftshift(np.fft.fftfreq(NFFT, 1 / FS)) + FC
    window = np.hanning(NFFT).astype(np.float32)
    window /= np.sqrt(np.mean(window ** 2))                 # hackrf_samples.py:314-316
    duty = an.DutyCycle()
    t0 = time.perf_counter()
    with SpectrumEngine(NFFT, max_frames=frames) as eng, an.DensityHistogram(NFFT, decay=0.999) as density, \
            an.WaterfallRing(1000, NFFT, -120.0) as waterfall:
        eng.set_window(window)
        eng.configure(db_mode="mag", log_floor=1e-12, dc_alpha=1.0, hold_max=True)

        def analyse(pipe):
            rows_dev, nf = pipe.collect_device()            # waits for the oldest second; rows stay on the GPU
            duty.update_from_rows(eng, rows_dev, nf, threshold_dbm=-30.0)
            bins, db = an.rows_top_peaks(eng, rows_dev, nf)
            _, _, band = an.rows_stats(eng, rows_dev, nf, freq_bins=freq, band=(FC - 1e6, FC + 1e6))
            density.update_rows(eng, rows_dev, nf)
            waterfall.push_rows(eng, rows_dev, nf)
            peaks = an.peaks_as_reference(freq, bins[-1], db[-1])
            print(f"  {nf} frames | duty {duty.duty_pct:5.1f} % | +-1 MHz band power {band.mean():7.2f} dB | "
                  + ", ".join(f"{f/1e6:.3f} MHz {p:.1f} dB" for f, p in peaks[:3]))
            return nf

        total = 0
        with eng.pipe(SECOND, n_slots=2, rows="device") as pipe:
            for iq in seconds_of_iq(sys.argv[1] if len(sys.argv) > 1 else None):
                if pipe.pending == 2:
                    total += analyse(pipe)
                pipe.acquire()[: iq.size] = iq              # straight into pinned memory
                pipe.submit(SECOND, HOP, frames)            # H2D + frame kernel, asynchronous
            while pipe.pending:
                total += analyse(pipe)
        peak_hold, _ = eng.hold()
        image, view = density.image(), waterfall.view()
    dt = time.perf_counter() - t0
    print(f"{total} frames in {dt:.2f} s wall (incl. synthesis); peak hold {peak_hold.max():.1f} dB at "
          f"{freq[int(np.argmax(peak_hold))]/1e6:.3f} MHz; density image {image.shape}, waterfall view {view.shape}")


if __name__ == "__main__":
    main()
