#!/usr/bin/env python3
"""Golden vectors for the trace analytics from the *imported reference*.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_analytics.py

Writes tests/golden/analytics.npz: seeded dB traces in, what DataProcessor._find_top_peaks,
DutyCycleAnalyser.update_from_power and MarkerManager._band_power returned out.  DATA only.
"""
import os
import sys
import types
from unittest.mock import MagicMock

sys.dont_write_bytecode = True
for _m in ("hackrf", "rtlsdr", "sounddevice"):
    sys.modules[_m] = MagicMock()
REF = os.environ.get("TDSA_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))

import numpy as np  # noqa: E402

from core.display_data_processor import DataProcessor  # noqa: E402
from core.duty_cycle import DutyCycleAnalyser  # noqa: E402
from core.marker_manager import MarkerManager  # noqa: E402


def traces(rng, n, kind):
    """dB traces with the statistics of a periodogram: exponential noise power + a few tones."""
    k = np.arange(n)
    p = rng.exponential(1.0, size=n) * 1e-9
    if kind >= 1:
        for c, a in ((n // 8, 1e-3), (n // 3 + 7, 3e-5), (2 * n // 3, 4e-6), (n - 40, 8e-7)):
            p += a * np.sinc((k - c - 0.3 * kind) / 1.5) ** 2
    if kind == 2:                                   # two close tones that must merge / split on excursion
        p += 2e-4 * np.sinc((k - n // 2) / 1.2) ** 2 + 1.5e-4 * np.sinc((k - n // 2 - 14) / 1.2) ** 2
    if kind == 3:                                   # smooth trace: few local maxima
        p = 1e-8 * (2 + np.cos(2 * np.pi * k / n * 5)) + 1e-12 * k
    return (10 * np.log10(p + 1e-12)).astype(np.float32)


def main():
    rng = np.random.default_rng(20240611)
    out = {}
    cases = []
    for n in (64, 1024, 4096, 16384):
        fb = np.linspace(2.44e9, 2.46e9, n)
        for kind in range(4):
            tr = traces(rng, n, kind)
            for exc in (6.0, 10.0):
                min_sep = max(10, n // 50)
                pk = DataProcessor._find_top_peaks(fb, tr, n=5, min_sep_bins=min_sep, min_excursion_db=exc)
                idx = [int(np.argmin(np.abs(fb - f))) for f, _ in pk]
                key = f"peaks_{n}_{kind}_{int(exc)}"
                out[key + "_trace"] = tr
                out[key + "_bins"] = np.array(idx, dtype=np.int32)
                out[key + "_pwr"] = np.array([p for _, p in pk], dtype=np.float64)
                cases.append(key)
    out["peak_cases"] = np.array(cases)

    # duty cycle: 260 frames of peaks, threshold -60 then -45
    dca = DutyCycleAnalyser()
    frames = np.stack([traces(rng, 1024, 1) + (0.0 if (i // 13) % 3 else -40.0) for i in range(260)])
    duty, on, off = [], [], []
    for i, fr in enumerate(frames):
        dca.update_from_power(fr, threshold_dbm=-60.0 if i < 130 else -45.0)
        duty.append(dca.duty_pct)
        on.append(np.nan if dca.on_power_dbm is None else dca.on_power_dbm)
        off.append(np.nan if dca.off_power_dbm is None else dca.off_power_dbm)
    out["duty_frames"] = frames.astype(np.float32)
    out["duty_pct"] = np.array(duty)
    out["duty_on"] = np.array(on)
    out["duty_off"] = np.array(off)

    # band power through MarkerManager._band_power with a stub self
    n = 4096
    fb = np.linspace(99.3e6, 101.3e6, n)
    tr = traces(rng, n, 1)
    stub = types.SimpleNamespace(_data=lambda: (fb, tr))
    bands = [(99.5e6, 99.9e6), (101.0e6, 100.2e6), (99.3e6, 101.3e6), (100.0e6, 100.0e6 + 1.0), (50e6, 60e6)]
    vals = []
    for a, b in bands:
        v = MarkerManager._band_power(stub, a, b)
        vals.append(np.nan if v is None else v)
    out["band_bins"] = fb
    out["band_trace"] = tr
    out["band_edges"] = np.array(bands)
    out["band_db"] = np.array(vals)

    np.savez_compressed(os.path.join(HERE, "analytics.npz"), **out)
    print("wrote analytics.npz:", len(cases), "peak cases")


if __name__ == "__main__":
    main()
