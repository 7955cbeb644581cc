#!/usr/bin/env python3
"""Golden vectors for the marker peak search from the *imported reference*.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_markers.py

Writes tests/golden/markers.npz: seeded dB traces in (`trace_<n>_<kind>`, shared by the three parameter sets of a case); out, what the reference's MarkerManager did with them
(core/marker_manager.py:74-127): where snap_to_peak put the F1 marker, where a run of snap_to_next_peak calls walked
it, and the (peaks, prominences) its own `_scipy_find_peaks(levels, height=, prominence=, distance=3)` returned.
The MarkerManager is the real class on a stub main window (plain namespace; the display sync and the status
read-out, which only touch widgets, are replaced by no-ops).  DATA only.

Traces never hold two EQUAL local maxima closer than 3 bins: scipy orders the distance rule by np.argsort, numpy's
default sort is not stable (AVX-512 / AVX2 sorting networks), so what the reference returns for such a pair depends
on the CPU it runs on.  Equal peaks further apart, flat tops and quantised traces are all in.
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

import core.marker_manager as mmod  # noqa: E402
from core.marker_manager import MarkerManager  # noqa: E402

DISTANCE = 3            # marker_manager.py:91, 118
N_NEXT = 6              # snap_to_next_peak calls per case


def local_maxima(x):
    """positions scipy's _local_maxima_1d reports (used only to keep ambiguous ties out of the fixture)."""
    out, i, last = [], 1, len(x) - 1
    while i < last:
        if x[i - 1] < x[i]:
            a = i + 1
            while a < last and x[a] == x[i]:
                a += 1
            if x[a] < x[i]:
                out.append((i + a - 1) // 2)
                i = a
        i += 1
    return out


def break_close_ties(x, step):
    """raise the right one of two equal maxima closer than DISTANCE by `step` until none is left"""
    x = x.copy()
    for _ in range(10000):
        lm = local_maxima(x)
        hit = [(a, b) for a, b in zip(lm, lm[1:]) if b - a < DISTANCE and x[a] == x[b]]
        if not hit:
            return x
        for _, b in hit:
            x[b] += step
    raise RuntimeError("ties did not clear")


def trace(rng, n, kind):
    k = np.arange(n)
    p = rng.exponential(1.0, size=n) * 1e-9                      # periodogram noise
    if kind in (1, 2, 5):
        for c, a in ((n // 8, 1e-3), (n // 3 + 7, 3e-5), (2 * n // 3, 4e-6), (n - 9, 8e-7), (3, 5e-5)):
            p += a * np.sinc((k - c - 0.3 * kind) / 1.5) ** 2
    if kind == 2:                                                 # twin tones: prominence decides between them
        p += 2e-4 * np.sinc((k - n // 2) / 1.2) ** 2 + 1.5e-4 * np.sinc((k - n // 2 - 6) / 1.2) ** 2
    x = 10 * np.log10(p + 1e-12)
    if kind == 3:                                                 # smooth ripple: few maxima, small prominences
        x = -70 + 4 * np.cos(2 * np.pi * k / n * 7) + 1e-3 * k
    if kind == 4:                                                 # quantised to 0.5 dB: flat tops, equal peaks
        x = np.round(2 * (x + rng.normal(0, 1, n))) / 2
    if kind == 5:                                                 # clipped: wide plateaus on the strong tones
        x = np.minimum(x, -55.0)
    if kind == 6:                                                 # two equal strongest peaks far apart
        x = rng.normal(-90, 1.5, n)
        x[n // 4] = x[3 * n // 4] = -30.0
        x[n // 2 - 1: n // 2 + 2] = -30.0                         # and a flat top of the same height
    if kind == 7:                                                 # nothing qualifies: monotone ramp (argmax fallback)
        x = -100 + 20.0 * k / n
    x = x.astype(np.float32)
    return break_close_ties(x, np.float32(0.5 if kind == 4 else 1e-3))


def main():
    rng = np.random.default_rng(20260929)
    out, cases = {}, []
    for n in (64, 1000, 4096, 16384):
        fb = np.linspace(2.44e9, 2.46e9, n)
        for kind in range(8):
            tr = trace(rng, n, kind)
            for thr, exc in ((None, None), (-80.0, 10.0), (-62.5, 3.0)):
                mw = types.SimpleNamespace(frequency_bins=fb, live_power_levels=tr)
                if thr is not None:
                    mw.peak_threshold, mw.peak_excursion = thr, exc
                mm = MarkerManager(mw)
                mm._sync_display = lambda name: None
                mm._refresh_status = lambda: None
                mm.active_marker = "F1"
                f1 = mm.markers["F1"]
                f1.enabled, f1.position = True, float(fb[n // 3])
                mm.snap_to_peak()
                snap = int(np.searchsorted(fb, f1.position))
                assert fb[snap] == f1.position
                start = int(rng.integers(0, n))
                f1.position = float(fb[start])
                walk = []
                for _ in range(N_NEXT):
                    mm.snap_to_next_peak()
                    walk.append(int(np.searchsorted(fb, f1.position)))
                t = -200.0 if thr is None else thr
                e = 6.0 if exc is None else exc
                pk, props = mmod._scipy_find_peaks(tr, height=t, prominence=e, distance=DISTANCE)
                key = f"m_{n}_{kind}_{'d' if thr is None else int(-thr)}"
                out[f"trace_{n}_{kind}"] = tr
                out[key + "_params"] = np.array([t, e, DISTANCE, start], dtype=np.float64)
                out[key + "_defaults"] = np.array(thr is None)
                out[key + "_snap"] = np.array(snap, dtype=np.int32)
                out[key + "_walk"] = np.array(walk, dtype=np.int32)
                out[key + "_peaks"] = pk.astype(np.int32)
                out[key + "_prom"] = props["prominences"].astype(np.float64)
                cases.append(key)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "markers.npz"), **out)
    print("wrote markers.npz:", len(cases), "cases")


if __name__ == "__main__":
    main()
