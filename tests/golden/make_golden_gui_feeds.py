#!/usr/bin/env python3
"""Golden vectors for the host-side GUI feeds of DataProcessor from the *imported reference*
(core/display_data_processor.py:230-311 zero-span trigger and constellation EVM read-out, :407-430 peak
list read-out, :185-229 sweep path incl. the lost-range recovery).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_gui_feeds.py

Writes tests/golden/gui_feeds.npz.  DATA only: seeded inputs, what the widgets / labels received.
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
from utils.constants import DisplayMode  # noqa: E402


class Label:
    text = None

    def setText(self, s):
        self.text = s


class ZeroSpanWidget:
    def update_zero_span_data(self, t, y):
        self.t, self.y = np.array(t), np.array(y)


class Source:
    sample_rate = 8000.0

    def __init__(self):
        self.block = None

    def read_samples_only(self):
        return self.block


def main():
    rng = np.random.default_rng(20240920)
    out = {}

    # ---- zero span: free run / rise / fall, buffers shorter and longer than one window -------------
    src = Source()
    mw = types.SimpleNamespace(current_source=src, zero_span_widget=ZeroSpanWidget())
    dm = types.SimpleNamespace(zero_span_buffer=None, zero_span_time_window=0.01, zero_span_trigger_mode="free_run",
                               zero_span_trigger_level=0.2)
    dp = DataProcessor.__new__(DataProcessor)
    dp.mw, dp.dm = mw, dm
    blocks, modes, shown = [], [], []
    t = 0
    for step in range(12):
        n = int(rng.integers(30, 400))
        k = np.arange(t, t + n)
        t += n
        blk = (np.sin(2 * np.pi * k / 57.0) * (0.6 + 0.3 * np.sin(k / 301.0))
               + 0.05 * rng.standard_normal(n)).astype(np.complex64)
        if step % 4 == 3:
            blk = np.stack([blk.real, 0.5 * blk.real], axis=1).astype(np.float32)      # stereo block
        dm.zero_span_trigger_mode = ("free_run", "rise", "fall")[step % 3]
        dm.zero_span_trigger_level = float((-0.3, 0.2, 0.45)[step % 3])
        src.block = blk
        dp._process_zero_span_data()
        blocks.append(blk)
        modes.append(dm.zero_span_trigger_mode)
        shown.append(mw.zero_span_widget.y.copy())
    for i, b in enumerate(blocks):
        out[f"zs_block_{i}"] = b
        out[f"zs_shown_{i}"] = shown[i]
    out["zs_modes"] = np.array(modes)
    out["zs_levels"] = np.array([(-0.3, 0.2, 0.45)[s % 3] for s in range(12)])
    out["zs_rate"] = np.float64(src.sample_rate)
    out["zs_window"] = np.float64(dm.zero_span_time_window)

    # ---- peak list read-out ------------------------------------------------------------------------
    n = 2048
    fb = np.linspace(88e6, 108e6, n)
    kk = np.arange(n)
    p = rng.exponential(1.0, size=n) * 1e-9
    for c, a in ((300, 1e-3), (900, 2e-4), (1500, 5e-5), (1900, 3e-6)):
        p += a * np.sinc((kk - c) / 1.5) ** 2
    tr = (10 * np.log10(p + 1e-12)).astype(np.float32)

    class TwoD:
        def set_peak_list(self, peaks):
            self.peaks = peaks

    mw2 = types.SimpleNamespace(two_d_widget=TwoD(), marker_readout_label=Label(), peak_excursion=8.0)
    dp2 = DataProcessor.__new__(DataProcessor)
    dp2.mw, dp2.dm = mw2, types.SimpleNamespace(peak_list_enabled=True)
    dp2._update_peak_list(fb, tr)
    out["pl_bins"], out["pl_trace"] = fb, tr
    out["pl_text"] = np.array(mw2.marker_readout_label.text)
    out["pl_peaks"] = np.array(mw2.two_d_widget.peaks, dtype=np.float64)

    # ---- constellation EVM read-out -------------------------------------------------------------------
    texts = []
    for evm in (0.0731, 0.5, None, 0.0):
        class View:
            last_evm_rms = evm

            def update_iq_data(self, s):
                self.got = s

        lab = Label()
        mw3 = types.SimpleNamespace(current_source=types.SimpleNamespace(read_samples_only=lambda: np.ones(8, np.complex64)),
                                    current_stacked_index=DisplayMode.CONSTELLATION_2D, constellation_2d_widget=View(),
                                    marker_readout_label=lab)
        dp3 = DataProcessor.__new__(DataProcessor)
        dp3.mw, dp3.dm = mw3, types.SimpleNamespace(constellation_modulation="qpsk")
        dp3._process_constellation_data()
        texts.append(lab.text)
    out["evm_values"] = np.array([0.0731, 0.5, np.nan, 0.0])
    out["evm_texts"] = np.array(texts)

    np.savez_compressed(os.path.join(HERE, "gui_feeds.npz"), **out)
    print("wrote gui_feeds.npz:", len(blocks), "zero-span steps;", repr(mw2.marker_readout_label.text[:60]), texts)


if __name__ == "__main__":
    main()
