#!/usr/bin/env python3
"""Golden vectors for the long-frame (C5) shape from the *imported reference itself*.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_c5.py

RtlSamplesDataSource.get_power_levels (datasources/rtl_samples.py:148-197) at fft_size = 2^20 with
TraceAverager("lin", 8) (utils/signal_processing.py:35-61) over K = 8 consecutive segments of the seeded
SURVEY.md 8(d) signal - the Welch average BASELINE.json config 5 asks for, minus the scalar calibration offset
that DataProcessor adds afterwards (core/display_data_processor.py:317-327).  A 2^20-point float64 trace is 8 MiB,
so the fixture keeps a comb of it: every 257th bin and the 64 strongest bins, after segment 1 (a plain frame)
and after segment 8 (the average).  DATA only; the input is regenerated from the stored seed.
"""
import os
import sys
from unittest.mock import MagicMock

sys.dont_write_bytecode = True
for _m in ("hackrf", "rtlsdr", "sounddevice"):
    sys.modules[_m] = MagicMock()
REF = os.environ.get("TDSA_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

import numpy as np  # noqa: E402

from datasources.rtl_samples import RtlSamplesDataSource  # noqa: E402
from oracle.spectrum_oracle import synth_iq_int8, unpack_iq_int8  # noqa: E402  (input generator only)

NFFT, K, SEED, FS, FC = 1 << 20, 8, 5, 2_000_000, 100_300_000


class _FakeSdr:
    def __init__(self, x):
        self._x, self._i = x, 0

    def read_samples(self, n):
        out = np.array(self._x[self._i: self._i + n], copy=True)
        self._i += n
        assert len(out) == n
        return out

    def get_sample_rate(self):
        return float(FS)

    def get_center_freq(self):
        return float(FC)


def main():
    x = unpack_iq_int8(synth_iq_int8(NFFT * K, NFFT, seed=SEED))        # complex64, (I + jQ)/128
    src = RtlSamplesDataSource(sample_rate=FS, centre_freq=FC)
    src.set_fft_size(NFFT)
    src.set_averaging("lin", K)
    src.sdr = _FakeSdr(x)
    src.running = True
    out = {}
    for k in range(K):
        p, fb = src.get_power_levels()
        if k in (0, K - 1):
            tag = "first" if k == 0 else "mean"
            p = np.asarray(p, dtype=np.float64)
            comb = np.arange(0, NFFT, 257)
            top = np.sort(np.argsort(p)[-64:])
            out[f"{tag}_comb_bins"], out[f"{tag}_comb_db"] = comb, p[comb]
            out[f"{tag}_top_bins"], out[f"{tag}_top_db"] = top, p[top]
            out[f"{tag}_sum_db"] = np.float64(p.sum())
    out.update(nfft=np.int64(NFFT), k=np.int64(K), seed=np.int64(SEED), sample_rate=np.float64(FS),
               centre_freq=np.float64(FC), freq_first=np.float64(fb[0]), freq_last=np.float64(fb[-1]))
    np.savez_compressed(os.path.join(HERE, "c5_million.npz"), **out)
    print("wrote c5_million.npz: comb", len(out["mean_comb_bins"]), "bins, strongest at", int(out["mean_top_bins"][np.argmax(out["mean_top_db"])]))


if __name__ == "__main__":
    main()
