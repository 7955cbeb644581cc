"""Software "devices" for the GPU sample sources when no SDR is attached: they speak the tiny part of
the pyhackrf / pyrtlsdr object interface the sources use (read_samples, setters, close) and replay a
recorded or synthetic IQ array, unpacking int8 with the (I + jQ)/128 convention."""
from typing import Optional

import numpy as np


def _to_c64(iq: np.ndarray) -> np.ndarray:
    iq = np.asarray(iq)
    if iq.dtype == np.int8:
        f = iq.astype(np.float32) * np.float32(1.0 / 128.0)
        return (f[0::2] + 1j * f[1::2]).astype(np.complex64)
    return iq.astype(np.complex64)


class ReplayHackRF:
    """read_samples(n) returns consecutive n-sample blocks (wrapping around)."""

    def __init__(self, iq: np.ndarray):
        self._x = _to_c64(iq)
        self._pos = 0
        self.closed = False

    def read_samples(self, n: int) -> np.ndarray:
        if self.closed:
            raise IOError("device closed")
        idx = (self._pos + np.arange(n)) % len(self._x)
        self._pos = (self._pos + n) % len(self._x)
        return self._x[idx]

    def set_sample_rate(self, v): self.sample_rate = v
    def set_freq(self, v): self.freq = v
    def set_lna_gain(self, v): self.lna = v
    def set_vga_gain(self, v): self.vga = v
    def enable_amp(self): self.amp = True
    def disable_amp(self): self.amp = False
    def close(self): self.closed = True


class ReplayRtlSdr:
    def __init__(self, iq: np.ndarray, sample_rate: Optional[float] = None, center_freq: float = 0.0):
        self._x = _to_c64(iq)
        self._pos = 0
        self.sample_rate = sample_rate
        self.center_freq = center_freq
        self.gain = "auto"

    def read_samples(self, n: int) -> np.ndarray:
        idx = (self._pos + np.arange(n)) % len(self._x)
        self._pos = (self._pos + n) % len(self._x)
        return self._x[idx]

    def get_sample_rate(self): return float(self.sample_rate)
    def get_center_freq(self): return float(self.center_freq)
    def close(self): pass
