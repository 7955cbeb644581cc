"""The two source interfaces the rest of an analyser program is written against.

This is the drop-in boundary of the path: the class names, method names, argument
meaning and defaults are those of the reference's datasources/base.py, so whatever drives a reference
source - its SourceManager, DisplayManager, DataProcessor, or a test with stub objects - drives these
classes unchanged.  Nothing here computes; a concrete sample source does its arithmetic on the GPU through
GpuSpectrumMixin.

  SweepDataSource    something that delivers finished dB sweeps (the reference wraps external command
                     line tools; none is built here, the interface exists for code that type-checks
                     against it)
  SampleDataSource   something that delivers IQ / audio samples and turns N of them into a spectrum:
                     `get_power_levels() -> (power_db[N], frequency_bins[N])`
"""
import threading
import time
from abc import ABC, abstractmethod
from typing import Optional, Tuple

import numpy as np

from ..utils.signal_processing import TraceAverager


class SweepDataSource(ABC):
    @abstractmethod
    def start(self, frequency=None):
        """Begin sweeping (`frequency` carries start / stop / centre / span when given)."""

    @abstractmethod
    def stop(self):
        """Stop sweeping and let go of the device or subprocess."""

    @abstractmethod
    def get_data(self):
        """Latest complete sweep as dB values, or None."""


class SampleDataSource(ABC):
    def __init__(self, sample_rate: Optional[int] = None, centre_freq: Optional[int] = None):
        self.sample_rate = sample_rate
        self.centre_freq = centre_freq
        self.last_data_time: float = 0.0                  # time.monotonic() of the newest stored frame
        self._averager = TraceAverager()                  # per-source trace averaging (off by default)
        self._last_raw_samples: Optional[np.ndarray] = None
        self._raw_lock = threading.Lock()

    # ---- what every concrete source has to provide ---------------------------------------------------
    @abstractmethod
    def start(self, frequency=None):
        """Open the device and start delivering samples; RuntimeError when that is impossible."""

    @abstractmethod
    def stop(self):
        """Stop and release the device; harmless when not running."""

    @abstractmethod
    def get_power_levels(self) -> Tuple[np.ndarray, np.ndarray]:
        """One spectrum: (dB values, frequency axis).  Never raises: a source without data answers with
        zeros / its floor value / its last good trace over a valid axis."""

    @property
    @abstractmethod
    def sample_count(self) -> int:
        """FFT length N."""

    @sample_count.setter
    @abstractmethod
    def sample_count(self, value: int):
        ...

    @abstractmethod
    def update_frequency(self, sample_rate: float, centre_freq: float):
        """Change span (= sample rate) and centre frequency together."""

    @abstractmethod
    def update_centre_frequency(self, centre_freq: float):
        """Retune without touching the sample rate."""

    # ---- shared behaviour ---------------------------------------------------------------------------------
    def set_averaging(self, mode: str, n: int) -> None:
        """'off' | 'exp' | 'lin' over n frames; any change restarts the average."""
        self._averager.set_mode(mode, n)

    def reset_averaging(self) -> None:
        self._averager.reset()

    def set_psd_mode(self, enabled: bool):
        """Sources that can normalise to power per hertz override this; the default ignores it."""

    def read_samples_only(self) -> Optional[np.ndarray]:
        """Raw samples without a spectrum (constellation / zero-span views); None when unsupported."""
        return None

    def get_raw_samples(self) -> Optional[np.ndarray]:
        """The samples behind the most recent spectrum (the stored object itself, not a copy)."""
        with self._raw_lock:
            return self._last_raw_samples

    def _store_raw(self, samples: np.ndarray) -> None:
        with self._raw_lock:
            self._last_raw_samples = samples
        self.last_data_time = time.monotonic()
