"""RtlSamplesDataSource - RTL-SDR IQ source whose DSP runs on the MI355X.

Surface and semantics follow the reference's datasources/rtl_samples.py (class :17-255): blocking
sdr.read_samples(N) per frame (:167), raw (un-normalised) Hann / Hamming / rectangular window
(:199-206), NO DC removal, |X|^2 -> averager -> 10*log10(P + 1e-10), or PSD P/(fs*N) with floor 1e-12
(:175-184), frequency axis rebuilt per call from the hardware-reported fs / fc (:159-160,188), zeros +
linspace axis when not running or on error (:149-155,191-197), pause/resume, gain, flush-after-retune.
Kept quirk: set_fft_size() always rebuilds a Hann window whatever set_window_type() chose (:208-215).
"""
import logging
from typing import Callable, Optional

import numpy as np

from .base import SampleDataSource
from ._gpu import GpuSpectrumMixin

try:
    from rtlsdr import RtlSdr  # type: ignore
    _RTL_AVAILABLE = True
except (ImportError, OSError):
    RtlSdr = None
    _RTL_AVAILABLE = False

logger = logging.getLogger(__name__)

_WINDOWS = {"hanning": np.hanning, "hamming": np.hamming, "rectangle": np.ones}


class RtlSamplesDataSource(GpuSpectrumMixin, SampleDataSource):
    def __init__(self, sample_rate: int, centre_freq: int, device_factory: Optional[Callable] = None,
                 gpu_device: int = 0):
        super().__init__(sample_rate, centre_freq)
        self.fft_size = 1024
        self.sdr = None
        self.window = np.hanning(self.fft_size)
        self.running = False
        self.last_sample_rate = sample_rate
        self.use_psd = False
        self._gain = "auto"
        self._flush_reads_remaining = 0
        self._device_factory = device_factory
        self._gpu_device = gpu_device
        self._engine_dirty = True
        self._averager._on_change = lambda mode, n: setattr(self, "_engine_dirty", True)
        self._averager._on_reset = self._gpu_reset_averager

    # ------------------------------------------------------------------ lifecycle
    def start(self, frequency=None):
        if self._device_factory is None and not _RTL_AVAILABLE:
            raise RuntimeError("RTL-SDR library (librtlsdr) not available on this system")
        if frequency:
            self.centre_freq = int(frequency.centre)
            self.sample_rate = int(frequency.span)
        if self.running:
            return
        try:
            self.sdr = self._device_factory() if self._device_factory is not None else RtlSdr()
            self.sdr.sample_rate = self.sample_rate
            self.sdr.center_freq = self.centre_freq
            self.sdr.gain = self._gain
            actual = self.sdr.get_sample_rate()         # hardware may round the requested rate
            self.sample_rate = actual
            self.last_sample_rate = actual
            self.running = True
        except Exception as e:
            self.running = False
            raise RuntimeError(f"RTL-SDR initialisation failed: {e}")

    def pause(self):
        self.running = False

    def resume(self):
        if self.sdr is not None:
            self.running = True

    def stop(self):
        if self.sdr is not None:
            try:
                self.sdr.close()
            except Exception as e:
                logger.error("Error closing RTL-SDR: %s", e)
            self.sdr = None
        self.running = False

    # ------------------------------------------------------------------ retuning
    def update_centre_frequency(self, centre_freq: float):
        if not self.running:
            return
        centre_freq = int(centre_freq)
        if centre_freq == self.centre_freq:
            return
        self.centre_freq = centre_freq
        try:
            self.sdr.center_freq = centre_freq
            # discard the reads taken while the PLL settles (~6 ms of samples, at least 3 frames)
            self._flush_reads_remaining = max(3, int(0.006 * self.sample_rate / self.fft_size))
        except Exception as e:
            raise RuntimeError(f"Error updating centre frequency: {e}")

    def update_sample_rate(self, sample_rate: float):
        sample_rate = int(sample_rate)
        if sample_rate == self.last_sample_rate:
            return
        if self.running and self.sdr is not None:
            try:
                self.sdr.sample_rate = sample_rate
                actual = self.sdr.get_sample_rate()
                self.sample_rate = actual
                self.last_sample_rate = actual
                self.sdr.center_freq = self.centre_freq  # the tuner can shift when the rate changes
                self.centre_freq = self.sdr.get_center_freq()
                self._engine_dirty = True
            except Exception as e:
                raise RuntimeError(f"Error updating sample rate: {e}")
        else:
            self.sample_rate = sample_rate

    def update_frequency(self, sample_rate: float, centre_freq: float):
        if int(sample_rate) != self.last_sample_rate:
            self.update_sample_rate(sample_rate)
        if int(centre_freq) != self.centre_freq:
            self.update_centre_frequency(centre_freq)

    # ------------------------------------------------------------------ the frame
    def _fallback(self):
        n = self.fft_size
        return np.zeros(n), np.linspace(self.centre_freq - self.sample_rate / 2,
                                        self.centre_freq + self.sample_rate / 2, n)

    def get_power_levels(self):
        if not self.running:
            return self._fallback()
        try:
            fs = self.sdr.get_sample_rate()
            fc = self.sdr.get_center_freq()
            for _ in range(self._flush_reads_remaining):
                self.sdr.read_samples(self.fft_size)
            self._flush_reads_remaining = 0
            samples = self.sdr.read_samples(self.fft_size)
            self._store_raw(np.array(samples, copy=True))
            if self._engine is None or self._engine_n != self.fft_size or self._engine_dirty \
                    or getattr(self, "_engine_fs", None) != fs:
                self._gpu_configure(self.fft_size, self.window.astype(np.float32), branch="rtl",
                                    use_psd=self.use_psd, sample_rate=fs, dc_alpha=-1.0)
                self._engine_fs = fs
            power_db = self._gpu_frame(samples).astype(np.float64)
            n = self.fft_size
            freq_bins = np.fft.fftshift(np.fft.fftfreq(n, 1 / fs)) + fc
            return power_db, freq_bins
        except Exception as e:
            logger.error("Error computing power levels: %s", e)
            return self._fallback()

    # ------------------------------------------------------------------ knobs
    def set_window_type(self, window_type: str):
        self.window = _WINDOWS.get(window_type.lower(), np.hanning)(self.fft_size)
        self._engine_dirty = True

    def set_fft_size(self, fft_size: int):
        if fft_size == self.fft_size:
            return
        self.fft_size = fft_size
        self.window = np.hanning(self.fft_size)
        self._averager.reset()
        self._engine_dirty = True

    @property
    def sample_count(self) -> int:
        return self.fft_size

    @sample_count.setter
    def sample_count(self, value: int):
        self.set_fft_size(value)

    def read_samples_only(self):
        if not self.running or self.sdr is None:
            return None
        try:
            self._store_raw(np.array(self.sdr.read_samples(self.fft_size), copy=True))
            return self._last_raw_samples
        except Exception as e:
            logger.error("Error reading samples: %s", e)
            return None

    def set_gain(self, gain) -> None:
        self._gain = gain
        if self.sdr is not None and self.running:
            try:
                self.sdr.gain = gain
            except Exception as e:
                logger.error("Error setting RTL-SDR gain: %s", e)

    def set_psd_mode(self, enabled: bool):
        self.use_psd = enabled
        self._engine_dirty = True

    def __del__(self):
        try:
            self._gpu_release()
        except Exception:
            pass
