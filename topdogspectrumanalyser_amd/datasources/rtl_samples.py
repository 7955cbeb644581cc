"""RTL-SDR sample source with the DSP on the MI355X.

API and observable behaviour are those of the reference's RtlSamplesDataSource
(datasources/rtl_samples.py): every frame is one blocking `read_samples(N)`; the window is the RAW
np.hanning / np.hamming / np.ones table (no power normalisation) and there is no DC removal; the trace
is `10*log10(|X|^2 + 1e-10)` after the averager, or `10*log10(|X|^2/(fs*N) + 1e-12)` in PSD mode; the
frequency axis is rebuilt on every call from what the dongle reports; a source that is not running, or
a frame that fails, yields zeros over a linspace axis instead of an exception.  One inherited oddity is
kept on purpose: resizing the FFT always goes back to a Hann window, whatever was selected before.

What differs is where the arithmetic happens: unpacked samples go to `SpectrumEngine` (frame kernel,
RTL configuration: dc_alpha = -1, raw window, power dB) through GpuSpectrumMixin.
"""
import logging
from contextlib import contextmanager
from typing import Callable, Optional

import numpy as np

from ._gpu import GpuSpectrumMixin
from .base import SampleDataSource

try:                                        # pyrtlsdr needs librtlsdr at import time
    from rtlsdr import RtlSdr  # type: ignore
except (ImportError, OSError):
    RtlSdr = None

log = logging.getLogger(__name__)

WINDOW_BUILDERS = {"hanning": np.hanning, "hamming": np.hamming, "rectangle": np.ones}
PLL_SETTLE_SECONDS = 0.006                  # samples taken this long after a retune are thrown away
MIN_SETTLE_READS = 3


@contextmanager
def _as_runtime_error(action: str):
    try:
        yield
    except Exception as exc:
        raise RuntimeError(f"{action}: {exc}")


class RtlSamplesDataSource(GpuSpectrumMixin, SampleDataSource):
    def __init__(self, sample_rate: int, centre_freq: int, device_factory: Optional[Callable] = None,
                 gpu_device: int = 0):
        super().__init__(sample_rate, centre_freq)
        self.sdr = None
        self.running = False
        self.fft_size = 1024
        self.window = np.hanning(self.fft_size)
        self.use_psd = False
        self.last_sample_rate = sample_rate
        self._gain = "auto"
        self._reads_to_discard = 0
        self._device_factory = device_factory
        self._gpu_device = gpu_device
        self._engine_dirty = True
        # averager settings are part of the GPU configuration: reconfigure / reset when they change
        self._averager._on_change = lambda mode, n: self._invalidate()
        self._averager._on_reset = self._gpu_reset_averager

    def _invalidate(self) -> None:
        self._engine_dirty = True

    # ------------------------------------------------------------------ size / window / mode
    @property
    def sample_count(self) -> int:
        return self.fft_size

    @sample_count.setter
    def sample_count(self, value: int):
        self.set_fft_size(value)

    def set_fft_size(self, fft_size: int):
        if fft_size != self.fft_size:
            self.fft_size = fft_size
            self.window = np.hanning(fft_size)          # (sic) the reference forgets the chosen window here
            self._averager.reset()
            self._invalidate()

    def set_window_type(self, window_type: str):
        build = WINDOW_BUILDERS.get(window_type.lower(), np.hanning)
        self.window = build(self.fft_size)
        self._invalidate()

    def set_psd_mode(self, enabled: bool):
        self.use_psd = enabled
        self._invalidate()

    def set_gain(self, gain) -> None:
        self._gain = gain
        if self.running and self.sdr is not None:
            try:
                self.sdr.gain = gain
            except Exception as exc:
                log.error("RTL-SDR rejected gain %r: %s", gain, exc)

    # ------------------------------------------------------------------ device lifetime
    def _new_device(self):
        if self._device_factory is not None:
            return self._device_factory()
        if RtlSdr is None:
            raise RuntimeError("RTL-SDR library (librtlsdr) not available on this system")
        return RtlSdr()

    def _adopt_hardware_rate(self) -> None:
        """The tuner rounds the requested rate; everything downstream uses what it actually runs at."""
        self.sample_rate = self.last_sample_rate = self.sdr.get_sample_rate()

    def start(self, frequency=None):
        if self._device_factory is None and RtlSdr is None:
            raise RuntimeError("RTL-SDR library (librtlsdr) not available on this system")
        if frequency:
            self.centre_freq, self.sample_rate = int(frequency.centre), int(frequency.span)
        if self.running:
            return
        try:
            with _as_runtime_error("RTL-SDR initialisation failed"):
                dev = self.sdr = self._new_device()
                dev.sample_rate, dev.center_freq, dev.gain = self.sample_rate, self.centre_freq, self._gain
                self._adopt_hardware_rate()
        except RuntimeError:
            self.running = False
            raise
        self.running = True

    def stop(self):
        dev, self.sdr, self.running = self.sdr, None, False
        if dev is not None:
            try:
                dev.close()
            except Exception as exc:
                log.error("closing the RTL-SDR failed: %s", exc)

    def pause(self):
        self.running = False

    def resume(self):
        if self.sdr is None:      # rtl_samples.py:65-71: nothing to resume, the flag stays as it is
            return
        self.running = True

    # ------------------------------------------------------------------ retune
    def update_frequency(self, sample_rate: float, centre_freq: float):
        if int(sample_rate) != self.last_sample_rate:
            self.update_sample_rate(sample_rate)
        if int(centre_freq) != self.centre_freq:
            self.update_centre_frequency(centre_freq)

    def update_sample_rate(self, sample_rate: float):
        wanted = int(sample_rate)
        if wanted == self.last_sample_rate:
            return
        if not (self.running and self.sdr is not None):
            self.sample_rate = wanted
            return
        with _as_runtime_error("Error updating sample rate"):
            self.sdr.sample_rate = wanted
            self._adopt_hardware_rate()
            self.sdr.center_freq = self.centre_freq       # a rate change can move the tuner: re-assert, re-read
            self.centre_freq = self.sdr.get_center_freq()
            self._invalidate()

    def update_centre_frequency(self, centre_freq: float):
        wanted = int(centre_freq)
        if not self.running or wanted == self.centre_freq:
            return
        self.centre_freq = wanted
        with _as_runtime_error("Error updating centre frequency"):
            self.sdr.center_freq = wanted
            frames_in_settle = int(PLL_SETTLE_SECONDS * self.sample_rate / self.fft_size)
            self._reads_to_discard = max(MIN_SETTLE_READS, frames_in_settle)

    # ------------------------------------------------------------------ frames
    def _blank(self):
        half = self.sample_rate / 2
        return np.zeros(self.fft_size), np.linspace(self.centre_freq - half, self.centre_freq + half, self.fft_size)

    def _fresh_block(self):
        while self._reads_to_discard > 0:                 # stale reads queued across a retune
            self.sdr.read_samples(self.fft_size)
            self._reads_to_discard -= 1
        return self.sdr.read_samples(self.fft_size)

    def _sync_engine(self, fs: float) -> None:
        stale = (self._engine is None or self._engine_dirty or self._engine_n != self.fft_size
                 or getattr(self, "_engine_fs", None) != fs)
        if stale:
            self._gpu_configure(self.fft_size, self.window.astype(np.float32), branch="rtl", use_psd=self.use_psd,
                                sample_rate=fs, dc_alpha=-1.0)
            self._engine_fs = fs

    def get_power_levels(self):
        if not self.running:
            return self._blank()
        try:
            fs, fc = self.sdr.get_sample_rate(), self.sdr.get_center_freq()
            block = self._fresh_block()
            self._store_raw(np.array(block, copy=True))
            self._sync_engine(fs)
            trace = self._gpu_frame(block).astype(np.float64)
            axis = np.fft.fftshift(np.fft.fftfreq(self.fft_size, 1 / fs)) + fc
            return trace, axis
        except Exception as exc:
            log.error("RTL frame failed: %s", exc)
            return self._blank()

    def read_samples_only(self):
        if self.sdr is None or not self.running:
            return None
        try:
            self._store_raw(np.array(self.sdr.read_samples(self.fft_size), copy=True))
        except Exception as exc:
            log.error("RTL read failed: %s", exc)
            return None
        return self._last_raw_samples

    def __del__(self):
        try:
            self._gpu_release()
        except Exception:
            pass
