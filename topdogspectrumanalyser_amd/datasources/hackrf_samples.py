"""HackrfSamplesDataSource - HackRF IQ source whose DSP runs on the MI355X.

Public surface and semantics follow the reference's datasources/hackrf_samples.py (class :20-728):
constructor (sample_rate, centre_freq), start/stop, reader thread + bounded queue with drop-oldest
(:191-252), "freshest chunk, last N samples" framing (:254-305), silence/underrun hold of the last good
frame (:351-355), DC tracker (:360-365), power-normalised Hann (:311-316), PSD / averaging / plain dB
branches (:374-383), setters (:392-440, :633-670), get_stats (:679-696).

What differs: the per-frame arithmetic (mean, window multiply, FFT, fftshift, abs, log10, averaging) is
ONE launch of the HIP frame kernel behind the C-ABI; and _store_raw keeps the untouched samples (the
reference stores a view it then DC-removes and windows in place - SURVEY.md 8(a) quirk i).
The USB device object is injected (`device_factory`) or comes from the `hackrf` module when installed;
without either start() raises RuntimeError exactly like the reference (:84).
"""
import logging
import queue
import threading
import time
from typing import Callable, Optional

import numpy as np

from .base import SampleDataSource
from ._gpu import GpuSpectrumMixin

try:  # pyhackrf is optional: there is no SDR hardware on a GPU box
    from hackrf import HackRF  # type: ignore
    _HACKRF_AVAILABLE = True
except (ImportError, OSError):
    HackRF = None
    _HACKRF_AVAILABLE = False

logger = logging.getLogger(__name__)


class HackrfSamplesDataSource(GpuSpectrumMixin, SampleDataSource):
    READ_CHUNK = 65536       # samples per USB read (3.3 ms at 20 Msps)
    MAX_QUEUE_SIZE = 4
    CONSUME_TIMEOUT = 0.5
    STOP_TIMEOUT = 2.0
    _DC_ALPHA = 1.0

    def __init__(self, sample_rate: int, centre_freq: int, device_factory: Optional[Callable] = None,
                 gpu_device: int = 0):
        super().__init__(sample_rate, centre_freq)
        self.num_samples = 1024
        self.device = None
        self.running = False
        self.last_sample_rate = sample_rate
        self.lna_gain, self.vga_gain, self.amplifier = 16, 20, True
        self.use_psd = False
        self._device_factory = device_factory
        self._gpu_device = gpu_device
        self._stop_requested = threading.Event()
        self._lock = threading.RLock()
        self._device_lock = threading.RLock()
        self._sample_queue: "queue.Queue[np.ndarray]" = queue.Queue(maxsize=self.MAX_QUEUE_SIZE)
        self._reader_thread: Optional[threading.Thread] = None
        self._window: Optional[np.ndarray] = None
        self._freq_bins: Optional[np.ndarray] = None
        self._reservoir = np.array([], dtype=np.complex64)
        self._last_good_power: Optional[np.ndarray] = None
        self._stats = dict(samples_dropped=0, queue_overflows=0, read_errors=0, last_read_time=0)
        self._averager._on_change = lambda mode, n: self._mark_dirty()
        self._averager._on_reset = self._gpu_reset_averager
        self._engine_dirty = True

    # ------------------------------------------------------------------ lifecycle
    def _open_device(self):
        if self._device_factory is not None:
            return self._device_factory()
        if not _HACKRF_AVAILABLE:
            raise RuntimeError("HackRF library (libhackrf) not available on this system")
        return HackRF()

    def _apply_device_settings(self) -> None:
        d = self.device
        d.set_sample_rate(self.sample_rate)
        d.set_freq(self.centre_freq)
        d.set_lna_gain(self.lna_gain)
        d.set_vga_gain(self.vga_gain)
        (d.enable_amp if self.amplifier else d.disable_amp)()

    def _setup_device(self) -> None:
        with self._device_lock:
            try:
                self.device = self._open_device()
                self._apply_device_settings()
            except Exception:
                if self.device is not None:
                    try:
                        self.device.close()
                    except Exception:
                        pass
                    self.device = None
                raise

    def start(self, frequency=None):
        if self._device_factory is None and not _HACKRF_AVAILABLE:
            raise RuntimeError("HackRF library (libhackrf) not available on this system")
        with self._lock:
            if frequency:
                self.centre_freq = int(frequency.centre)
                self.sample_rate = int(frequency.span)
            if self.running:
                logger.warning("Already running")
                return
            self._stop_requested.clear()
            self._setup_device()
            self._allocate_fft_resources()
            self._flush_buffers()
            self._spawn_reader()

    def _spawn_reader(self) -> None:
        self.running = True
        self._reader_thread = threading.Thread(target=self._reader_loop, daemon=True, name="HackRF-Reader")
        self._reader_thread.start()

    def _join_reader(self, timeout: float, force_close: bool) -> None:
        th = self._reader_thread
        if th is not None and th.is_alive():
            th.join(timeout=timeout)
            if th.is_alive() and force_close and self.device is not None:
                try:                      # a blocked USB read only returns when the device goes away
                    self.device.close()
                except Exception:
                    pass
                th.join(timeout=1.0)
        self._reader_thread = None

    def stop(self):
        with self._lock:
            if not self.running:
                return
            self.running = False
            self._stop_requested.set()
            self._join_reader(self.STOP_TIMEOUT, force_close=True)
            self._cleanup_device()
            self._flush_buffers()

    def _cleanup_device(self) -> None:
        with self._device_lock:
            if self.device is not None:
                try:
                    self.device.close()
                except Exception as e:  # pragma: no cover
                    logger.debug("error closing device: %s", e)
                finally:
                    self.device = None

    @property
    def is_running(self) -> bool:
        with self._lock:
            return self.running

    # ------------------------------------------------------------------ streaming front end
    def _reader_loop(self) -> None:
        consecutive_errors = 0
        while self.running and not self._stop_requested.is_set():
            try:
                with self._device_lock:
                    if self.device is None:
                        break
                    chunk = self.device.read_samples(self.READ_CHUNK)
                if chunk is None or len(chunk) == 0:
                    continue
                consecutive_errors = 0
                self._stats["last_read_time"] = time.time()
                try:
                    self._sample_queue.put_nowait(chunk)
                except queue.Full:                       # keep the newest data: drop the oldest chunk
                    try:
                        dropped = self._sample_queue.get_nowait()
                        self._sample_queue.put_nowait(chunk)
                        self._stats["samples_dropped"] += len(dropped)
                        self._stats["queue_overflows"] += 1
                    except (queue.Empty, queue.Full):
                        self._stats["samples_dropped"] += len(chunk)
            except Exception as e:
                consecutive_errors += 1
                self._stats["read_errors"] += 1
                if consecutive_errors >= 5:
                    logger.error("5 consecutive read errors: %s", e)
                    with self._lock:
                        self.running = False
                    break
                time.sleep(0.01)

    def _drain_newest(self) -> Optional[np.ndarray]:
        newest = None
        while True:
            try:
                newest = self._sample_queue.get_nowait()
            except queue.Empty:
                return newest

    def _take_tail(self, count: int) -> np.ndarray:
        tail = self._reservoir[-count:]
        self._reservoir = self._reservoir[:-count]
        return tail

    def _consume_samples(self, count: int):
        """Exactly `count` samples from the END of the freshest chunk (then walking backwards through
        it on later calls); None after CONSUME_TIMEOUT without enough data."""
        if count <= 0:
            return np.array([], dtype=np.complex64)
        fresh = self._drain_newest()
        if fresh is not None:
            self._reservoir = fresh
        if len(self._reservoir) >= count:
            return self._take_tail(count)
        deadline = time.time() + self.CONSUME_TIMEOUT
        while len(self._reservoir) < count:
            if time.time() > deadline:
                return None
            try:
                chunk = self._sample_queue.get(timeout=0.01)
            except queue.Empty:
                continue
            later = self._drain_newest()
            self._reservoir = later if later is not None else chunk
        return self._take_tail(count)

    # ------------------------------------------------------------------ FFT resources
    def _allocate_fft_resources(self) -> None:
        n = self.num_samples
        w = np.hanning(n).astype(np.float32)
        w /= np.sqrt(np.mean(w ** 2))                    # unit mean power (hackrf_samples.py:314-315)
        self._window = w
        self._freq_bins = np.fft.fftshift(np.fft.fftfreq(n, 1 / self.sample_rate)) + self.centre_freq
        self._mark_dirty()

    def _mark_dirty(self) -> None:
        self._engine_dirty = True

    def _ready_engine(self):
        if self._engine is None or self._engine_n != self.num_samples or self._engine_dirty:
            self._gpu_configure(self.num_samples, self._window, branch="hackrf", use_psd=self.use_psd,
                                sample_rate=self.sample_rate, dc_alpha=self._DC_ALPHA)
        return self._engine

    @property
    def _dc_estimate(self) -> complex:
        return self._engine.dc_estimate if self._engine is not None else 0j

    # ------------------------------------------------------------------ public API
    def get_samples(self) -> np.ndarray:
        with self._lock:
            if not self.running:
                return np.zeros(self.num_samples, dtype=np.complex64)
            s = self._consume_samples(self.num_samples)
            return s if s is not None else np.zeros(self.num_samples, dtype=np.complex64)

    def get_power_levels(self):
        with self._lock:
            if not self.running or self._freq_bins is None:
                bins = self._freq_bins if self._freq_bins is not None else np.zeros(self.num_samples)
                return np.zeros(self.num_samples), bins
            samples = self._consume_samples(self.num_samples)
            silent = samples is None or float(np.vdot(samples, samples).real) / len(samples) < 1e-20
            if silent:                                   # underrun / silence: hold the last good frame
                if self._last_good_power is not None:
                    return self._last_good_power, self._freq_bins
                return np.zeros(self.num_samples), self._freq_bins
            self._store_raw(samples)
            self._ready_engine()
            power_db = self._gpu_frame(samples)
            self._last_good_power = power_db
            return power_db, self._freq_bins

    def set_num_samples(self, num_samples: int):
        if num_samples <= 0:
            raise ValueError("num_samples must be positive")
        with self._lock:
            if num_samples == self.num_samples:
                return
            self.num_samples = num_samples
            self._averager.reset()
            if self.running:
                self._allocate_fft_resources()

    @property
    def sample_count(self) -> int:
        with self._lock:
            return self.num_samples

    @sample_count.setter
    def sample_count(self, value: int):
        self.set_num_samples(value)

    def read_samples_only(self):
        with self._lock:
            if not self.running:
                return None
            s = self._consume_samples(self.num_samples)
            if s is not None:
                self._store_raw(s)
            return s

    def set_psd_mode(self, enabled: bool):
        with self._lock:
            if self.use_psd != enabled:
                self.use_psd = enabled
                self._mark_dirty()

    # ------------------------------------------------------------------ retuning
    def _flush_buffers(self) -> None:
        self._drain_newest()
        self._reservoir = np.array([], dtype=np.complex64)
        self._gpu_reset_dc()
        self._last_good_power = None

    def _stop_internal(self) -> None:
        if not self.running:
            return
        self.running = False
        self._stop_requested.set()
        self._join_reader(0.5, force_close=False)
        self._flush_buffers()

    def _start_internal(self) -> None:
        if self.running:
            return
        self._stop_requested.clear()
        with self._device_lock:
            if self.device is not None:
                try:
                    self._apply_device_settings()
                except Exception:
                    self._cleanup_device()
                    self._setup_device()
            else:
                self._setup_device()
        self._allocate_fft_resources()
        self._flush_buffers()
        self._spawn_reader()

    def _retune(self, sample_rate: Optional[int], centre_freq: Optional[int]) -> None:
        was_running = self.running
        if was_running:
            self._stop_internal()
        if sample_rate is not None:
            self.sample_rate = sample_rate
            self.last_sample_rate = sample_rate
        if centre_freq is not None:
            self.centre_freq = centre_freq
        if was_running:
            self._start_internal()

    def update_centre_frequency(self, centre_freq: float):
        centre_freq = int(centre_freq)
        with self._lock:
            if centre_freq != self.centre_freq:
                self._retune(None, centre_freq)

    def update_sample_rate(self, sample_rate: float):
        sample_rate = int(sample_rate)
        with self._lock:
            if sample_rate != self.last_sample_rate:
                self._retune(sample_rate, None)

    def update_frequency(self, sample_rate: float, centre_freq: float):
        sample_rate, centre_freq = int(sample_rate), int(centre_freq)
        with self._lock:
            new_rate = sample_rate if sample_rate != self.last_sample_rate else None
            new_freq = centre_freq if centre_freq != self.centre_freq else None
            if new_rate is not None or new_freq is not None:
                self._retune(new_rate, new_freq)

    # ------------------------------------------------------------------ gain / misc controls
    def set_gains(self, lna_gain: Optional[int] = None, vga_gain: Optional[int] = None):
        with self._lock:
            if lna_gain is not None:
                if not 0 <= lna_gain <= 40:
                    raise ValueError(f"LNA gain must be between 0 and 40, got {lna_gain}")
                self.lna_gain = lna_gain
            if vga_gain is not None:
                if not 0 <= vga_gain <= 62:
                    raise ValueError(f"VGA gain must be between 0 and 62, got {vga_gain}")
                self.vga_gain = vga_gain
            if self.running and self.device is not None:
                with self._device_lock:
                    if lna_gain is not None:
                        self.device.set_lna_gain(self.lna_gain)
                    if vga_gain is not None:
                        self.device.set_vga_gain(self.vga_gain)

    def set_dc_alpha(self, alpha: float) -> None:
        self._DC_ALPHA = max(0.0, min(1.0, float(alpha)))
        self._mark_dirty()

    def set_amplifier(self, enabled: bool):
        with self._lock:
            self.amplifier = enabled
            if self.running and self.device is not None:
                with self._device_lock:
                    (self.device.enable_amp if enabled else self.device.disable_amp)()

    @property
    def amp_enabled(self) -> bool:
        return self.amplifier

    def get_stats(self) -> dict:
        with self._lock:
            st = dict(self._stats)
            dc = self._dc_estimate
            st.update(queue_size=self._sample_queue.qsize(), reservoir_size=len(self._reservoir),
                      queue_capacity=self._sample_queue.maxsize, is_running=self.running,
                      thread_alive=bool(self._reader_thread and self._reader_thread.is_alive()),
                      num_samples=self.num_samples, sample_rate=self.sample_rate,
                      centre_freq=self.centre_freq, dc_estimate_mag=abs(dc),
                      dc_estimate_phase=float(np.angle(dc)), timestamp=time.time())
            return st

    def reset_stats(self):
        with self._lock:
            self._stats = dict(samples_dropped=0, queue_overflows=0, read_errors=0, last_read_time=0)

    def __enter__(self):
        self.start()
        return self

    def __exit__(self, *exc):
        self.stop()

    def __del__(self):
        try:
            if self.running:
                self.stop()
            self._gpu_release()
        except Exception:
            pass
