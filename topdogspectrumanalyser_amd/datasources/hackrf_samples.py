"""HackRF sample source with the DSP on the MI355X.

API and observable behaviour are those of the reference's HackrfSamplesDataSource
(datasources/hackrf_samples.py): a reader thread pulls 65536-sample chunks off the USB device into a
four-deep queue that drops its OLDEST chunk when full; a frame is the last `num_samples` samples of the
freshest chunk (later frames walk backwards through that chunk until a newer one arrives); a frame that
is missing after 0.5 s or carries no power returns the last good trace instead; DC is removed with the
tracker `dc <- (1-a)*dc + a*mean(x)` (a = 1 by default), the window is a Hann of unit mean power, and the
trace is `20*log10(|X| + 1e-12)`, or `10*log10(avg(|X|^2) + 1e-10)` with averaging, or the PSD form.

Different on purpose: the frame arithmetic is one launch of the HIP frame kernel (`_gpu_frame`), and
`get_raw_samples()` hands back the untouched samples (the reference keeps a view that its in-place DC
removal and windowing then overwrite - SURVEY.md 8(a) quirk i).

The structure is this module's own: `_Radio` owns the USB handle, `_Inbox` owns the queue and the
"freshest chunk, newest samples first" framing, the source class ties them to the GPU engine.  The USB
object comes from `device_factory()` or from the `hackrf` module when that is installed.
"""
import logging
import queue
import threading
import time
from typing import Callable, Optional

import numpy as np

from ._gpu import GpuSpectrumMixin
from .base import SampleDataSource

try:                                        # pyhackrf needs libhackrf at import time
    from hackrf import HackRF  # type: ignore
except (ImportError, OSError):
    HackRF = None

log = logging.getLogger(__name__)

NO_LIBRARY = "HackRF library (libhackrf) not available on this system"
LNA_RANGE, VGA_RANGE = (0, 40), (0, 62)
SILENCE_POWER = 1e-20                       # mean |x|^2 below this counts as "no signal"


class _Radio:
    """The USB device and the one lock that serialises every call into it."""

    def __init__(self, factory: Optional[Callable]):
        self._factory = factory
        self.handle = None
        self.lock = threading.RLock()

    @property
    def available(self) -> bool:
        return self._factory is not None or HackRF is not None

    def open(self, **settings) -> None:
        with self.lock:
            try:
                if self._factory is not None:
                    self.handle = self._factory()
                elif HackRF is not None:
                    self.handle = HackRF()
                else:
                    raise RuntimeError(NO_LIBRARY)
                self.configure(**settings)
            except Exception:
                self.close()
                raise

    def configure(self, *, sample_rate, centre_freq, lna_gain, vga_gain, amplifier) -> None:
        dev = self.handle
        dev.set_sample_rate(sample_rate)
        dev.set_freq(centre_freq)
        dev.set_lna_gain(lna_gain)
        dev.set_vga_gain(vga_gain)
        self.amplifier(amplifier)

    def amplifier(self, on: bool) -> None:
        (self.handle.enable_amp if on else self.handle.disable_amp)()

    def read(self, n: int):
        with self.lock:
            return None if self.handle is None else self.handle.read_samples(n)

    def close(self) -> None:
        with self.lock:
            dev, self.handle = self.handle, None
        if dev is not None:
            try:
                dev.close()
            except Exception as exc:  # pragma: no cover
                log.debug("closing the HackRF failed: %s", exc)


class _Inbox:
    """Chunks from the reader thread on one side, frames for the display thread on the other."""

    def __init__(self, depth: int):
        self.chunks: "queue.Queue[np.ndarray]" = queue.Queue(maxsize=depth)
        self.current = np.array([], dtype=np.complex64)   # the chunk frames are being cut from
        self.dropped_samples = 0
        self.overflows = 0

    def offer(self, chunk) -> None:
        """Reader side: never blocks; when the queue is full the oldest chunk makes room."""
        try:
            self.chunks.put_nowait(chunk)
            return
        except queue.Full:
            pass
        try:
            stale = self.chunks.get_nowait()
            self.chunks.put_nowait(chunk)
            self.dropped_samples += len(stale)
            self.overflows += 1
        except (queue.Empty, queue.Full):                 # raced with the consumer: give up on this chunk
            self.dropped_samples += len(chunk)

    def _freshest(self):
        latest = None
        try:
            while True:
                latest = self.chunks.get_nowait()
        except queue.Empty:
            return latest

    def _cut(self, count: int) -> np.ndarray:
        frame, self.current = self.current[-count:], self.current[:-count]
        return frame

    def take(self, count: int, patience: float):
        """`count` samples from the end of the freshest chunk, or None when `patience` seconds pass."""
        latest = self._freshest()
        if latest is not None:
            self.current = latest
        give_up = time.time() + patience
        while len(self.current) < count:
            if time.time() > give_up:
                return None
            try:
                first = self.chunks.get(timeout=0.01)
            except queue.Empty:
                continue
            newer = self._freshest()
            self.current = first if newer is None else newer
        return self._cut(count)

    def clear(self) -> None:
        self._freshest()
        self.current = np.array([], dtype=np.complex64)


class HackrfSamplesDataSource(GpuSpectrumMixin, SampleDataSource):
    READ_CHUNK = 65536                      # samples per USB read (3.3 ms at 20 Msps)
    MAX_QUEUE_SIZE = 4
    CONSUME_TIMEOUT = 0.5
    STOP_TIMEOUT = 2.0
    MAX_READ_ERRORS = 5
    _DC_ALPHA = 1.0

    def __init__(self, sample_rate: int, centre_freq: int, device_factory: Optional[Callable] = None,
                 gpu_device: int = 0):
        super().__init__(sample_rate, centre_freq)
        self.num_samples = 1024
        self.running = False
        self.use_psd = False
        self.last_sample_rate = sample_rate
        self.lna_gain, self.vga_gain, self.amplifier = 16, 20, True
        self._gpu_device = gpu_device
        self._radio = _Radio(device_factory)
        self._inbox = _Inbox(self.MAX_QUEUE_SIZE)
        self._lock = threading.RLock()                    # display-thread API
        self._halt = threading.Event()
        self._reader: Optional[threading.Thread] = None
        self._read_errors = 0
        self._last_read_time = 0
        self._window: Optional[np.ndarray] = None
        self._freq_bins: Optional[np.ndarray] = None
        self._last_good_power: Optional[np.ndarray] = None
        self._engine_dirty = True
        self._engine_failed = False
        self._averager._on_change = lambda mode, n: self._mark_dirty()
        self._averager._on_reset = self._gpu_reset_averager

    # names other code (and the reference's own debugging habits) reach for
    @property
    def device(self):
        return self._radio.handle

    @property
    def _reservoir(self) -> np.ndarray:
        return self._inbox.current

    @_reservoir.setter
    def _reservoir(self, samples) -> None:
        self._inbox.current = samples

    @property
    def _sample_queue(self):
        return self._inbox.chunks

    @property
    def is_running(self) -> bool:
        with self._lock:
            return self.running

    @property
    def amp_enabled(self) -> bool:
        return self.amplifier

    def _radio_settings(self) -> dict:
        return dict(sample_rate=self.sample_rate, centre_freq=self.centre_freq, lna_gain=self.lna_gain,
                    vga_gain=self.vga_gain, amplifier=self.amplifier)

    # ------------------------------------------------------------------ start / stop
    def start(self, frequency=None):
        if not self._radio.available:
            raise RuntimeError(NO_LIBRARY)
        with self._lock:
            if frequency:
                self.centre_freq, self.sample_rate = int(frequency.centre), int(frequency.span)
            if self.running:
                log.warning("HackRF source already running")
                return
            self._halt.clear()
            self._radio.open(**self._radio_settings())
            self._go()

    def _go(self) -> None:
        """Common tail of start() and of a retune: fresh FFT tables, empty buffers, reader thread."""
        self._allocate_fft_resources()
        self._flush_buffers()
        self.running = True
        self._reader = threading.Thread(target=self._reader_loop, name="HackRF-Reader", daemon=True)
        self._reader.start()

    def _park_reader(self, patience: float, yank_device: bool) -> None:
        thread, self._reader = self._reader, None
        if thread is None or not thread.is_alive():
            return
        thread.join(timeout=patience)
        if thread.is_alive() and yank_device:             # a blocked USB read returns only when the device goes
            self._radio.close()
            thread.join(timeout=1.0)

    def stop(self):
        with self._lock:
            if not self.running:
                return
            self.running = False
            self._halt.set()
            self._park_reader(self.STOP_TIMEOUT, yank_device=True)
            self._radio.close()
            self._flush_buffers()

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

    # ------------------------------------------------------------------ reader thread
    def _reader_loop(self) -> None:
        failures = 0
        while self.running and not self._halt.is_set():
            try:
                if self._radio.handle is None:
                    return
                chunk = self._radio.read(self.READ_CHUNK)
                if chunk is None or len(chunk) == 0:
                    continue
                failures = 0
                self._last_read_time = time.time()
                self._inbox.offer(chunk)
            except Exception as exc:
                failures += 1
                self._read_errors += 1
                if failures >= self.MAX_READ_ERRORS:
                    log.error("%d consecutive HackRF read errors, giving up: %s", failures, exc)
                    with self._lock:
                        self.running = False
                    return
                time.sleep(0.01)

    def _consume_samples(self, count: int):
        if count <= 0:
            return np.array([], dtype=np.complex64)
        return self._inbox.take(count, self.CONSUME_TIMEOUT)

    def _flush_buffers(self) -> None:
        self._inbox.clear()
        self._gpu_reset_dc()
        self._last_good_power = None

    # ------------------------------------------------------------------ FFT set-up
    def _allocate_fft_resources(self) -> None:
        n = self.num_samples
        taper = np.hanning(n).astype(np.float32)
        self._window = taper / np.sqrt(np.mean(taper ** 2, dtype=np.float32))     # unit mean power
        self._freq_bins = np.fft.fftshift(np.fft.fftfreq(n, 1 / self.sample_rate)) + self.centre_freq
        self._mark_dirty()

    def _mark_dirty(self) -> None:
        self._engine_dirty = True

    def _ready_engine(self):
        if self._engine_dirty or self._engine is None or self._engine_n != self.num_samples:
            self._gpu_configure(self.num_samples, self._window, branch="hackrf", use_psd=self.use_psd,
                                sample_rate=self.sample_rate, dc_alpha=self._DC_ALPHA)
        return self._engine

    @property
    def _dc_estimate(self) -> complex:
        return 0j if self._engine is None else self._engine.dc_estimate

    # ------------------------------------------------------------------ frames
    def _nothing(self) -> np.ndarray:
        return np.zeros(self.num_samples)

    def get_power_levels(self):
        with self._lock:
            axis = self._freq_bins
            if axis is None or not self.running:
                return self._nothing(), (self._nothing() if axis is None else axis)
            frame = self._consume_samples(self.num_samples)
            if frame is None or np.mean(np.abs(frame) ** 2) < SILENCE_POWER:
                held = self._last_good_power               # underrun or silence: keep showing the last trace
                return (self._nothing() if held is None else held), axis
            self._store_raw(frame)
            try:
                self._ready_engine()
                trace = self._gpu_frame(frame)
            except (ValueError, RuntimeError) as exc:      # e.g. an FFT size the device library has no plan
                if not self._engine_failed:                # for: get_power_levels() never raises
                    log.error("GPU spectrum path unavailable for %d points: %s", self.num_samples, exc)
                self._engine_failed = True
                return self._nothing(), axis
            self._engine_failed = False
            self._last_good_power = trace
            return trace, axis

    def get_samples(self) -> np.ndarray:
        with self._lock:
            frame = self._consume_samples(self.num_samples) if self.running else None
            return np.zeros(self.num_samples, dtype=np.complex64) if frame is None else frame

    def read_samples_only(self):
        with self._lock:
            if not self.running:
                return None
            frame = self._consume_samples(self.num_samples)
            if frame is not None:
                self._store_raw(frame)
            return frame

    # ------------------------------------------------------------------ settings
    @property
    def sample_count(self) -> int:
        with self._lock:
            return self.num_samples

    @sample_count.setter
    def sample_count(self, value: int):
        self.set_num_samples(value)

    def set_num_samples(self, num_samples: int):
        if num_samples <= 0:
            raise ValueError("num_samples must be positive")
        with self._lock:
            if num_samples != self.num_samples:
                self.num_samples = num_samples
                self._averager.reset()
                if self.running:
                    self._allocate_fft_resources()

    def set_psd_mode(self, enabled: bool):
        with self._lock:
            if enabled != self.use_psd:
                self.use_psd = enabled
                self._mark_dirty()

    def set_dc_alpha(self, alpha: float) -> None:
        self._DC_ALPHA = min(1.0, max(0.0, float(alpha)))
        self._mark_dirty()

    @staticmethod
    def _checked_gain(name: str, value: int, limits) -> int:
        lo, hi = limits
        if not lo <= value <= hi:
            raise ValueError(f"{name} gain must be between {lo} and {hi}, got {value}")
        return value

    def set_gains(self, lna_gain: Optional[int] = None, vga_gain: Optional[int] = None):
        with self._lock:
            if lna_gain is not None:
                self.lna_gain = self._checked_gain("LNA", lna_gain, LNA_RANGE)
            if vga_gain is not None:
                self.vga_gain = self._checked_gain("VGA", vga_gain, VGA_RANGE)
            if self.running and self._radio.handle is not None:
                with self._radio.lock:
                    if lna_gain is not None:
                        self._radio.handle.set_lna_gain(self.lna_gain)
                    if vga_gain is not None:
                        self._radio.handle.set_vga_gain(self.vga_gain)

    def set_amplifier(self, enabled: bool):
        with self._lock:
            self.amplifier = enabled
            if self.running and self._radio.handle is not None:
                with self._radio.lock:
                    self._radio.amplifier(enabled)

    # ------------------------------------------------------------------ retune
    def _retune(self, sample_rate: Optional[int], centre_freq: Optional[int]) -> None:
        """Change rate and / or centre frequency; a running stream is paused around the change."""
        live = self.running
        if live:                                          # pause: reader off, buffers dropped, device kept
            self.running = False
            self._halt.set()
            self._park_reader(0.5, yank_device=False)
            self._flush_buffers()     # also drops the DC estimate: the reference saves and restores it around the
            #                           restart (hackrf_samples.py:473-482) but its _start_internal flushes again
            #                           (:615), so the estimate does start from zero after a retune there too
        if sample_rate is not None:
            self.sample_rate = self.last_sample_rate = sample_rate
        if centre_freq is not None:
            self.centre_freq = centre_freq
        if live:
            self._halt.clear()
            with self._radio.lock:
                try:
                    if self._radio.handle is None:
                        raise RuntimeError("device lost")
                    self._radio.configure(**self._radio_settings())
                except Exception:                         # re-open once if the handle went bad
                    self._radio.close()
                    self._radio.open(**self._radio_settings())
            self._go()

    def update_frequency(self, sample_rate: float, centre_freq: float):
        rate, freq = int(sample_rate), int(centre_freq)
        with self._lock:
            new_rate = rate if rate != self.last_sample_rate else None
            new_freq = freq if freq != self.centre_freq else None
            if new_rate is not None or new_freq is not None:
                self._retune(new_rate, new_freq)

    def update_sample_rate(self, sample_rate: float):
        self.update_frequency(sample_rate, self.centre_freq)

    def update_centre_frequency(self, centre_freq: float):
        self.update_frequency(self.last_sample_rate, centre_freq)

    # ------------------------------------------------------------------ statistics
    def get_stats(self) -> dict:
        with self._lock:
            dc = self._dc_estimate
            reader = self._reader
            return dict(samples_dropped=self._inbox.dropped_samples, queue_overflows=self._inbox.overflows,
                        read_errors=self._read_errors, last_read_time=self._last_read_time,
                        queue_size=self._inbox.chunks.qsize(), queue_capacity=self._inbox.chunks.maxsize,
                        reservoir_size=len(self._inbox.current), is_running=self.running,
                        thread_alive=bool(reader and reader.is_alive()), num_samples=self.num_samples,
                        sample_rate=self.sample_rate, centre_freq=self.centre_freq, dc_estimate_mag=abs(dc),
                        dc_estimate_phase=float(np.angle(dc)), timestamp=time.time())

    def reset_stats(self):
        with self._lock:
            self._inbox.dropped_samples = self._inbox.overflows = 0
            self._read_errors = self._last_read_time = 0
