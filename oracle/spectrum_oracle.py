"""CPU oracle for the "sample method" IQ -> spectrum hot path.  TEST INFRASTRUCTURE ONLY.

This module is a plain-numpy restatement of the arithmetic the reference
(CWNE88/topdogspectrumanalyser, mounted at /root/reference in the build container) performs
between "a frame of IQ samples arrives" and "dB trace + hold traces are handed to a widget".
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the reported CPU baseline.  Nothing under
``topdogspectrumanalyser_amd/`` imports it: the product path is the HIP library and fails
loudly when that library is missing.

Parity status: the reference's own tests never execute an FFT (scipy.fft is mocked in every
test), so the numerics of this path are *unpinned by the reference's tests*.
They are pinned instead by ``tests/golden/*.npz``: inputs + outputs captured from the imported
reference itself by ``tests/golden/make_golden.py`` (run in the build container, hardware
modules mocked).  ``tests/test_oracle_golden.py`` checks every function below against them
(bit-identical for the ``precision="ref"`` variants).

Two precisions are offered everywhere:

* ``precision="ref"``   - reproduces the reference's dtypes under numpy >= 2 (complex64 FFT
  stays float32 on the HackRF branch; RTL/audio branches are float64 because their windows are).
* ``precision="gold"``  - the same equations with every array promoted to complex128/float64
  before the first arithmetic step.  This is what the float32 GPU output is compared against
  (tolerance 1e-4 relative to the frame maximum in linear power, BASELINE.json north_star),
  because the reference's own precision depends on the numpy version (pinned 1.26.4 upcasts
  complex64 FFTs to float64, 2.x does not).

Each function cites the reference file:line it restates.
"""

from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

# utils/constants.py:152-155 (DSPConstants) and :141 (UIConstants.TARE_NUM_SAMPLES)
LOG_FLOOR = 1e-12
POWER_LOG_FLOOR = 1e-10
TARE_NUM_SAMPLES = 32


# ----------------------------------------------------------------------------------------------
# a1 - int8 -> complex unpack.  Not in /root/reference (lives in pyhackrf==0.2.0 / pyrtlsdr==0.3.0,
# requirements.txt:80,89; call sites hackrf_samples.py:207, rtl_samples.py:167).  Build contract
# from SURVEY.md section 8(a) row a1: interleaved [I0,Q0,I1,Q1,...] int8 -> (I + jQ)/128 float32.
# ----------------------------------------------------------------------------------------------
def unpack_iq_int8(iq: np.ndarray) -> np.ndarray:
    iq = np.asarray(iq, dtype=np.int8)
    f = iq.astype(np.float32) * np.float32(1.0 / 128.0)
    return (f[0::2] + 1j * f[1::2]).astype(np.complex64)


def unpack_iq_uint8_rtl(iq: np.ndarray) -> np.ndarray:
    """pyrtlsdr convention (packed_bytes_to_iq): (u8 / 127.5) - 1, complex128."""
    iq = np.asarray(iq, dtype=np.uint8)
    f = iq.astype(np.float64) / 127.5 - 1.0
    return f[0::2] + 1j * f[1::2]


# a2 - deterministic batch framing (frame k = iq[k*hop : k*hop + N]).
def num_frames(n_samples: int, nfft: int, hop: int) -> int:
    if n_samples < nfft:
        return 0
    return (n_samples - nfft) // hop + 1


def frame(iq_c: np.ndarray, nfft: int, hop: int, k: int) -> np.ndarray:
    return iq_c[k * hop: k * hop + nfft]


# ----------------------------------------------------------------------------------------------
# a5 - windows
# ----------------------------------------------------------------------------------------------
def hackrf_window(n: int) -> np.ndarray:
    """hackrf_samples.py:311-316 - symmetric Hann, float32, normalised to unit mean power."""
    window = np.hanning(n).astype(np.float32)
    window /= np.sqrt(np.mean(window ** 2))
    return window


def rtl_window(kind: str, n: int) -> np.ndarray:
    """rtl_samples.py:22,199-206,213 / audio_samples.py:35-37 - raw float64 window."""
    funcs = {"hanning": np.hanning, "hamming": np.hamming, "rectangle": np.ones}
    return funcs.get(kind.lower(), np.hanning)(n)


# a12 - frequency axis
def shifted_freq_bins(n: int, fs: float, fc: float) -> np.ndarray:
    """hackrf_samples.py:318-323, rtl_samples.py:188."""
    return np.fft.fftshift(np.fft.fftfreq(n, 1 / fs)) + fc


def audio_freq_bins(n: int, fs: float) -> np.ndarray:
    """audio_samples.py:117-119."""
    return np.linspace(0, fs / 2, n // 2 + 1)


# ----------------------------------------------------------------------------------------------
# a10 - TraceAverager (utils/signal_processing.py:5-73)
# ----------------------------------------------------------------------------------------------
class TraceAveragerOracle:
    def __init__(self):
        self.mode = "off"
        self.n = 1
        self.buffer: Optional[np.ndarray] = None
        self.count = 0

    def set_mode(self, mode: str, n: int) -> None:       # :19-28
        self.mode = mode
        self.n = max(1, n)
        self.reset()

    def reset(self) -> None:                             # :30-33
        self.buffer = None
        self.count = 0

    @property
    def is_active(self) -> bool:                         # :63-65
        return self.mode != "off" and self.n > 1

    def process(self, linear_power: np.ndarray) -> np.ndarray:   # :35-61
        if self.mode == "off" or self.n <= 1:
            return linear_power
        if self.buffer is None or self.buffer.shape != linear_power.shape:
            self.buffer = linear_power.astype(np.float64).copy()
            self.count = 1
            return self.buffer
        if self.mode == "exp":
            alpha = 1.0 / self.n
            self.buffer *= (1.0 - alpha)
            self.buffer += alpha * linear_power
        elif self.mode == "lin":
            if self.count < self.n:
                self.count += 1
            self.buffer += (linear_power - self.buffer) / self.count
        return self.buffer


# ----------------------------------------------------------------------------------------------
# HackRF branch: hackrf_samples.py:339-386 (a3 silence guard is the caller's business)
# ----------------------------------------------------------------------------------------------
class HackrfBranchOracle:
    def __init__(self, nfft: int, sample_rate: float, dc_alpha: float = 1.0,
                 use_psd: bool = False, precision: str = "ref"):
        assert precision in ("ref", "gold")
        self.nfft = nfft
        self.sample_rate = sample_rate
        self.dc_alpha = dc_alpha                       # hackrf_samples.py:32, setter :654-657
        self.use_psd = use_psd
        self.precision = precision
        self.window = hackrf_window(nfft)
        self.dc_estimate = 0.0 + 0.0j                  # :62
        self.averager = TraceAveragerOracle()

    def power_levels(self, samples: np.ndarray) -> np.ndarray:
        if self.precision == "gold":
            samples = np.asarray(samples).astype(np.complex128)
            window = self.window.astype(np.float64)
        else:
            samples = np.array(samples, dtype=np.complex64, copy=True)
            window = self.window
        mean = np.mean(samples)                                          # :360
        self.dc_estimate = (1.0 - self.dc_alpha) * self.dc_estimate + self.dc_alpha * mean  # :361-364
        samples = samples - (self.dc_estimate if self.precision == "gold"
                             else np.complex64(self.dc_estimate))       # :365 (in-place c64 -=)
        samples = samples * window                                       # :368
        spectrum = np.fft.fftshift(np.fft.fft(samples))                  # :370
        magnitude = np.abs(spectrum)                                     # :372
        if self.use_psd:                                                 # :374-377
            psd = (magnitude ** 2) / (self.sample_rate * self.nfft)
            psd = self.averager.process(psd)
            return 10 * np.log10(psd + LOG_FLOOR)
        if self.averager.is_active:                                      # :378-381
            power = self.averager.process(magnitude ** 2)
            return 10 * np.log10(power + POWER_LOG_FLOOR)
        return 20 * np.log10(magnitude + LOG_FLOOR)                      # :382-383


# ----------------------------------------------------------------------------------------------
# RTL branch: rtl_samples.py:148-197
# ----------------------------------------------------------------------------------------------
class RtlBranchOracle:
    def __init__(self, nfft: int, sample_rate: float, window: str = "hanning",
                 use_psd: bool = False, precision: str = "ref"):
        self.nfft = nfft
        self.sample_rate = sample_rate
        self.use_psd = use_psd
        self.precision = precision
        self.window = rtl_window(window, nfft)
        self.averager = TraceAveragerOracle()

    def power_levels(self, samples: np.ndarray) -> np.ndarray:
        from scipy import fft as sfft
        samples = np.asarray(samples)
        if self.precision == "gold":
            samples = samples.astype(np.complex128)
        samples = samples * self.window                                  # :169 (float64 window)
        spectrum = sfft.fftshift(sfft.fft(samples, n=self.nfft))         # :170-173
        if self.use_psd:                                                 # :175-179
            psd = (np.abs(spectrum) ** 2) / (self.sample_rate * self.nfft)
            psd = self.averager.process(psd)
            return 10 * np.log10(psd + LOG_FLOOR)
        power = np.abs(spectrum) ** 2                                    # :181-184
        power = self.averager.process(power)
        return 10 * np.log10(power + POWER_LOG_FLOOR)


def welch_gold(segment, n_segments: int, nfft: int, sample_rate: float, threads: int = 1) -> np.ndarray:
    """Float64 gold of a Welch capture in the RTL branch (BASELINE config 5): segment(k) -> complex samples of segment k;
    returns get_power_levels() of the LAST segment with `lin` averaging over all of them (rtl_samples.py:167-184 +
    TraceAverager.process, utils/signal_processing.py:56-59) - exactly RtlBranchOracle(precision="gold") called segment
    by segment.  The segments' power spectra are independent of each other: `threads` host threads form them (numpy and
    scipy release the GIL in their loops), the running mean takes them in capture order, so the result does not depend
    on `threads` (tests/test_oracle_golden.py)."""
    from concurrent.futures import ThreadPoolExecutor
    from scipy import fft as sfft
    window = rtl_window("hanning", nfft)

    def power(k: int) -> np.ndarray:
        x = np.asarray(segment(k)).astype(np.complex128) * window        # :169
        return np.abs(sfft.fftshift(sfft.fft(x, n=nfft))) ** 2          # :170-173, :181
    avg = TraceAveragerOracle()
    avg.set_mode("lin", n_segments)
    out = None
    with ThreadPoolExecutor(max(1, threads)) as ex:
        block = max(1, threads) * 2                                      # bounded memory: two blocks of spectra alive
        for k0 in range(0, n_segments, block):
            for p in ex.map(power, range(k0, min(n_segments, k0 + block))):
                out = avg.process(p)                                     # :184 (float64 state)
    return 10 * np.log10(out + POWER_LOG_FLOOR)


# ----------------------------------------------------------------------------------------------
# Audio branch: audio_samples.py:121-132 (_compute_power) and :158-180 (dB + floors)
# ----------------------------------------------------------------------------------------------
def audio_compute_power(signal: np.ndarray, window: np.ndarray, nfft: int, sample_rate: float,
                        use_psd: bool, precision: str = "ref") -> np.ndarray:
    from scipy import fft as sfft
    signal = np.asarray(signal)
    if precision == "gold":
        signal = signal.astype(np.float64)
    signal = signal - signal.mean()                                      # :123
    signal *= window                                                     # :124 (in place: keeps the
    #                                                                      signal dtype, float32 in "ref")
    spectrum = sfft.rfft(signal, n=nfft)                                 # :125
    if use_psd:
        power = (np.abs(spectrum) ** 2) / (sample_rate * nfft)           # :127
    else:
        power = np.abs(spectrum) ** 2                                    # :129
    power[1:-1] *= 2                                                     # :131
    return power


def audio_db(power: np.ndarray, use_psd: bool) -> np.ndarray:
    floor = LOG_FLOOR if use_psd else POWER_LOG_FLOOR                    # :160
    return 10 * np.log10(power + floor)                                  # :164,179


# ----------------------------------------------------------------------------------------------
# DataProcessor DSP helpers: core/display_data_processor.py
# ----------------------------------------------------------------------------------------------
def apply_cal_offset(power_db: np.ndarray, offset: float) -> np.ndarray:
    """:317-327 - add the per-source scalar, new array, only when offset != 0."""
    return power_db + offset if offset != 0.0 else power_db


def nan_safe(arr: np.ndarray, fill: float) -> np.ndarray:
    """:473-480."""
    if not np.any(np.isnan(arr)):
        return arr
    out = arr.copy()
    out[np.isnan(out)] = fill
    return out


class TareOracle:
    """:329-369 with core/tare_state.py:9-13."""

    def __init__(self):
        self.collecting = False
        self.buffer: Optional[np.ndarray] = None
        self.count = 0
        self.active = False
        self.baseline: Optional[np.ndarray] = None

    def start_collecting(self) -> None:
        self.collecting, self.buffer, self.count = True, None, 0

    def apply(self, power_db: np.ndarray) -> np.ndarray:
        if self.collecting:
            linear = 10.0 ** (power_db / 10.0)
            if self.buffer is None or self.buffer.shape != linear.shape:
                self.buffer = linear.copy()
                self.count = 1
            else:
                self.buffer += linear
                self.count += 1
            if self.count >= TARE_NUM_SAMPLES:
                avg_linear = self.buffer / self.count
                self.baseline = 10.0 * np.log10(np.maximum(avg_linear, 1e-30))
                self.active = True
                self.collecting, self.buffer, self.count = False, None, 0
        if self.active and self.baseline is not None:
            if power_db.shape != self.baseline.shape:
                self.active, self.baseline = False, None
            else:
                power_db = power_db - self.baseline
        return power_db


class HoldOracle:
    """_update_max_hold :371-382 / _update_min_hold :384-395.

    ``alias_quirk=False`` (default, the parity target for the HIP build): independent running
    fmax / fmin traces - the documented intent, and exactly what the reference computes when only
    one of the two holds is enabled.

    ``alias_quirk=True`` reproduces what the reference really does when max AND min hold are both
    enabled from the same first frame: _nan_safe (:473-480) returns a clean array uncopied, so
    mw.max_power_levels and mw.min_power_levels are one ndarray and each later frame leaves
    fmin(fmax(h, x), x) == x in both traces (SURVEY.md 8(a) quirk ii).  Pinned by
    tests/golden/processor_1024.npz; deliberately NOT carried into the product.
    """

    def __init__(self, max_on: bool = True, min_on: bool = True, alias_quirk: bool = False):
        self.max_on, self.min_on, self.alias_quirk = max_on, min_on, alias_quirk
        self.max: Optional[np.ndarray] = None
        self.min: Optional[np.ndarray] = None

    def update(self, power_db: np.ndarray) -> None:
        if self.max_on:
            if self.max is None or self.max.shape != power_db.shape:
                m = nan_safe(power_db, -500.0)
                self.max = m if self.alias_quirk else np.array(m, copy=True)
            else:
                np.fmax(self.max, power_db, out=self.max)
        if self.min_on:
            if self.min is None or self.min.shape != power_db.shape:
                m = nan_safe(power_db, 500.0)
                self.min = m if self.alias_quirk else np.array(m, copy=True)
            else:
                np.fmin(self.min, power_db, out=self.min)


# ----------------------------------------------------------------------------------------------
# Batch drivers (what bench.py's cpu_baseline and the parity tests call)
# ----------------------------------------------------------------------------------------------
def hackrf_batch(iq_i8: np.ndarray, nfft: int, hop: int, sample_rate: float, *,
                 n_frames: Optional[int] = None, use_psd: bool = False,
                 avg: Tuple[str, int] = ("off", 1), dc_alpha: float = 1.0,
                 cal_offset_db: float = 0.0, precision: str = "gold",
                 hold: bool = True):
    """int8 IQ -> [frames, N] dB (+ max/min hold) with HackRF-branch semantics."""
    x = unpack_iq_int8(iq_i8)
    nf = num_frames(len(x), nfft, hop) if n_frames is None else n_frames
    br = HackrfBranchOracle(nfft, sample_rate, dc_alpha, use_psd, precision)
    br.averager.set_mode(*avg)
    out = np.empty((nf, nfft), dtype=np.float64 if precision == "gold" else np.float32)
    h = HoldOracle()
    for k in range(nf):
        db = np.asarray(br.power_levels(frame(x, nfft, hop, k)))
        db = apply_cal_offset(db, cal_offset_db)
        out[k] = db
        if hold:
            h.update(out[k])
    return out, h.max, h.min


def rtl_batch(iq_i8: np.ndarray, nfft: int, hop: int, sample_rate: float, *,
              n_frames: Optional[int] = None, window: str = "hanning", use_psd: bool = False,
              avg: Tuple[str, int] = ("off", 1), cal_offset_db: float = 0.0,
              precision: str = "gold", hold: bool = True):
    """int8 IQ -> [frames, N] dB with RTL-branch semantics (no DC removal, raw window)."""
    x = unpack_iq_int8(iq_i8)
    nf = num_frames(len(x), nfft, hop) if n_frames is None else n_frames
    br = RtlBranchOracle(nfft, sample_rate, window, use_psd, precision)
    br.averager.set_mode(*avg)
    out = np.empty((nf, nfft), dtype=np.float64)
    h = HoldOracle()
    for k in range(nf):
        db = apply_cal_offset(np.asarray(br.power_levels(frame(x, nfft, hop, k))), cal_offset_db)
        out[k] = db
        if hold:
            h.update(out[k])
    return out, h.max, h.min


def max_hold_at_positions(iq_i8: np.ndarray, nfft: int, hop: int, positions, branch: str = "hackrf",
                          n_frames: Optional[int] = None, chunk: int = 128, allowance_units: float = 0.0):
    """float64 gold of the max-hold trace of a whole capture at a FEW fftshift-ed positions, evaluated from the DFT
    definition X[k] = sum_n x[n] exp(-2 pi i k n / N) as one matrix product per chunk of frames - what bench.py checks
    the (combined) hold trace against where a full FFT of every frame of every rank would take minutes.  Per frame the
    arithmetic is the branch's own, in float64: HackRF (hackrf_samples.py:360-383: mean removal, power-normalised
    Hann, 20 log10(|X| + 1e-12)) or RTL (rtl_samples.py:169-184: raw Hann, 10 log10(|X|^2 + 1e-10)); the hold is
    np.fmax over the frames (core/display_data_processor.py:371-382).
    allowance_units > 0: returns (hold, allowance) - the |dB| allowance of the trace at those positions by HoldAllowance's
    rule (the largest allowance any held frame had there, with `allowance_units` float32 rounding units as the amplitude
    floor); a frame's maximum is taken over the positions given, so they must include its strongest bin."""
    x = unpack_iq_int8(iq_i8)
    nf = num_frames(len(x), nfft, hop) if n_frames is None else n_frames
    pos = np.asarray(positions, dtype=np.int64)
    k = (pos + nfft // 2) % nfft                                   # fftshift: shifted[p] = X[(p + N/2) mod N]
    n = np.arange(nfft, dtype=np.int64)
    e = np.exp(-2j * np.pi * ((n[:, None] * k[None, :]) % nfft) / nfft)
    if branch == "hackrf":
        w = hackrf_window(nfft).astype(np.float64)
    else:
        w = rtl_window("hanning", nfft).astype(np.float64)
    ew = e * w[:, None]
    frames = np.lib.stride_tricks.as_strided(x, shape=(nf, nfft), strides=(hop * x.strides[0], x.strides[0]),
                                             writeable=False)
    hold = np.full(len(pos), -np.inf)
    allow = np.zeros(len(pos))
    for f0 in range(0, nf, chunk):
        blk = frames[f0: f0 + chunk].astype(np.complex128)
        if branch == "hackrf":
            blk = blk - blk.mean(axis=1, keepdims=True)
            db = 20.0 * np.log10(np.abs(blk @ ew) + LOG_FLOOR)
        else:
            db = 10.0 * np.log10(np.abs(blk @ ew) ** 2 + POWER_LOG_FLOOR)
        hold = np.fmax(hold, db.max(axis=0))
        if allowance_units > 0:
            allow = np.maximum(allow, row_allowance_db(db, 100.0, allowance_units * AMP_FLOOR).max(axis=0))
    return (hold, allow) if allowance_units > 0 else hold


# ----------------------------------------------------------------------------------------------
# Synthetic IQ of SURVEY.md section 8(d): 3 tones + DC + complex Gaussian noise, int8 interleaved
# ----------------------------------------------------------------------------------------------
def synth_iq_int8(n_samples: int, nfft: int, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    n = np.arange(n_samples, dtype=np.float64)
    bins = (nfft / 8 + 0.0, -nfft / 5 + 0.3, 3 * nfft / 7 + 0.5)
    amps = (40.0, 12.0, 3.0)
    sig = np.zeros(n_samples, dtype=np.complex128)
    for b, a in zip(bins, amps):
        sig += a * np.exp(2j * np.pi * b * n / nfft)
    sig += (2.0 + 1.0j)
    sig += 4.0 * (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples)) / np.sqrt(2.0)
    out = np.empty(2 * n_samples, dtype=np.int8)
    out[0::2] = np.clip(np.rint(sig.real), -128, 127).astype(np.int8)
    out[1::2] = np.clip(np.rint(sig.imag), -128, 127).astype(np.int8)
    return out


AMP_FLOOR = 2.0 ** -24   # 5.96e-8: the float32 rounding unit, relative to the frame's largest amplitude


def parity_metrics(db_gpu: np.ndarray, db_gold: np.ndarray, floor_rel_db: float = 100.0,
                   amp_floor: float = AMP_FLOOR):
    """SURVEY.md 8(d) parity definition, per frame:
      rel - linear power error relative to the frame maximum (bound 1e-4, every bin);
      ddb - |dB| error on the bins within ``floor_rel_db`` (100 dB) of the frame maximum, bound 1e-3 dB.
    A float32 FFT cannot resolve an amplitude difference below about one ulp of the LARGEST amplitude it
    carries: where a bin is so deep that ``amp_floor`` * A_max (2^-24 A_max, the float32 rounding unit) is worth
    more than 1e-3 dB of that bin - from 66 dB below the maximum on - the allowance is that amplitude instead.  (On MI355X the errors
    sit at 1.2e-8 * A_max in the 31 bins that share the last radix-32 butterfly with a strong tone and at
    5e-10 * A_max elsewhere; numpy's own float32 path reaches 6e-7 * A_max next to a tone.)  `ddb` is
    returned scaled to the 1e-3 dB bound, i.e. max(|dB error| / allowance) * 1e-3, so ``ddb <= 1e-3`` is
    the test.

    Which ``amp_floor`` is a BOUND (tools/parity_soak.py, 3000 random configurations + 120 long frames on MI355X):
    in units of one rounding unit the worst bin of a case sits at a median of 0.14, 99 % of the cases stay below 0.9,
    the worst seen is 1.38 - always the one bin N/2 away from a full-scale tone that falls exactly on a bin, where
    the last radix-2 stage cancels two half-amplitude terms; long frames (two FFT kernels) worst 1.03; the tracked
    DC remover worst 1.19 (3.8 while its estimate still reached the frame kernel folded into one float32
    "128 + dc").  The tests therefore bound with TWO units (2^-23 A_max)."""
    db_gpu = np.asarray(db_gpu, dtype=np.float64)
    db_gold = np.asarray(db_gold, dtype=np.float64)
    p_gpu = 10.0 ** (db_gpu / 10.0)
    p_gold = 10.0 ** (db_gold / 10.0)
    pmax = p_gold.max(axis=-1, keepdims=True)
    rel = np.abs(p_gpu - p_gold) / pmax
    depth = db_gold.max(axis=-1, keepdims=True) - db_gold
    mask = depth <= floor_rel_db
    allowance = np.maximum(1e-3, (20.0 / np.log(10.0)) * amp_floor * 10.0 ** (depth / 20.0))
    with np.errstate(invalid="ignore"):
        ddb = np.where(mask, np.abs(db_gpu - db_gold) / allowance, 0.0) * 1e-3
    return float(rel.max()), float(np.nanmax(ddb))


def row_allowance_db(db_gold: np.ndarray, floor_rel_db: float = 100.0, amp_floor: float = AMP_FLOOR) -> np.ndarray:
    """The |dB| allowance parity_metrics grants every bin of every row (same shape as db_gold); +inf where the bin lies
    more than floor_rel_db below its frame's maximum (not checked)."""
    db_gold = np.asarray(db_gold, dtype=np.float64)
    depth = db_gold.max(axis=-1, keepdims=True) - db_gold
    allowance = np.maximum(1e-3, (20.0 / np.log(10.0)) * amp_floor * 10.0 ** (depth / 20.0))
    return np.where(depth <= floor_rel_db, allowance, np.inf)


class HoldAllowance:
    """What a max / min hold trace may differ from the gold trace by - NOT a statistical figure of its own but what
    follows from the rows' allowance: for any two sequences |max_f a_f - max_f b_f| <= max_f |a_f - b_f| (the same for
    min), bin by bin, so a trace held over rows that each keep their allowance differs from the gold trace by at most
    the LARGEST allowance any held row had at that bin (and its linear power by at most 1e-4 of the largest frame
    maximum).  Judging a hold trace by parity_metrics instead - i.e. against the allowance of a single row whose
    maximum is the trace's own - is wrong on two counts: a min-hold trace keeps per bin the lowest of up to hundreds
    of values, an extreme of as many draws of the rounding error (round 5's soak: 2.49 units on a trace whose rows
    stood at 0.1), and the trace's own maximum is not the amplitude any frame's transform carried.
    (core/display_data_processor.py:371-395: np.fmax / np.fmin over the displayed rows.)

    update(gold_rows) with every batch of rows the trace has seen since its reset, then metrics(trace_gpu, trace_gold)
    -> (rel, ddb) with the meaning and bounds of parity_metrics (rel <= 1e-4, ddb <= 1e-3)."""

    def __init__(self, floor_rel_db: float = 100.0, amp_floor: float = AMP_FLOOR):
        self.floor_rel_db, self.amp_floor = floor_rel_db, amp_floor
        self.allow = None
        self.pmax = 0.0

    def update(self, gold_rows: np.ndarray) -> "HoldAllowance":
        rows = np.atleast_2d(np.asarray(gold_rows, dtype=np.float64))
        a = row_allowance_db(rows, self.floor_rel_db, self.amp_floor).max(axis=0)
        self.allow = a if self.allow is None else np.maximum(self.allow, a)
        self.pmax = max(self.pmax, float(10.0 ** (rows.max() / 10.0)))
        return self

    def metrics(self, trace_gpu: np.ndarray, trace_gold: np.ndarray):
        g = np.asarray(trace_gpu, dtype=np.float64)
        d = np.asarray(trace_gold, dtype=np.float64)
        rel = np.abs(10.0 ** (g / 10.0) - 10.0 ** (d / 10.0)) / self.pmax
        with np.errstate(invalid="ignore"):
            ddb = np.where(np.isfinite(self.allow), np.abs(g - d) / self.allow, 0.0) * 1e-3
        return float(rel.max()), float(np.nanmax(ddb))


def parity_raw_db(db_gpu: np.ndarray, db_gold: np.ndarray, floor_rel_db: float) -> float:
    """largest plain |dB| error over the bins within ``floor_rel_db`` of the frame maximum (reporting)"""
    db_gpu = np.asarray(db_gpu, dtype=np.float64)
    db_gold = np.asarray(db_gold, dtype=np.float64)
    mask = db_gold >= (db_gold.max(axis=-1, keepdims=True) - floor_rel_db)
    return float(np.where(mask, np.abs(db_gpu - db_gold), 0.0).max())
