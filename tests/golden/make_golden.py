#!/usr/bin/env python3
"""Generate golden input/output vectors by running the *imported reference itself*.

Run in the build container only (needs /root/reference; the GPU box never sees it):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Hardware modules (hackrf / rtlsdr / sounddevice) are replaced by MagicMock exactly as the
reference's own test_smoke.py:26-37 does, but the real numpy/scipy are kept so the FFT runs.
The fixtures hold DATA only: seeded synthetic IQ in, the arrays the reference returned out.

Files written next to this script (np.savez_compressed):
  hackrf_<N>.npz    HackrfSamplesDataSource.get_power_levels  (plain / psd / exp / lin / dc_alpha)
  rtl_<N>.npz       RtlSamplesDataSource.get_power_levels     (hanning / hamming / rectangle / psd / lin)
  audio_1024.npz    MicrophoneSamplesDataSource.get_power_levels (mono / left / psd)
  averager.npz      TraceAverager.process sequences (off / exp / lin with cap)
  processor_1024.npz  DataProcessor._process_sample_data sequence (cal offset, tare, max/min hold)
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

from datasources.hackrf_samples import HackrfSamplesDataSource  # noqa: E402
from datasources.rtl_samples import RtlSamplesDataSource  # noqa: E402
from datasources.audio_samples import MicrophoneSamplesDataSource  # noqa: E402
from utils.signal_processing import TraceAverager  # noqa: E402
from core.display_data_processor import DataProcessor  # noqa: E402
from core.tare_state import TareState  # noqa: E402

from oracle.spectrum_oracle import synth_iq_int8, unpack_iq_int8  # noqa: E402  (input generator only)

FS_HACKRF = 20_000_000
FC_HACKRF = 2_450_000_000
FS_RTL = 2_000_000
FC_RTL = 100_300_000


def run_hackrf(frames_c64, nfft, *, psd=False, avg=("off", 1), dc_alpha=1.0):
    src = HackrfSamplesDataSource(sample_rate=FS_HACKRF, centre_freq=FC_HACKRF)
    src.num_samples = nfft
    src.running = True
    src._allocate_fft_resources()
    src.set_psd_mode(psd)
    src.set_averaging(*avg)
    src.set_dc_alpha(dc_alpha)
    outs = []
    for fr in frames_c64:
        src._reservoir = np.array(fr, dtype=np.complex64, copy=True)
        p, fb = src.get_power_levels()
        outs.append(np.array(p, copy=True))
    return np.stack(outs), np.array(fb, copy=True), np.array(src._window, copy=True)


class _FakeSdr:
    def __init__(self, frames, fs, fc):
        self._frames = list(frames)
        self._i = 0
        self._fs, self._fc = fs, fc

    def read_samples(self, n):
        fr = self._frames[self._i]
        self._i += 1
        assert len(fr) == n
        return np.array(fr, copy=True)

    def get_sample_rate(self):
        return self._fs

    def get_center_freq(self):
        return self._fc


def run_rtl(frames, nfft, *, window="hanning", psd=False, avg=("off", 1)):
    src = RtlSamplesDataSource(sample_rate=FS_RTL, centre_freq=FC_RTL)
    src.set_fft_size(nfft)
    src.set_window_type(window)
    src.set_psd_mode(psd)
    src.set_averaging(*avg)
    src.sdr = _FakeSdr(frames, float(FS_RTL), float(FC_RTL))
    src.running = True
    outs = []
    for _ in frames:
        p, fb = src.get_power_levels()
        outs.append(np.array(p, copy=True))
    return np.stack(outs), np.array(fb, copy=True)


class _FakeStream:
    def __init__(self, blocks):
        self._blocks = list(blocks)
        self._i = 0

    def read(self, n):
        b = self._blocks[self._i]
        self._i += 1
        assert len(b) == n
        return np.array(b, copy=True), False


def run_audio(blocks, nfft, *, mode="mono", psd=False):
    src = MicrophoneSamplesDataSource(sample_rate=44100, centre_freq=0)
    src.set_fft_size(nfft)
    src.set_channel_mode(mode)
    src.set_psd_mode(psd)
    src.stream = _FakeStream(blocks)
    src.running = True
    src._audio_block = nfft
    outs = []
    for _ in blocks:
        p, fb = src.get_power_levels()
        outs.append(np.array(p, copy=True))
    return np.stack(outs), np.array(fb, copy=True)


def frames_of(iq_i8, nfft, hop, nf):
    x = unpack_iq_int8(iq_i8)
    return [x[k * hop:k * hop + nfft] for k in range(nf)]


def gen_hackrf(nfft, nf, hop, seed):
    iq = synth_iq_int8(hop * (nf - 1) + nfft, nfft, seed)
    fr = frames_of(iq, nfft, hop, nf)
    d = {"iq_i8": iq, "nfft": nfft, "hop": hop, "n_frames": nf,
         "sample_rate": FS_HACKRF, "centre_freq": FC_HACKRF}
    d["plain"], d["freq_bins"], d["window"] = run_hackrf(fr, nfft)
    d["psd"], _, _ = run_hackrf(fr, nfft, psd=True)
    d["exp4"], _, _ = run_hackrf(fr, nfft, avg=("exp", 4))
    d["lin3"], _, _ = run_hackrf(fr, nfft, avg=("lin", 3))
    d["psd_exp2"], _, _ = run_hackrf(fr, nfft, psd=True, avg=("exp", 2))
    d["dc_alpha_0p25"], _, _ = run_hackrf(fr, nfft, dc_alpha=0.25)
    np.savez_compressed(os.path.join(HERE, f"hackrf_{nfft}.npz"), **d)


def gen_rtl(nfft, nf, seed):
    iq = synth_iq_int8(nfft * nf, nfft, seed)
    fr = frames_of(iq, nfft, nfft, nf)
    d = {"iq_i8": iq, "nfft": nfft, "hop": nfft, "n_frames": nf,
         "sample_rate": FS_RTL, "centre_freq": FC_RTL}
    d["hanning"], d["freq_bins"] = run_rtl(fr, nfft)
    d["hamming"], _ = run_rtl(fr, nfft, window="hamming")
    d["rectangle"], _ = run_rtl(fr, nfft, window="rectangle")
    d["psd"], _ = run_rtl(fr, nfft, psd=True)
    d["lin3"], _ = run_rtl(fr, nfft, avg=("lin", 3))
    d["exp4"], _ = run_rtl(fr, nfft, avg=("exp", 4))
    np.savez_compressed(os.path.join(HERE, f"rtl_{nfft}.npz"), **d)


def gen_audio(nfft, nf, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(nfft * nf) / 44100.0
    left = 0.3 * np.sin(2 * np.pi * 1000.0 * t) + 0.01 * rng.standard_normal(len(t)) + 0.05
    right = 0.2 * np.sin(2 * np.pi * 3300.0 * t) + 0.01 * rng.standard_normal(len(t)) - 0.02
    st = np.stack([left, right], axis=1).astype(np.float32)
    blocks = [st[k * nfft:(k + 1) * nfft] for k in range(nf)]
    d = {"stereo_f32": st, "nfft": nfft, "n_frames": nf, "sample_rate": 44100}
    d["mono"], d["freq_bins"] = run_audio(blocks, nfft)
    d["left"], _ = run_audio(blocks, nfft, mode="left")
    d["mono_psd"], _ = run_audio(blocks, nfft, psd=True)
    np.savez_compressed(os.path.join(HERE, f"audio_{nfft}.npz"), **d)


def gen_averager(seed):
    rng = np.random.default_rng(seed)
    frames = (rng.random((10, 256)) * 3.0).astype(np.float32)
    d = {"frames": frames}
    for name, (mode, n) in {"off": ("off", 1), "exp8": ("exp", 8), "lin4": ("lin", 4),
                            "lin64": ("lin", 64), "exp1": ("exp", 1)}.items():
        av = TraceAverager()
        av.set_mode(mode, n)
        d[name] = np.stack([np.array(av.process(f), copy=True) for f in frames])
    np.savez_compressed(os.path.join(HERE, "averager.npz"), **d)


class _Label:
    def setText(self, s):
        self.text = s


class _Cal:
    def __init__(self, off):
        self.off = off

    def get_offset(self, source_type):
        return self.off


class _NS:
    pass


def gen_processor(nfft, nf, seed, cal_offset=-0.8087054556396822, tare_start=3):
    """Drive DataProcessor._process_sample_data with stub mw/dm objects (as test_smoke.py does).

    Three runs over the same frames: max hold only, min hold only, and both enabled from the
    first frame.  The last one exposes the reference's aliasing quirk (SURVEY.md 8(a) quirk ii):
    _nan_safe returns its argument uncopied, so max_power_levels and min_power_levels become the
    SAME ndarray and every later frame leaves fmin(fmax(h, x), x) == x in both.
    """
    iq = synth_iq_int8(nfft * nf, nfft, seed)
    fr = frames_of(iq, nfft, nfft, nf)
    # per-frame amplitude wobble so that hold traces actually move
    rng = np.random.default_rng(seed + 1)
    gains = (0.5 + rng.random(nf)).astype(np.float32)
    fr = [(f * g).astype(np.complex64) for f, g in zip(fr, gains)]

    def run(max_on, min_on):
        src = HackrfSamplesDataSource(sample_rate=FS_HACKRF, centre_freq=FC_HACKRF)
        src.num_samples = nfft
        src.running = True
        src._allocate_fft_resources()
        mw, dm = _NS(), _NS()
        mw.current_source = src
        mw.calibration_manager = _Cal(cal_offset)
        mw.source_manager = _NS()
        mw.source_manager.last_source_type = "hackrf_samples"
        mw.status_label = _Label()
        mw.tare_active = False
        mw.baseline_power_levels = None
        mw.live_power_levels = None
        mw.max_power_levels = None
        mw.min_power_levels = None
        mw.min_hold_enabled = min_on
        mw.frequency_bins = None
        dm.tare_state = TareState()
        dm.max_peak_search_enabled = max_on
        dm.duty_cycle_enabled = False
        dm.peak_list_enabled = False
        dm._update_tare_button_label = lambda s: None

        def _clear_tare():
            mw.tare_active = False
            mw.baseline_power_levels = None
        dm._clear_tare = _clear_tare

        dp = DataProcessor(mw, dm)
        live, mx, mn = [], [], []
        for k, f in enumerate(fr):
            if k == tare_start:
                dm.tare_state = TareState(collecting=True)
            src._reservoir = np.array(f, copy=True)
            dp._process_sample_data()
            live.append(np.array(mw.live_power_levels, copy=True))
            if max_on:
                mx.append(np.array(mw.max_power_levels, copy=True))
            if min_on:
                mn.append(np.array(mw.min_power_levels, copy=True))
        return (np.stack(live), np.stack(mx) if mx else None, np.stack(mn) if mn else None,
                np.array(mw.baseline_power_levels, copy=True), bool(mw.tare_active))

    live, mx, _, baseline, active = run(True, False)
    live2, _, mn, _, _ = run(False, True)
    live3, mx_b, mn_b, _, _ = run(True, True)
    assert np.array_equal(live, live2) and np.array_equal(live, live3)
    d = {"frames_c64": np.stack(fr), "nfft": nfft, "n_frames": nf, "cal_offset": cal_offset,
         "tare_start": tare_start, "sample_rate": FS_HACKRF, "centre_freq": FC_HACKRF,
         "live": live, "max_hold": mx, "min_hold": mn,
         "max_hold_both": mx_b, "min_hold_both": mn_b,
         "baseline": baseline, "tare_active_at_end": active}
    np.savez_compressed(os.path.join(HERE, f"processor_{nfft}.npz"), **d)


def main():
    gen_hackrf(1024, 6, 1024, seed=1)
    gen_hackrf(4096, 4, 2048, seed=2)
    gen_hackrf(16384, 3, 8192, seed=3)
    gen_hackrf(1000, 5, 500, seed=4)          # sizes that are not a power of two: np.fft.fft takes any N
    gen_rtl(1024, 5, seed=11)
    gen_rtl(4096, 4, seed=12)
    gen_rtl(1500, 4, seed=13)
    gen_audio(1024, 4, seed=21)
    gen_averager(seed=31)
    gen_processor(1024, 40, seed=41)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
    assert not any(d == "__pycache__" for _, ds, _ in os.walk(REF) for d in ds), \
        "bytecode was written into the reference tree"


if __name__ == "__main__":
    main()
