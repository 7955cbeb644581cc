"""Pin the CPU oracle against vectors captured from the imported reference (tests/golden/*.npz).

precision="ref" restatements must be bit-identical to what the reference returned; the float64
"gold" variants (what the GPU is compared against) must agree with the reference within the
float32 noise of the reference itself.
"""
import os

import numpy as np
import pytest

from oracle import spectrum_oracle as so


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _frames(g):
    x = so.unpack_iq_int8(g["iq_i8"])
    n, hop, nf = int(g["nfft"]), int(g["hop"]), int(g["n_frames"])
    assert so.num_frames(len(x), n, hop) == nf
    return [so.frame(x, n, hop, k) for k in range(nf)], n


HACKRF_MODES = {
    "plain": dict(),
    "psd": dict(use_psd=True),
    "exp4": dict(avg=("exp", 4)),
    "lin3": dict(avg=("lin", 3)),
    "psd_exp2": dict(use_psd=True, avg=("exp", 2)),
    "dc_alpha_0p25": dict(dc_alpha=0.25),
}


@pytest.mark.parametrize("nfft", [1000, 1024, 4096, 16384])
@pytest.mark.parametrize("mode", sorted(HACKRF_MODES))
def test_hackrf_branch_bit_identical(golden_dir, nfft, mode):
    g = _load(golden_dir, f"hackrf_{nfft}.npz")
    frames, n = _frames(g)
    kw = dict(HACKRF_MODES[mode])
    avg = kw.pop("avg", ("off", 1))
    br = so.HackrfBranchOracle(n, float(g["sample_rate"]), precision="ref", **kw)
    br.averager.set_mode(*avg)
    assert np.array_equal(br.window, g["window"])
    for k, fr in enumerate(frames):
        out = np.asarray(br.power_levels(fr))
        assert out.dtype == g[mode].dtype
        assert np.array_equal(out, g[mode][k]), f"{mode} frame {k}"
    assert np.array_equal(so.shifted_freq_bins(n, float(g["sample_rate"]), float(g["centre_freq"])),
                          g["freq_bins"])


@pytest.mark.parametrize("nfft", [1024, 4096, 16384])
def test_hackrf_gold_close_to_reference(golden_dir, nfft):
    g = _load(golden_dir, f"hackrf_{nfft}.npz")
    frames, n = _frames(g)
    br = so.HackrfBranchOracle(n, float(g["sample_rate"]), precision="gold")
    gold = np.stack([br.power_levels(fr) for fr in frames])
    rel, ddb = so.parity_metrics(g["plain"], gold)
    assert rel < 1e-5 and ddb < 1e-3, (rel, ddb)


RTL_MODES = {
    "hanning": dict(),
    "hamming": dict(window="hamming"),
    "rectangle": dict(window="rectangle"),
    "psd": dict(use_psd=True),
    "lin3": dict(avg=("lin", 3)),
    "exp4": dict(avg=("exp", 4)),
}


@pytest.mark.parametrize("nfft", [1024, 1500, 4096])
@pytest.mark.parametrize("mode", sorted(RTL_MODES))
def test_rtl_branch_bit_identical(golden_dir, nfft, mode):
    g = _load(golden_dir, f"rtl_{nfft}.npz")
    frames, n = _frames(g)
    kw = dict(RTL_MODES[mode])
    avg = kw.pop("avg", ("off", 1))
    br = so.RtlBranchOracle(n, float(g["sample_rate"]), precision="ref", **kw)
    br.averager.set_mode(*avg)
    for k, fr in enumerate(frames):
        out = np.asarray(br.power_levels(fr))
        assert np.array_equal(out, g[mode][k]), f"{mode} frame {k}"
    assert np.array_equal(so.shifted_freq_bins(n, float(g["sample_rate"]), float(g["centre_freq"])),
                          g["freq_bins"])


@pytest.mark.parametrize("mode,chan,psd", [("mono", "mono", False), ("left", "left", False),
                                           ("mono_psd", "mono", True)])
def test_audio_branch_bit_identical(golden_dir, mode, chan, psd):
    g = _load(golden_dir, "audio_1024.npz")
    n, nf = int(g["nfft"]), int(g["n_frames"])
    st = g["stereo_f32"]
    win = so.rtl_window("hanning", n)
    for k in range(nf):
        blk = st[k * n:(k + 1) * n]
        left, right = blk[:, 0], blk[:, 1]
        sig = (left + right) * 0.5 if chan == "mono" else left
        p = so.audio_compute_power(sig, win, n, int(g["sample_rate"]), psd)
        assert np.array_equal(so.audio_db(p, psd), g[mode][k])
    assert np.array_equal(so.audio_freq_bins(n, int(g["sample_rate"])), g["freq_bins"])


@pytest.mark.parametrize("name,mode,n", [("off", "off", 1), ("exp8", "exp", 8), ("lin4", "lin", 4),
                                         ("lin64", "lin", 64), ("exp1", "exp", 1)])
def test_trace_averager_bit_identical(golden_dir, name, mode, n):
    g = _load(golden_dir, "averager.npz")
    av = so.TraceAveragerOracle()
    av.set_mode(mode, n)
    for k, f in enumerate(g["frames"]):
        out = np.array(av.process(f), copy=True)
        assert out.dtype == g[name].dtype
        assert np.array_equal(out, g[name][k])


def test_trace_averager_reference_properties():
    """The three things the reference's own test_smoke.py:137-175 asserts."""
    av = so.TraceAveragerOracle()
    x = np.ones(8, dtype=np.float32)
    assert av.process(x) is x                       # passthrough when off
    av.set_mode("exp", 4)
    first = av.process(x * 2)
    assert np.allclose(first, 2.0)                  # first frame == input
    moved = np.array(av.process(x * 6), copy=True)
    assert np.all(moved > 2.0) and np.all(moved < 6.0)   # moves toward new input
    av.reset()
    assert av.buffer is None and av.count == 0


def _processor_run(g, hold):
    n = int(g["nfft"])
    br = so.HackrfBranchOracle(n, float(g["sample_rate"]), precision="ref")
    tare = so.TareOracle()
    lives, mx, mn = [], [], []
    for k, fr in enumerate(g["frames_c64"]):
        if k == int(g["tare_start"]):
            tare.start_collecting()
        db = so.apply_cal_offset(np.asarray(br.power_levels(fr)), float(g["cal_offset"]))
        db = tare.apply(db)
        hold.update(db)
        lives.append(np.array(db, copy=True))
        mx.append(None if hold.max is None else np.array(hold.max, copy=True))
        mn.append(None if hold.min is None else np.array(hold.min, copy=True))
    return lives, mx, mn, tare


def test_processor_sequence_bit_identical(golden_dir):
    g = _load(golden_dir, "processor_1024.npz")
    lives, mx, _, tare = _processor_run(g, so.HoldOracle(True, False))
    _, _, mn, _ = _processor_run(g, so.HoldOracle(False, True))
    for k in range(int(g["n_frames"])):
        assert np.array_equal(lives[k], g["live"][k]), k
        assert np.array_equal(mx[k], g["max_hold"][k]), k
        assert np.array_equal(mn[k], g["min_hold"][k]), k
    assert tare.active == bool(g["tare_active_at_end"])
    assert np.array_equal(tare.baseline, g["baseline"])


def test_processor_hold_alias_quirk_pinned(golden_dir):
    """Both holds enabled from frame 0: the reference aliases the two buffers (quirk ii)."""
    g = _load(golden_dir, "processor_1024.npz")
    _, mx, mn, _ = _processor_run(g, so.HoldOracle(True, True, alias_quirk=True))
    for k in range(int(g["n_frames"])):
        assert np.array_equal(mx[k], g["max_hold_both"][k]), k
        assert np.array_equal(mn[k], g["min_hold_both"][k]), k
    # and the quirk is real: from frame 1 on both traces just equal the live frame
    assert np.array_equal(g["max_hold_both"][5], g["live"][5])
    assert np.array_equal(g["min_hold_both"][5], g["live"][5])
    # the intended (independent) traces differ from that
    assert not np.array_equal(g["max_hold"][5], g["live"][5])


def test_nan_safe_identity_on_clean():
    """test_smoke.py:262-272: clean arrays are returned without copy."""
    a = np.arange(4.0)
    assert so.nan_safe(a, -500.0) is a
    b = np.array([1.0, np.nan])
    out = so.nan_safe(b, -500.0)
    assert out is not b and out[1] == -500.0


def test_rbw_examples():
    """test_rbw_calculation.py:54,65,76 - RBW = fs/N."""
    for fs, n, rbw in ((20e6, 1024, 19531.25), (2e6, 1024, 1953.125), (44100, 1024, 43.06640625)):
        fb = so.shifted_freq_bins(n, fs, 0.0)
        assert abs((fb[1] - fb[0]) - rbw) < 1e-9


def test_c5_million_point_fixture(golden_dir):
    """Long-frame (C5) vectors from the imported reference (tests/golden/make_golden_c5.py): RTL branch at 2^20
    points, TraceAverager lin over 8 segments.  The restatement reproduces the comb of the first frame and of
    the Welch mean bit for bit (input regenerated from the stored seed)."""
    g = _load(golden_dir, "c5_million.npz")
    n, k = int(g["nfft"]), int(g["k"])
    x = so.unpack_iq_int8(so.synth_iq_int8(n * k, n, seed=int(g["seed"])))
    br = so.RtlBranchOracle(n, float(g["sample_rate"]), precision="ref")
    br.averager.set_mode("lin", k)
    for seg in range(k):
        p = np.asarray(br.power_levels(x[seg * n:(seg + 1) * n]), dtype=np.float64)
        tag = {0: "first", k - 1: "mean"}.get(seg)
        if tag:
            assert np.array_equal(p[g[f"{tag}_comb_bins"]], g[f"{tag}_comb_db"]), tag
            assert np.array_equal(p[g[f"{tag}_top_bins"]], g[f"{tag}_top_db"]), tag
            assert np.array_equal(np.sort(np.argsort(p)[-64:]), g[f"{tag}_top_bins"])
            assert p.sum() == float(g[f"{tag}_sum_db"])
    fb = so.shifted_freq_bins(n, float(g["sample_rate"]), float(g["centre_freq"]))
    assert fb[0] == float(g["freq_first"]) and fb[-1] == float(g["freq_last"])


def test_hold_allowance_follows_from_the_rows():
    """oracle.HoldAllowance: |max_f a_f - max_f b_f| <= max_f |a_f - b_f| (the same for min) bin by bin - rows perturbed by
    anything up to their own allowance give hold traces within the inherited allowance, whatever the perturbation's sign
    pattern (the extreme of many draws included); a trace pushed beyond it is caught; bins no row had checked stay unchecked."""
    rng = np.random.default_rng(7)
    for trial in range(40):
        nf, n = int(rng.integers(1, 60)), int(rng.choice([64, 257, 1024]))
        gold = -60.0 + 10 * np.log10(rng.exponential(1.0, size=(nf, n)) + 1e-9)
        gold[:, n // 3] = 40.0 + rng.normal(0, 3, nf)                       # a tone: the other bins sit ~100 dB down
        units = float(rng.choice([1.0, 2.0, 3.0]))
        allow = so.row_allowance_db(gold, 100.0, units * so.AMP_FLOOR)
        sign = rng.choice([-1.0, 1.0], size=gold.shape) if trial % 3 else -np.ones_like(gold)
        rows = gold + sign * np.where(np.isfinite(allow), allow, 0.0) * rng.uniform(0.0, 1.0, gold.shape)
        rel_r, ddb_r = so.parity_metrics(rows, gold, amp_floor=units * so.AMP_FLOOR)
        assert ddb_r <= 1e-3 * (1 + 1e-9)
        ha = so.HoldAllowance(amp_floor=units * so.AMP_FLOOR).update(gold[: nf // 2 + 1]).update(gold[nf // 2 + 1:] if nf // 2 + 1 < nf else gold[:1])
        for fn in (np.max, np.min):
            rel, ddb = ha.metrics(fn(rows, axis=0), fn(gold, axis=0))
            assert ddb <= 1e-3 * (1 + 1e-9), (trial, fn.__name__, ddb)
        # the old rule - the trace judged like a single row of its own maximum - rejects some of these min traces
        worse = fn(rows, axis=0).copy()
        k = int(np.argmax(np.isfinite(ha.allow)))
        worse[k] += 3.0 * ha.allow[k]
        assert ha.metrics(worse, fn(gold, axis=0))[1] > 1e-3
