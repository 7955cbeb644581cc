"""GPU parity of the trace analytics / display accumulators against
oracle/analytics_oracle.py and the vectors captured from the imported reference."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import analytics_oracle as ao
from oracle import spectrum_oracle as so

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import topdogspectrumanalyser_amd as p
    return p


@pytest.fixture(scope="module")
def an():
    from topdogspectrumanalyser_amd import analytics
    return analytics


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "analytics.npz"))


class DevRows:
    """rows uploaded to the device for the duration of a with-block"""

    def __init__(self, pkg, rows):
        self.nat = pkg._native
        self.rows = np.ascontiguousarray(rows, dtype=np.float32)
        self.ptr = C.c_void_p()

    def __enter__(self):
        nat = self.nat
        nat.check(nat.lib.tdsa_dev_alloc(0, self.rows.nbytes, C.byref(self.ptr)))
        nat.check(nat.lib.tdsa_memcpy_h2d(0, self.ptr, self.rows.ctypes.data_as(C.c_void_p), self.rows.nbytes))
        return self.ptr.value

    def __exit__(self, *exc):
        self.nat.lib.tdsa_dev_free(0, self.ptr)


def test_top_peaks_match_reference_vectors(pkg, an, gold):
    for key in gold["peak_cases"]:
        key = str(key)
        n, kind, exc = key.split("_")[1:]
        n = int(n)
        tr = gold[key + "_trace"]
        with pkg.SpectrumEngine(max(n, 64), max_frames=1) as e, DevRows(pkg, tr[None, :]) as d:
            bins, db = an.rows_top_peaks(e, d, 1, n_bins=n, n=5, min_excursion_db=float(exc))
        want = list(gold[key + "_bins"])
        got = [int(b) for b in bins[0] if b >= 0]
        assert got == want, key
        assert np.array_equal(db[0][: len(want)].astype(np.float64), gold[key + "_pwr"]), key
        assert np.all(bins[0][len(want):] == -1) and np.all(np.isnan(db[0][len(want):]))


def test_top_peaks_batch_on_spectra(pkg, an):
    """Peak lists of real GPU spectra (C3 shape, 64 frames) equal the restated reference row by row."""
    nfft, hop, nf = 16384, 8192, 64
    iq = so.synth_iq_int8(hop * (nf - 1) + nfft, nfft, seed=77)
    fb = so.shifted_freq_bins(nfft, 20e6, 2.45e9)
    with pkg.SpectrumEngine(nfft, max_frames=nf) as e:
        e.set_window(so.hackrf_window(nfft))
        e.configure(db_mode="mag", log_floor=so.LOG_FLOOR, dc_alpha=1.0)
        rows = e.process(iq, hop=hop)
        with DevRows(pkg, rows) as d:
            for exc in (6.0, 10.0):
                bins, db = an.rows_top_peaks(e, d, nf, min_excursion_db=exc)
                for r in range(nf):
                    want = ao.find_top_peaks(fb, rows[r], 5, *ao.peak_list_params(nfft, exc))
                    got = an.peaks_as_reference(fb, bins[r], db[r])
                    assert got == want, (r, exc)


def test_rows_stats_peak_argmax_band(pkg, an, gold):
    bins, tr = gold["band_bins"], gold["band_trace"]
    rng = np.random.default_rng(5)
    rows = np.stack([tr, tr[::-1].copy(), np.roll(tr, 100), rng.normal(-80, 5, tr.size).astype(np.float32)])
    rows[1, 7] = rows[1, 900] = rows[1].max() + 1.0          # equal maxima: first index wins
    with pkg.SpectrumEngine(4096, max_frames=4) as e, DevRows(pkg, rows) as d:
        for (a, b), want in zip(gold["band_edges"], gold["band_db"]):
            peak, pbin, bdb = an.rows_stats(e, d, 4, freq_bins=bins, band=(a, b))
            for r in range(4):
                assert peak[r] == rows[r].max() and pbin[r] == int(np.argmax(rows[r]))
                ref = ao.band_power_db(bins, rows[r], a, b)
                if ref is None:
                    assert np.isnan(bdb[r])
                else:
                    assert abs(bdb[r] - ref) <= 1e-4, (r, a, b)
            if not np.isnan(want):
                assert abs(bdb[0] - want) <= 1e-4                  # the value the reference itself returned
        peak, pbin, bdb = an.rows_stats(e, d, 4)
        assert bdb is None and pbin[1] == 7
    nanrow = tr.copy()
    nanrow[[300, 20]] = np.nan
    with pkg.SpectrumEngine(4096, max_frames=1) as e, DevRows(pkg, nanrow[None, :]) as d:
        peak, pbin, _ = an.rows_stats(e, d, 1)
        assert np.isnan(peak[0]) and pbin[0] == 20 == int(np.argmax(nanrow))    # numpy: first NaN


def test_duty_cycle_from_device_rows(pkg, an, gold):
    frames = gold["duty_frames"]
    d = an.DutyCycle()
    with pkg.SpectrumEngine(1024, max_frames=130) as e:
        with DevRows(pkg, frames[:130]) as a:
            d.update_from_rows(e, a, 130, threshold_dbm=-60.0)
        assert d.duty_pct == gold["duty_pct"][129]
        with DevRows(pkg, frames[130:]) as b:
            d.update_from_rows(e, b, 130, threshold_dbm=-45.0)
    ref = ao.DutyCycleOracle()
    for i, fr in enumerate(frames):
        ref.update_from_power(fr, threshold_dbm=-60.0 if i < 130 else -45.0)
    assert d.duty_pct == ref.duty_pct == gold["duty_pct"][-1]
    assert d.on_power_dbm == ref.on_power_dbm and d.off_power_dbm == ref.off_power_dbm
    h = an.DutyCycle()                                       # host-array entry point, reference sequence
    for i, fr in enumerate(frames[:40]):
        h.update_from_power(fr, threshold_dbm=-60.0)
        assert h.duty_pct == gold["duty_pct"][i]


@pytest.mark.parametrize("decay", [0.96, 0.5, 1.0])
def test_density_histogram_matches_restatement(pkg, an, decay):
    n, nf = 1000, 37                                         # n not a multiple of the 16-bin tile
    rng = np.random.default_rng(int(decay * 100))
    rows = rng.normal(-70, 25, size=(nf, n)).astype(np.float32)
    rows[3, 5] = np.nan
    rows[4, 6] = 100.0
    rows[5, 7] = -200.3                                      # truncation toward zero lands in bin 0
    rows[6, 8] = -201.0
    rows[7, 9] = 99.99
    ref = ao.DensityOracle(decay)
    for r in rows:
        ref.update(r)
    with an.DensityHistogram(n, decay) as dh, DevRows(pkg, rows) as d:
        dh.update_rows(None, d, nf)
        h = dh.hist()
        assert h.shape == (n, ao.AMP_BINS)
        assert np.array_equal(h, ref.hist)                   # same float32 sequence: bit exact
        assert np.allclose(dh.image(), np.log1p(ref.hist), rtol=2e-6, atol=1e-7)
        one = an.DensityHistogram(n, decay)
        for r in rows[:5]:
            one.update(r)                                    # per-tick host entry point
        ref5 = ao.DensityOracle(decay)
        for r in rows[:5]:
            ref5.update(r)
        assert np.array_equal(one.hist(), ref5.hist)
        one.reset()
        assert one.hist().sum() == 0
        with pytest.raises(Exception):
            one.update(rows[0][:10])
        one.close()


@pytest.mark.filterwarnings("ignore:invalid value encountered in cast")     # the restated astype(int32) of +-inf
@pytest.mark.parametrize("seed", range(int(os.environ.get("TDSA_DENSITY_CASES", "8"))))
def test_density_histogram_random(pkg, an, seed):
    """Seeded random row sets (sizes off the 16-bin tile, row counts off the 32-row chunk, clustered and spread
    amplitudes, NaN / inf / edge values, several calls in a row): bit for bit the restated float32 sequence."""
    rng = np.random.default_rng(500 + seed)
    n = int(rng.choice([16, 17, 100, 1000, 1024, 4096, 5000]))
    decay = float(rng.choice([0.96, 0.5, 1.0, 0.999]))
    ref = ao.DensityOracle(decay)
    with an.DensityHistogram(n, decay) as dh:
        for _call in range(int(rng.integers(1, 4))):
            nf = int(rng.integers(1, 100))
            centre, spread = rng.uniform(-150, 50), float(rng.choice([0.2, 3.0, 25.0, 120.0]))
            rows = rng.normal(centre, spread, size=(nf, n)).astype(np.float32)
            for _ in range(int(rng.integers(0, 6))):
                rows[rng.integers(0, nf), rng.integers(0, n)] = rng.choice(
                    np.array([np.nan, np.inf, -np.inf, -200.0, 100.0, 99.999, -200.3, -201.0], dtype=np.float32))
            if rng.integers(0, 2):
                rows = np.round(rows * 512 / 300) * np.float32(300 / 512) - np.float32(200.0) + np.float32(200.0)   # near bin edges
            for r in rows:
                ref.update(r)
            with DevRows(pkg, rows) as d:
                dh.update_rows(None, d, nf)
            assert np.array_equal(dh.hist(), ref.hist), (seed, n, decay, nf)


def test_density_from_engine_rows(pkg, an):
    nfft, nf = 2048, 50
    iq = so.synth_iq_int8(nfft * nf, nfft, seed=9)
    nat = pkg._native
    with pkg.SpectrumEngine(nfft, max_frames=nf) as e, an.DensityHistogram(nfft, 0.9) as dh:
        e.set_window(so.hackrf_window(nfft))
        e.configure(db_mode="mag", log_floor=so.LOG_FLOOR, dc_alpha=1.0)
        rows = e.process(iq, hop=nfft)
        d_in, d_out = C.c_void_p(), C.c_void_p()
        nat.check(nat.lib.tdsa_dev_alloc(0, iq.nbytes, C.byref(d_in)))
        nat.check(nat.lib.tdsa_dev_alloc(0, rows.nbytes, C.byref(d_out)))
        nat.check(nat.lib.tdsa_memcpy_h2d(0, d_in, iq.ctypes.data_as(C.c_void_p), iq.nbytes))
        e.set_overlap(2)
        e.process_device(nat.IN_I8, d_in.value, nfft * nf, nfft, nf, d_out.value)
        dh.update_rows(e, d_out.value, nf)                   # ordered after the producer by the library
        ref = ao.DensityOracle(0.9)
        for r in rows:
            ref.update(r)
        assert np.array_equal(dh.hist(), ref.hist)
        nat.lib.tdsa_dev_free(0, d_in)
        nat.lib.tdsa_dev_free(0, d_out)


def test_waterfall_ring_matches_restatement(pkg, an):
    H, W = 7, 300
    rng = np.random.default_rng(3)
    base = rng.normal(-90, 3, size=(30, W)).astype(np.float32)
    seq = [base[0], base[0], base[1], base[2], base[2], base[2], base[3]] + [base[i] for i in range(4, 20)]
    nanrow = base[20].copy()
    nanrow[5] = np.nan
    seq += [nanrow, nanrow, base[21]]                        # NaN rows never compare equal
    ref = ao.WaterfallOracle(H, W, -120.0)
    with an.WaterfallRing(H, W, -120.0) as wf:
        assert np.all(wf.view() == -120.0)
        for row in seq[:9]:                                  # per-tick host entry point
            assert wf.push(row) == ref.update(row)
            assert wf.ptr == ref.ptr
        assert np.array_equal(wf.view(), ref.view(), equal_nan=True)
        rest = np.stack(seq[9:])
        n_ref = sum(ref.update(r) for r in rest)
        with DevRows(pkg, rest) as d:
            assert wf.push_rows(None, d, len(rest)) == n_ref      # batch larger than the history
        assert wf.ptr == ref.ptr
        assert np.array_equal(wf.view(), ref.view(), equal_nan=True)
        with pytest.raises(Exception):
            wf.push(base[0][:10])


@pytest.mark.parametrize("seed", range(int(os.environ.get("TDSA_WATERFALL_CASES", "8"))))
def test_waterfall_ring_random(pkg, an, seed):
    """Seeded random update streams (repeated rows, NaN rows, batches shorter and longer than the history, host and
    device entry points mixed): pointer, number of new rows and the displayed view against the restatement."""
    rng = np.random.default_rng(700 + seed)
    H, W = int(rng.integers(1, 40)), int(rng.choice([16, 100, 1000, 1024]))
    pool = rng.normal(-90, 5, size=(12, W)).astype(np.float32)
    pool[3, rng.integers(0, W)] = np.nan
    ref = ao.WaterfallOracle(H, W, -120.0)
    with an.WaterfallRing(H, W, -120.0) as wf:
        for _call in range(int(rng.integers(2, 12))):
            k = int(rng.integers(1, 3 * H + 4))
            picks = rng.integers(0, len(pool), k)
            picks = np.repeat(picks, rng.integers(1, 3, k))[:k]          # runs of identical rows
            rows = np.ascontiguousarray(pool[picks])
            if rng.integers(0, 3) == 0:
                for r in rows:
                    assert wf.push(r) == ref.update(r)
            else:
                want = sum(ref.update(r) for r in rows)
                with DevRows(pkg, rows) as d:
                    assert wf.push_rows(None, d, len(rows)) == want
            assert wf.ptr == ref.ptr
            assert np.array_equal(wf.view(), ref.view(), equal_nan=True), (seed, H, W)


@pytest.mark.parametrize("H,W,k", [(50, 4096, 3000), (2000, 1001, 2500), (7, 2048, 1025), (600, 5000, 1500)])
def test_waterfall_ring_long_batches(pkg, an, H, W, k):
    """Batches of more than 1024 rows (the pointer walk is a scan with several rows per thread), shorter and longer than
    the history, rows that differ from their predecessor only beyond the first 1024 bins or in the last bin, row lengths
    that rule out 16-byte accesses: new-row count, pointer and view against the restatement, twice in a row."""
    rng = np.random.default_rng(H * 7 + W)
    pool = rng.normal(-90, 5, size=(9, W)).astype(np.float32)
    pool[1] = pool[0]
    pool[1, -1] += 1.0                                       # differs in the last bin only
    pool[2] = pool[0]
    pool[2, min(W - 1, 1500)] -= 2.0                         # differs beyond the first block of bins
    pool[4, W // 2] = np.nan
    ref = ao.WaterfallOracle(H, W, -120.0)
    with an.WaterfallRing(H, W, -120.0) as wf:
        for _call in range(2):
            picks = rng.integers(0, len(pool), k)
            picks = np.repeat(picks, rng.integers(1, 4, k))[:k]
            rows = np.ascontiguousarray(pool[picks])
            want = sum(ref.update(r) for r in rows)
            with DevRows(pkg, rows) as d:
                assert wf.push_rows(None, d, k) == want
            assert wf.ptr == ref.ptr
            assert np.array_equal(wf.view(), ref.view(), equal_nan=True)
            lo, hi = -110.0, -70.0
            assert np.array_equal(wf.view_u8(lo, hi), _levels_u8(ref.view(), lo, hi))


@pytest.mark.parametrize("seed", range(int(os.environ.get("TDSA_STATS_CASES", "8"))))
def test_rows_stats_random(pkg, an, seed):
    """Seeded random rows (ties at the maximum, NaNs, -inf, bands of every width incl. empty and out of range):
    peak value, argmax (numpy's first-index / first-NaN rules) and band power against numpy."""
    rng = np.random.default_rng(800 + seed)
    n = int(2 ** rng.integers(6, 15))
    nr = int(rng.integers(1, 20))
    rows = rng.normal(-80, 10, size=(nr, n)).astype(np.float32)
    if rng.integers(0, 2):
        rows = np.round(rows)                                   # ties
    for _ in range(int(rng.integers(0, 5))):
        rows[rng.integers(0, nr), rng.integers(0, n)] = rng.choice(np.array([np.nan, -np.inf, 30.0], dtype=np.float32))
    bins = np.linspace(100e6, 120e6, n)
    a, b = sorted(rng.uniform(95e6, 125e6, 2))
    if rng.integers(0, 6) == 0:
        a, b = 130e6, 140e6                                     # no bin inside
    with pkg.SpectrumEngine(max(n, 64), max_frames=1) as e, DevRows(pkg, rows) as d:
        peak, pbin, bdb = an.rows_stats(e, d, nr, n_bins=n, freq_bins=bins, band=(a, b))
    for r in range(nr):
        want_bin = int(np.argmax(rows[r]))
        assert pbin[r] == want_bin, (seed, r)
        assert (np.isnan(peak[r]) and np.isnan(rows[r][want_bin])) or peak[r] == rows[r][want_bin]
        ref = ao.band_power_db(bins, rows[r], a, b)
        if ref is None:
            assert np.isnan(bdb[r])
        else:
            assert (np.isnan(ref) and np.isnan(bdb[r])) or abs(bdb[r] - ref) <= 1e-4, (seed, r, bdb[r], ref)


@pytest.mark.parametrize("mode", ["medium", "fast", "off"])
def test_density_histogram_matches_reference_fixture(pkg, an, golden_dir, mode):
    """density_kernel against DensityDisplay._update_hist of the imported reference (displays.npz): NaN,
    -inf, out-of-range and on-the-edge values included, histogram bit for bit after rows 0, 7 and 47."""
    g = np.load(os.path.join(golden_dir, "displays.npz"))
    rows = np.ascontiguousarray(g["density_rows"])
    decay = float(g[f"density_{mode}_decay"])
    prev = 0
    with an.DensityHistogram(rows.shape[1], decay) as dh:
        for i, r in enumerate(g["density_snap_rows"]):
            chunk = np.ascontiguousarray(rows[prev:int(r) + 1])
            with DevRows(pkg, chunk) as d:
                dh.update_rows(None, d, len(chunk))         # batch entry point, state carried across calls
            assert np.array_equal(dh.hist(), g[f"density_{mode}_hist"][i]), (mode, int(r))
            prev = int(r) + 1
    with an.DensityHistogram(rows.shape[1], decay) as one:      # per-tick host entry point
        for row in rows[:8]:
            one.update(row)
        assert np.array_equal(one.hist(), g[f"density_{mode}_hist"][1])


def test_waterfall_ring_matches_reference_fixture(pkg, an, golden_dir):
    """waterfall ring against Waterfall.update_widget_data of the imported reference: pointer walk, row
    de-duplication and the displayed view at five points of a 40-update sequence."""
    g = np.load(os.path.join(golden_dir, "displays.npz"))
    H, W = int(g["wf_history_lines"]), g["wf_rows"].shape[1]
    steps = {int(s): i for i, s in enumerate(g["wf_view_steps"])}
    with an.WaterfallRing(H, W, float(g["wf_min_db"])) as wf:
        for step, idx in enumerate(g["wf_order"]):
            assert wf.push(g["wf_rows"][idx]) == bool(g["wf_added"][step])
            assert wf.ptr == int(g["wf_ptr"][step])
            if step in steps:
                assert np.array_equal(wf.view(), g["wf_views"][steps[step]])
    with an.WaterfallRing(H, W, float(g["wf_min_db"])) as wf:    # the same sequence as ONE device batch
        seq = np.ascontiguousarray(g["wf_rows"][g["wf_order"]])
        with DevRows(pkg, seq) as d:
            assert wf.push_rows(None, d, len(seq)) == int(g["wf_added"].sum())
        assert wf.ptr == int(g["wf_ptr"][-1])
        assert np.array_equal(wf.view(), g["wf_views"][-1])


def test_analytics_error_paths(pkg, an):
    with pkg.SpectrumEngine(1024, max_frames=1) as e, DevRows(pkg, np.zeros((1, 1024), np.float32)) as d:
        with pytest.raises(Exception):
            an.rows_top_peaks(e, d, 1, n=9)
        with pytest.raises(Exception):
            an.rows_top_peaks(e, d, 1, n_bins=32768)
        assert an.rows_top_peaks(e, d, 1)[0].tolist() == [[-1] * 5]          # flat row: no strict maximum
        peak, pbin, _ = an.rows_stats(e, d, 1)
        assert peak[0] == 0.0 and pbin[0] == 0
    with pytest.raises(Exception):
        an.DensityHistogram(0)
    with pytest.raises(Exception):
        an.WaterfallRing(0, 16, -100.0)


@pytest.mark.parametrize("seed", range(int(os.environ.get("TDSA_SWEEP_CASES", "24"))))
def test_top_peaks_random_traces(pkg, an, seed):
    """Seeded random traces (noise floors, tone combs, plateaus quantised to 0.5 dB so that equal values and
    near-threshold valleys occur) against the restated reference, all rows of a batch at once."""
    rng = np.random.default_rng(9000 + seed)
    n = int(2 ** rng.integers(6, 15))
    if seed % 4 == 3:
        n = int(rng.choice([3, 5, 33, 70, 100, 1000, 5000, 12345, 16383]))     # tiny rows, ragged last block of 32 bins
    rows = []
    for _ in range(12):
        k = np.arange(n)
        p = rng.exponential(1.0, size=n) * 10.0 ** rng.uniform(-12, -6)
        for _t in range(int(rng.integers(0, 7))):
            c, a, w = rng.integers(0, n), 10.0 ** rng.uniform(-9, -2), rng.uniform(0.6, 4.0)
            p += a * np.sinc((k - c) / w) ** 2
        tr = 10 * np.log10(p + 1e-15)
        if rng.integers(0, 3) == 0:
            tr = np.round(tr * 2) / 2                      # ties and exact-threshold excursions
        rows.append(tr.astype(np.float32))
    rows = np.stack(rows)
    exc = float(rng.choice([3.0, 6.0, 10.0]))
    sep = int(rng.choice([max(10, n // 50), 2, 5, 600, 31, 33]))  # (600: wider than the stride of a thread's bins)
    npk = int(rng.integers(1, 9))
    with pkg.SpectrumEngine(64 if n < 64 else 1 << (n - 1).bit_length(), max_frames=1) as e, DevRows(pkg, rows) as d:
        bins, db = an.rows_top_peaks(e, d, len(rows), n_bins=n, n=npk, min_sep_bins=sep, min_excursion_db=exc)
    for r, tr in enumerate(rows):
        want = ao.find_top_peak_bins(tr, npk, sep, exc)
        got = [int(b) for b in bins[r] if b >= 0]
        if got != want:
            # The reference visits EQUAL-valued candidates in the order np.argsort's unstable introsort
            # happens to leave them; the device visits the larger index first.  The two lists may part
            # ways only at such a tie (and are unrelated afterwards); anything else is a bug.
            j = next(i for i, (a, b) in enumerate(zip(got + [-1], want + [-1])) if a != b)
            assert j < len(got) and j < len(want) and tr[got[j]] == tr[want[j]], (seed, r, got, want)
        assert np.array_equal(db[r][: len(got)], tr[got])


# ---- marker peak search (core/marker_manager.py:74-127) ----------------------------------------------------------
@pytest.fixture(scope="module")
def markers(golden_dir):
    return np.load(os.path.join(golden_dir, "markers.npz"))


def test_marker_peaks_match_reference_vectors(pkg, an, markers):
    """Every case of tests/golden/markers.npz (what the imported MarkerManager did, real scipy find_peaks): the snap
    target, a walk of snap_to_next_peak calls, the peak list and the prominences, bit for bit."""
    keys = [str(k) for k in markers["cases"]]
    assert len(keys) >= 32
    for key in keys:
        _, n, kind, _ = key.split("_")
        n = int(n)
        tr = markers[f"trace_{n}_{kind}"]
        thr, exc, dist, start = markers[key + "_params"]
        want_pk, want_prom = markers[key + "_peaks"], markers[key + "_prom"]
        cap = max(len(want_pk) + 3, 8)
        with pkg.SpectrumEngine(max(1 << (n - 1).bit_length(), 64), max_frames=1) as e, DevRows(pkg, tr[None, :]) as d:
            kw = {} if bool(markers[key + "_defaults"]) else dict(peak_threshold=float(thr), peak_excursion=float(exc))
            r = an.rows_marker_peaks(e, d, 1, n_bins=n, current_idx=int(start), max_list=cap, **kw)
            assert r["n_peaks"][0] == len(want_pk), key
            assert np.array_equal(r["peaks"][0][: len(want_pk)], want_pk), key
            assert np.all(r["peaks"][0][len(want_pk):] == -1), key
            assert np.array_equal(r["prominences"][0][: len(want_pk)], want_prom), key
            assert np.all(np.isnan(r["prominences"][0][len(want_pk):])), key
            assert r["snap_bin"][0] == int(markers[key + "_snap"]), key
            pos, walk = int(start), []
            for _ in range(len(markers[key + "_walk"])):
                nxt = int(an.rows_marker_peaks(e, d, 1, n_bins=n, current_idx=pos, **kw)["next_bin"][0])
                pos = pos if nxt < 0 else nxt
                walk.append(pos)
            assert walk == list(markers[key + "_walk"]), key
            # a list shorter than the peaks: the first ones in bin order
            if len(want_pk) > 2:
                short = an.rows_marker_peaks(e, d, 1, n_bins=n, max_list=2, **kw)
                assert list(short["peaks"][0]) == list(want_pk[:2]) and short["n_peaks"][0] == len(want_pk)


def _marker_rows(rng, n, count):
    rows = []
    for _ in range(count):
        k = np.arange(n)
        kind = int(rng.integers(0, 6))
        p = rng.exponential(1.0, size=n) * 10.0 ** rng.uniform(-12, -6)
        for _t in range(int(rng.integers(0, 7))):
            c, a, w = rng.integers(0, n), 10.0 ** rng.uniform(-9, -2), rng.uniform(0.6, 4.0)
            p += a * np.sinc((k - c) / w) ** 2
        tr = 10 * np.log10(p + 1e-15)
        if kind == 1:
            tr = np.round(tr * 2) / 2                              # equal values, flat tops
        elif kind == 2:
            tr = np.minimum(tr, np.percentile(tr, 97))             # clipped: long plateaus
        elif kind == 3:
            tr = np.full(n, -120.0)                                # flat row (+ one flat-topped bump)
            a = int(rng.integers(1, max(2, n - 40)))
            tr[a:a + int(rng.integers(1, 38))] = -60.0
        elif kind == 4:
            tr = np.round(rng.normal(-80, 2, n))                   # dense ties
        elif kind == 5 and n > 8:
            tr[rng.integers(0, n, 3)] = np.nan
        rows.append(tr.astype(np.float32))
    return np.stack(rows)


@pytest.mark.parametrize("seed", range(int(os.environ.get("TDSA_SWEEP_CASES", "24"))))
def test_marker_peaks_random_traces(pkg, an, seed):
    """Seeded random rows (noise, tone combs, quantised and clipped traces, flat rows, NaNs; any length from 3 bins to
    16384, distances 1 ... 40) against the restatement of scipy's find_peaks, a batch of rows per call."""
    rng = np.random.default_rng(31000 + seed)
    n = int(rng.choice([3, 5, 33, 64, 1000, 1024, 4096, 5000, 16384, int(rng.integers(3, 16385))]))
    rows = _marker_rows(rng, n, 10)
    thr = float(rng.choice([-200.0, -90.0, -75.0]))
    exc = float(rng.choice([0.0, 3.0, 6.0, 10.0, 6.3, 0.7, 1e-3, 2.5000001, 250.0]))   # (exact and inexact against 0.5 dB steps)
    dist = int(rng.choice([1, 2, 3, 3, 3, 7, 40]))
    cur = int(rng.integers(-1, n + 1))
    with pkg.SpectrumEngine(64, max_frames=1) as e, DevRows(pkg, rows) as d:
        r = an.rows_marker_peaks(e, d, len(rows), n_bins=n, peak_threshold=thr, peak_excursion=exc, distance=dist,
                                 current_idx=cur, max_list=n // 2 + 1)
    for i, tr in enumerate(rows):
        pk, _, prom = ao.marker_find_peaks(tr, thr, exc, dist)
        got = r["peaks"][i][: r["n_peaks"][i]]
        assert np.array_equal(got, pk), (seed, i, n, dist)
        assert np.array_equal(r["prominences"][i][: len(pk)], prom), (seed, i)
        assert r["snap_bin"][i] == ao.snap_to_peak_bin(tr, thr, exc, dist), (seed, i)
        assert r["next_bin"][i] == ao.snap_to_next_peak_bin(tr, cur, thr, exc, dist), (seed, i)


def test_marker_peaks_on_spectra_and_errors(pkg, an):
    """C3-shaped GPU spectra (64 frames) row by row against the restatement, and the argument checks."""
    nfft, hop, nf = 16384, 8192, 64
    iq = so.synth_iq_int8(hop * (nf - 1) + nfft, nfft, seed=78)
    with pkg.SpectrumEngine(nfft, max_frames=nf) as e:
        e.set_window(so.hackrf_window(nfft))
        e.configure(db_mode="mag", log_floor=so.LOG_FLOOR, dc_alpha=1.0)
        rows = e.process(iq, hop=hop)
        with DevRows(pkg, rows) as d:
            r = an.rows_marker_peaks(e, d, nf, current_idx=nfft // 2, max_list=nfft // 2)
            for i in range(nf):
                pk, _, prom = ao.marker_find_peaks(rows[i])
                assert np.array_equal(r["peaks"][i][: r["n_peaks"][i]], pk), i
                assert np.array_equal(r["prominences"][i][: len(pk)], prom), i
                assert r["snap_bin"][i] == ao.snap_to_peak_bin(rows[i])
                assert r["next_bin"][i] == ao.snap_to_next_peak_bin(rows[i], nfft // 2)
            with pytest.raises(Exception):
                an.rows_marker_peaks(e, d, 1, n_bins=32768)
            with pytest.raises(Exception):
                an.rows_marker_peaks(e, d, 1, distance=0)
            with pytest.raises(Exception):
                an.rows_marker_peaks(e, d, 1, peak_excursion=float("nan"))


# ---- per-frame scalars from the frame kernel's epilogue (tdsa_set_frame_stats) -------------------------------------
def _stats_of_rows(rows, lo, hi):
    """what tdsa_rows_stats / the reference's np.max, np.argmax and sum(10 ** (levels / 10)) give per row"""
    peak = rows.max(axis=1)
    pbin = rows.argmax(axis=1).astype(np.int32)
    if lo > hi:
        band = np.zeros(len(rows))
    else:
        band = (10.0 ** (rows[:, lo:hi + 1] / np.float32(10.0))).astype(np.float64).sum(axis=1)
    return peak, pbin, band


def _check_frame_stats(e, rows, lo, hi, tag, exact_band=False):
    peak, pbin, band = e.frame_stats()
    wp, wb, wband = _stats_of_rows(rows, lo, hi)
    assert np.array_equal(peak, wp, equal_nan=True), tag
    assert np.array_equal(pbin, wb), tag
    if exact_band:
        assert np.allclose(band, wband, rtol=1e-6, atol=0), tag        # exp10f on the device, float32 10 ** x in numpy
    else:
        # the fused sum is formed from the linear power the kernel holds, the rows carry its rounded dB
        # (one float32 unit of a dB value near -110 is 1.8e-6 of the power; v_log_f32 and numpy's float32 10 ** x add theirs)
        assert np.allclose(band, wband, rtol=1e-5, atol=1e-300), (tag, np.max(np.abs(band / np.maximum(wband, 1e-300) - 1)))


@pytest.mark.parametrize("nfft", [1024, 2048, 4096, 8192, 16384])
@pytest.mark.parametrize("fmt", ["i8", "u8", "c64"])
def test_frame_stats_fused_match_rows(pkg, an, nfft, fmt):
    """peak / argmax bit for bit and the band power within 3e-6 of what the rows give, every fused size x input format x
    hold none / max x dB mode, bands of every kind (none, one bin, ends inside a wave's bins, the whole row)."""
    rng = np.random.default_rng(nfft + len(fmt))
    nf, hop = 13, nfft // 2
    iq = so.synth_iq_int8(hop * (nf - 1) + nfft, nfft, seed=nfft % 97)
    if fmt == "u8":
        iq = (iq.astype(np.int16) + 128).astype(np.uint8)
    elif fmt == "c64":
        iq = ((iq[0::2].astype(np.float32) + 1j * iq[1::2].astype(np.float32)) / 128).astype(np.complex64)
    bands = [(1, 0), (0, nfft - 1), (nfft // 2, nfft // 2), (5, 5), (nfft - 1, nfft - 1)]
    bands += [tuple(sorted(int(x) for x in rng.integers(0, nfft, 2))) for _ in range(5)]
    with pkg.SpectrumEngine(nfft, max_frames=nf) as e:
        e.set_window(so.hackrf_window(nfft))
        for k, (lo, hi) in enumerate(bands):
            mode = [dict(db_mode="mag", log_floor=so.LOG_FLOOR, dc_alpha=1.0, cal_offset_db=0.0),
                    dict(db_mode="pow", power_scale=1.0 / (20e6 * nfft), log_floor=1e-12, dc_alpha=-1.0, cal_offset_db=-0.8087),
                    dict(db_mode="pow", power_scale=1.0, log_floor=1e-10, dc_alpha=1.0, cal_offset_db=2.5)][k % 3]
            e.configure(hold_max=bool(k & 1), hold_min=False, **mode)
            e.set_frame_stats(True, (lo, hi))
            rows = e.process(iq, hop=hop)
            _check_frame_stats(e, rows, lo, hi, (nfft, fmt, k, lo, hi))
            # the same scalars as tdsa_rows_stats of those rows
            with DevRows(pkg, rows) as d:
                fb = np.arange(nfft, dtype=np.float64)
                peak, pbin, bdb = an.rows_stats(e, d, nf, freq_bins=fb, band=(lo, hi) if lo <= hi else None)
            p2, b2, band2 = e.frame_stats(bin_width=1.0)
            assert np.array_equal(peak, p2) and np.array_equal(pbin, b2)
            if lo <= hi:
                assert np.max(np.abs(band2 - bdb)) <= 5e-5, (nfft, fmt, k)             # dB


def test_frame_stats_special_rows(pkg):
    """silence (every bin at the floor: argmax 0), a NaN sample in complex64 input (np.max -> NaN, np.argmax -> the first
    NaN bin), equal maxima, and the statistics of a call that wrote no rows."""
    nfft, nf = 4096, 6
    with pkg.SpectrumEngine(nfft, max_frames=nf) as e:
        e.set_window(np.ones(nfft, dtype=np.float32))
        e.configure(db_mode="mag", log_floor=so.LOG_FLOOR, dc_alpha=-1.0)
        e.set_frame_stats(True, (100, 3000))
        silent = np.zeros(2 * nfft * nf, dtype=np.int8)
        rows = e.process(silent)
        assert np.all(rows == rows[0, 0])
        _check_frame_stats(e, rows, 100, 3000, "silence")
        assert np.all(e.frame_stats()[1] == 0)
        # an impulse: |X| the same in every bin up to rounding - many equal maxima
        imp = np.zeros(2 * nfft * nf, dtype=np.int8)
        imp[0::2 * nfft] = 64
        rows = e.process(imp)
        _check_frame_stats(e, rows, 100, 3000, "impulse")
        x = (np.random.default_rng(3).normal(size=nfft * nf) + 0j).astype(np.complex64)
        x[nfft + 7] = np.nan                                  # frame 1: every bin NaN
        rows = e.process(x)
        peak, pbin, band = e.frame_stats()
        assert np.isnan(peak[1]) and pbin[1] == 0 and np.isnan(band[1]) and not np.isnan(peak[[0, 2, 3]]).any()
        good = [0, 2, 3, 4, 5]
        wp, wb, wband = _stats_of_rows(rows[good], 100, 3000)
        assert np.array_equal(peak[good], wp) and np.array_equal(pbin[good], wb) and np.allclose(band[good], wband, rtol=1e-5)
        # no rows wanted: the fused path still reports
        iq = so.synth_iq_int8(nfft * nf, nfft, seed=5)
        rows = e.process(iq)
        want = e.frame_stats()
        assert e.process(iq, want_db=False) is None
        got = e.frame_stats()
        assert all(np.array_equal(a, b) for a, b in zip(want, got))
        assert all(np.array_equal(a, b) for a, b in zip(want, e.frame_stats(calls_back=1)))
        with pytest.raises(Exception):
            e.frame_stats(calls_back=4)
        e.set_frame_stats(False)
        with pytest.raises(Exception):
            e.frame_stats()


@pytest.mark.parametrize("case", ["n512", "n1000", "n1021", "avg", "tare", "holdmin"])
def test_frame_stats_unfused_plans_take_them_from_the_rows(pkg, case):
    """Plans / modes without the fused epilogue (frames below 1024 points, sizes that are not a power of two, averaging,
    tare, min hold) run rows_stats_kernel on the rows they wrote: identical to numpy on those rows."""
    nfft = {"n512": 512, "n1000": 1000, "n1021": 1021}.get(case, 2048)
    nf = 9
    iq = so.synth_iq_int8(nfft * nf, max(nfft, 64), seed=11)
    with pkg.SpectrumEngine(nfft, max_frames=nf) as e:
        e.set_window(np.hanning(nfft).astype(np.float32))
        e.configure(db_mode="pow", power_scale=1.0, log_floor=1e-10, dc_alpha=-1.0,
                    avg=("exp", 4) if case == "avg" else ("off", 1), hold_min=case == "holdmin")
        if case == "tare":
            e.set_tare_baseline(np.linspace(-3, 3, nfft).astype(np.float32))
        lo, hi = nfft // 5, nfft // 2
        e.set_frame_stats(True, (lo, hi))
        rows = e.process(iq)
        _check_frame_stats(e, rows, lo, hi, case, exact_band=True)
        if case in ("n512", "tare"):
            e.process(iq, want_db=False)
            with pytest.raises(Exception):
                e.frame_stats()                               # nothing to take them from


def test_frame_stats_full_c3_second_and_overlapped_calls(pkg, an):
    """One second of C3 (2440 frames of 16384 points, hop 8192, max hold) on the device path: the fused scalars against
    tdsa_rows_stats of the rows the same call wrote; then three overlapped calls, each read back by calls_back."""
    nat = pkg._native
    nfft, hop, nf = 16384, 8192, 2440
    n_samples = hop * (nf - 1) + nfft
    iq = so.synth_iq_int8(n_samples, nfft, seed=3)
    d_in, d_out = C.c_void_p(), C.c_void_p()
    nat.check(nat.lib.tdsa_dev_alloc(0, iq.nbytes, C.byref(d_in)))
    nat.check(nat.lib.tdsa_dev_alloc(0, nf * nfft * 4, C.byref(d_out)))
    try:
        nat.check(nat.lib.tdsa_memcpy_h2d(0, d_in, iq.ctypes.data_as(C.c_void_p), iq.nbytes))
        with pkg.SpectrumEngine(nfft, max_frames=nf) as e:
            e.set_window(so.hackrf_window(nfft))
            e.configure(db_mode="mag", log_floor=so.LOG_FLOOR, dc_alpha=1.0, hold_max=True)
            lo, hi = 3000, 11000
            e.set_frame_stats(True, (lo, hi))
            e.process_device(nat.IN_I8, d_in.value, n_samples, hop, nf, d_out.value)
            peak, pbin, band = e.frame_stats(bin_width=1.0)
            rp, rb, rband = an.rows_stats(e, d_out.value, nf, freq_bins=np.arange(nfft, dtype=np.float64), band=(lo, hi))
            assert np.array_equal(peak, rp) and np.array_equal(pbin, rb)
            assert np.max(np.abs(band - rband)) <= 5e-5
            mx, _ = e.hold()
            rows = np.empty((nf, nfft), dtype=np.float32)
            nat.check(nat.lib.tdsa_memcpy_d2h(0, rows.ctypes.data_as(C.c_void_p), d_out, rows.nbytes))
            assert np.array_equal(mx, rows.max(axis=0))           # the hold trace is untouched by the extra epilogue
            assert peak.max() == mx.max()
            # overlapped calls of different lengths: each slot keeps its own call
            e.set_overlap(3)
            lens = [2440, 1000, 37]
            for n in lens:
                e.process_device(nat.IN_I8, d_in.value, hop * (n - 1) + nfft, hop, n, None)
            for back, n in enumerate(reversed(lens)):
                p2, b2, _ = e.frame_stats(calls_back=back)
                assert len(p2) == n and np.array_equal(p2, peak[:n]) and np.array_equal(b2, pbin[:n])
    finally:
        nat.lib.tdsa_dev_free(0, d_in)
        nat.lib.tdsa_dev_free(0, d_out)


def test_frame_stats_batched_captures(pkg):
    """tdsa_process_dev_batch: the frames of all captures of the one launch, capture after capture."""
    nat = pkg._native
    nfft, hop, nf, nseg = 8192, 8192, 32, 5
    per = hop * (nf - 1) + nfft
    iq = so.synth_iq_int8(per * nseg, nfft, seed=9)
    d_in, d_out = C.c_void_p(), C.c_void_p()
    nat.check(nat.lib.tdsa_dev_alloc(0, iq.nbytes, C.byref(d_in)))
    nat.check(nat.lib.tdsa_dev_alloc(0, nseg * nf * nfft * 4, C.byref(d_out)))
    try:
        nat.check(nat.lib.tdsa_memcpy_h2d(0, d_in, iq.ctypes.data_as(C.c_void_p), iq.nbytes))
        with pkg.SpectrumEngine(nfft, max_frames=nf) as e:
            e.set_window(np.hanning(nfft).astype(np.float32))
            e.configure(db_mode="pow", power_scale=1.0, log_floor=1e-10, dc_alpha=-1.0)
            e.set_frame_stats(True, (10, 8000))
            e.process_device_batch(nat.IN_I8, d_in.value, per * 2, nseg, per, hop, nf, d_out.value)
            rows = np.empty((nseg * nf, nfft), dtype=np.float32)
            e.synchronize()
            nat.check(nat.lib.tdsa_memcpy_d2h(0, rows.ctypes.data_as(C.c_void_p), d_out, rows.nbytes))
            _check_frame_stats(e, rows, 10, 8000, "batch")
    finally:
        nat.lib.tdsa_dev_free(0, d_in)
        nat.lib.tdsa_dev_free(0, d_out)


# ---- levels -> uint8 views (what ImageItem.setImage makes of the images before the colour table) -------------------
def _levels_u8(img, lo, hi):
    """np.clip((view - lo) / (hi - lo) * 255, 0, 255).astype(uint8) in float32 (numpy >= 2 keeps float32 against Python
    floats); NaN pixels -> 0"""
    with np.errstate(invalid="ignore"):
        t = np.clip((img - lo) / (hi - lo) * 255, 0, 255)
    return np.where(np.isnan(t), 0, t).astype(np.uint8)


def test_waterfall_view_u8_matches_setimage_levels(pkg, an, golden_dir):
    """displays/waterfall.py:353-356: the view of the reference fixture's sequence under three level pairs, byte for byte;
    NaN / -inf / +inf pixels included."""
    g = np.load(os.path.join(golden_dir, "displays.npz"))
    H, W = int(g["wf_history_lines"]), g["wf_rows"].shape[1]
    with an.WaterfallRing(H, W, float(g["wf_min_db"])) as wf:
        for idx in g["wf_order"]:
            wf.push(g["wf_rows"][idx])
        view = wf.view()
        assert np.array_equal(view, g["wf_views"][-1])
        for lo, hi in ((float(g["wf_min_db"]), float(g["wf_min_db"]) + 80.0), (-100.0, -20.0), (-63.7, -61.2)):
            assert np.array_equal(wf.view_u8(lo, hi), _levels_u8(view, lo, hi)), (lo, hi)
        with pytest.raises(Exception):
            wf.view_u8(-20.0, -20.0)
    rng = np.random.default_rng(12)
    with an.WaterfallRing(33, 1001, -120.0) as wf:               # ragged sizes, special values
        rows = rng.normal(-70, 15, size=(50, 1001)).astype(np.float32)
        rows[3, 5], rows[7, 9], rows[9, 11] = np.nan, -np.inf, np.inf
        for r in rows:
            wf.push(r)
        assert np.array_equal(wf.view_u8(-110.0, -30.0), _levels_u8(wf.view(), -110.0, -30.0))


def test_density_image_u8_matches_setimage_autolevels(pkg, an, golden_dir):
    """displays/density_display.py:318: np.log1p(hist) under its own minimum / maximum as levels."""
    g = np.load(os.path.join(golden_dir, "displays.npz"))
    rows = np.ascontiguousarray(g["density_rows"])
    with an.DensityHistogram(rows.shape[1], float(g["density_medium_decay"])) as dh:
        u8, (lo, hi) = dh.image_u8()
        assert not u8.any() and lo == hi == 0.0                    # empty histogram
        for row in rows[:20]:
            dh.update(row)
        img = dh.image()
        u8, (lo, hi) = dh.image_u8()
        assert lo == img.min() and hi == img.max()
        assert np.array_equal(u8, _levels_u8(img, lo, hi)) and u8.max() == 255


@pytest.mark.parametrize("seed", range(int(os.environ.get("TDSA_SWEEP_CASES", "24"))))
def test_frame_stats_random(pkg, seed):
    """Seeded random plans - any size from 64 to 16384 incl. sizes that are not a power of two, the three input formats,
    dB modes, calibration offsets, hold combinations, hops, frame counts, bands - every one against numpy on the rows the
    same call wrote: peak and argmax bit for bit (fused or not), the band within 1e-5 (fused) / 1e-6 (from the rows)."""
    rng = np.random.default_rng(52000 + seed)
    nfft = int(rng.choice([64, 256, 512, 1000, 1024, 1024, 2048, 2048, 3000, 4096, 4096, 8192, 8192, 16384, 16384]))
    nf = int(rng.integers(1, 40))
    hop = int(rng.choice([nfft, nfft // 2, int(rng.integers(1, 2 * nfft))]))
    iq = so.synth_iq_int8(hop * (nf - 1) + nfft, max(nfft, 64), seed=int(rng.integers(1, 1 << 30)))
    fmt = str(rng.choice(["i8", "u8", "c64"]))
    if fmt == "u8":
        iq = (iq.astype(np.int16) + 128).astype(np.uint8)
    elif fmt == "c64":
        iq = ((iq[0::2].astype(np.float32) + 1j * iq[1::2].astype(np.float32)) / 128).astype(np.complex64)
    mode = [dict(db_mode="mag", log_floor=so.LOG_FLOOR), dict(db_mode="pow", power_scale=1.0 / (20e6 * nfft), log_floor=1e-12),
            dict(db_mode="pow", power_scale=1.0, log_floor=1e-10)][int(rng.integers(0, 3))]
    lo, hi = sorted(int(x) for x in rng.integers(0, nfft, 2))
    if rng.integers(0, 5) == 0:
        lo, hi = 1, 0
    with pkg.SpectrumEngine(nfft, max_frames=nf) as e:
        e.set_window(np.hanning(nfft).astype(np.float32) if rng.integers(0, 2) else so.hackrf_window(nfft))
        e.configure(dc_alpha=float(rng.choice([1.0, -1.0, 0.3])), cal_offset_db=float(rng.choice([0.0, -0.8087, 3.5])),
                    hold_max=bool(rng.integers(0, 2)), hold_min=bool(rng.integers(0, 4) == 0), **mode)
        e.set_frame_stats(True, (lo, hi))
        rows = e.process(iq, hop=hop, n_frames=nf)
        peak, pbin, band = e.frame_stats()
    wp, wb, wband = _stats_of_rows(rows, lo, hi)
    tag = (seed, nfft, fmt, nf, hop, lo, hi)
    assert np.array_equal(peak, wp) and np.array_equal(pbin, wb), tag
    assert np.allclose(band, wband, rtol=1e-5, atol=1e-300), tag
