"""oracle/analytics_oracle.py against the vectors captured from the imported reference
(tests/golden/analytics.npz) and hand-derived known-answer cases for the two display accumulators the
reference cannot run headless (PyQt6 / pyqtgraph absent: parity unpinned by the reference there)."""
import os

import numpy as np
import pytest

from oracle import analytics_oracle as ao


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "analytics.npz"))


def test_top_peaks_match_reference(gold):
    for key in gold["peak_cases"]:
        key = str(key)
        n, kind, exc = key.split("_")[1:]
        tr = gold[key + "_trace"]
        min_sep, _ = ao.peak_list_params(int(n))
        bins = ao.find_top_peak_bins(tr, n=5, min_sep_bins=min_sep, min_excursion_db=float(exc))
        assert bins == list(gold[key + "_bins"]), key
        assert np.array_equal(tr[bins].astype(np.float64), gold[key + "_pwr"]), key


def test_top_peaks_edge_cases():
    assert ao.find_top_peaks(np.arange(2.0), np.array([1.0, 2.0])) == []
    assert ao.find_top_peaks(np.arange(5.0), np.array([5.0, 4, 3, 2, 1])) == []          # edges never count
    flat = np.zeros(16)
    assert ao.find_top_peaks(np.arange(16.0), flat) == []                                # strict maxima only
    tr = np.full(64, -100.0)
    tr[10], tr[30], tr[33] = -20.0, -30.0, -25.0
    assert ao.find_top_peak_bins(tr, min_sep_bins=10, min_excursion_db=10.0) == [10, 33]   # 30 too close to 33
    assert ao.find_top_peak_bins(tr, min_sep_bins=2, min_excursion_db=10.0) == [10, 33, 30]
    tr2 = np.full(64, -100.0)
    tr2[20:41] = -24.0
    tr2[20], tr2[40] = -20.0, -21.0                        # valley between them only 3-4 dB deep
    assert ao.find_top_peak_bins(tr2, min_sep_bins=5, min_excursion_db=10.0) == [20]


def test_duty_cycle_matches_reference(gold):
    d = ao.DutyCycleOracle()
    frames = gold["duty_frames"]
    for i, fr in enumerate(frames):
        d.update_from_power(fr, threshold_dbm=-60.0 if i < 130 else -45.0)
        assert d.duty_pct == gold["duty_pct"][i]
        on, off = gold["duty_on"][i], gold["duty_off"][i]
        assert (d.on_power_dbm is None and np.isnan(on)) or d.on_power_dbm == on
        assert (d.off_power_dbm is None and np.isnan(off)) or d.off_power_dbm == off
    d.reset()
    assert d.duty_pct == 0.0 and d.on_power_dbm is None


def test_band_power_matches_reference(gold):
    bins, tr = gold["band_bins"], gold["band_trace"]
    for (a, b), want in zip(gold["band_edges"], gold["band_db"]):
        got = ao.band_power_db(bins, tr, a, b)
        if np.isnan(want):
            assert got is None
            assert ao.band_bin_range(bins, a, b) == (0, -1)
        else:
            assert got == want
            lo, hi = ao.band_bin_range(bins, a, b)
            assert np.array_equal(np.where((bins >= min(a, b)) & (bins <= max(a, b)))[0], np.arange(lo, hi + 1))


def test_frame_peak():
    tr = np.array([-3.0, 7.5, 7.5, -1.0], dtype=np.float32)
    assert ao.frame_peak(tr) == (7.5, 1)            # first of equal maxima


def test_density_known_answers():
    d = ao.DensityOracle(decay=0.5)
    row = np.array([-200.0, -199.5, 99.9, 100.0, np.nan, -201.0, -200.1], dtype=np.float32)
    h = d.update(row)
    assert h.shape == (7, ao.AMP_BINS) and h.dtype == np.float32
    assert h[0, 0] == 1 and h[1, 0] == 1 and h[2, 511] == 1          # 0.5 dB * 512/300 = 0.85 -> bin 0
    assert h[3].sum() == 0 and h[4].sum() == 0 and h[5].sum() == 0   # == +100 dB, NaN, below range: dropped
    assert h[6, 0] == 1                       # astype(int32) truncates toward zero: (-200.59, -200) -> bin 0
    h = d.update(row)
    assert h[0, 0] == 1.5 and h[2, 511] == 1.5
    d.update(np.full(7, -50.0, dtype=np.float32))
    assert d.hist[0, 0] == 0.75 and d.hist[0, int((150.0 / 300.0) * 512)] == 1.0
    assert np.allclose(d.image(), np.log1p(d.hist))
    d.update(np.zeros(3, dtype=np.float32))                           # size change: fresh histogram
    assert d.hist.shape == (3, ao.AMP_BINS) and d.hist.sum() == 3.0
    nodecay = ao.DensityOracle(decay=1.0)
    for _ in range(4):
        nodecay.update(np.array([0.0], dtype=np.float32))
    assert nodecay.hist.sum() == 4.0


def test_waterfall_known_answers():
    w = ao.WaterfallOracle(history_lines=3, n_bins=2, min_db=-120.0)
    assert np.all(w.view() == -120.0) and w.view().shape == (3, 2)
    rows = [np.array([1.0, 2.0]), np.array([1.0, 2.0]), np.array([3.0, 4.0]), np.array([5.0, 6.0]),
            np.array([7.0, 8.0])]
    news = [w.update(r) for r in rows]
    assert news == [True, False, True, True, True]                    # identical consecutive row is skipped
    assert np.array_equal(w.view(), np.array([[7, 8], [5, 6], [3, 4]], dtype=np.float32))   # newest first
    assert w.buf.shape == (6, 2) and np.array_equal(w.buf[:3], w.buf[3:])


# ---- display accumulators pinned to the imported reference classes (tests/golden/displays.npz) ----
@pytest.mark.parametrize("mode", ["medium", "fast", "off"])
def test_density_oracle_matches_reference_fixture(golden_dir, mode):
    g = np.load(os.path.join(golden_dir, "displays.npz"))
    d = ao.DensityOracle(decay=float(g[f"density_{mode}_decay"]))
    snaps = {int(r): i for i, r in enumerate(g["density_snap_rows"])}
    for r, row in enumerate(g["density_rows"]):
        with np.errstate(invalid="ignore"):
            d.update(row)
        if r in snaps:
            assert np.array_equal(d.hist, g[f"density_{mode}_hist"][snaps[r]]), (mode, r)


def test_waterfall_oracle_matches_reference_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "displays.npz"))
    w = ao.WaterfallOracle(int(g["wf_history_lines"]), g["wf_rows"].shape[1], float(g["wf_min_db"]))
    steps = {int(s): i for i, s in enumerate(g["wf_view_steps"])}
    for step, idx in enumerate(g["wf_order"]):
        assert w.update(g["wf_rows"][idx]) == bool(g["wf_added"][step])
        assert w.ptr == int(g["wf_ptr"][step])
        if step in steps:
            assert np.array_equal(w.view(), g["wf_views"][steps[step]])


# ---- marker peak search: oracle vs what the imported MarkerManager did (tests/golden/markers.npz) ------------------
@pytest.fixture(scope="module")
def markers(golden_dir):
    return np.load(os.path.join(golden_dir, "markers.npz"))


def marker_case(markers, key):
    _, n, kind, _ = key.split("_")
    thr, exc, dist, start = markers[key + "_params"]
    return markers[f"trace_{n}_{kind}"], float(thr), float(exc), int(dist), int(start)


def test_marker_find_peaks_matches_reference(markers):
    assert len(markers["cases"]) >= 32
    some_peaks = some_fallback = some_plateau = 0
    for key in map(str, markers["cases"]):
        tr, thr, exc, dist, start = marker_case(markers, key)
        pk, heights, prom = ao.marker_find_peaks(tr, thr, exc, dist)
        assert np.array_equal(pk, markers[key + "_peaks"]), key
        assert np.array_equal(prom, markers[key + "_prom"]), key
        assert np.array_equal(heights, tr[pk].astype(np.float64)), key
        assert ao.snap_to_peak_bin(tr, thr, exc, dist) == int(markers[key + "_snap"]), key
        pos, walk = start, []
        for _ in range(len(markers[key + "_walk"])):
            nxt = ao.snap_to_next_peak_bin(tr, pos, thr, exc, dist)
            pos = pos if nxt < 0 else nxt                     # no peak: the reference leaves the marker alone
            walk.append(pos)
        assert walk == list(markers[key + "_walk"]), key
        some_peaks += len(pk) > 0
        some_fallback += len(pk) == 0
        some_plateau += bool(np.any(tr[pk] == tr[np.minimum(pk + 1, len(tr) - 1)])) if len(pk) else 0
    assert some_peaks and some_fallback and some_plateau     # the fixture exercises every branch


def test_marker_find_peaks_edge_cases():
    flat = np.zeros(16, dtype=np.float32)
    assert len(ao.marker_find_peaks(flat)[0]) == 0 and ao.snap_to_peak_bin(flat) == 0
    assert ao.snap_to_next_peak_bin(flat, 5) == -1
    x = np.full(32, -100.0, dtype=np.float32)
    x[10:13] = -20.0                                          # flat top of three: its middle
    x[28:] = -10.0                                            # a plateau that runs into the last sample is no peak
    pk, h, pr = ao.marker_find_peaks(x)
    assert list(pk) == [11] and h[0] == -20.0 and pr[0] == 80.0
    x[20], x[22] = -30.0, -31.0                               # closer than 3 bins: the higher one stays
    assert list(ao.marker_find_peaks(x)[0]) == [11, 20]
    assert list(ao.marker_find_peaks(x, distance=1)[0]) == [11, 20, 22]
    assert list(ao.marker_find_peaks(x, height=-25.0)[0]) == [11]
    # prominence: the walk stops at the first higher sample on each side, the higher of the two bases counts
    y = np.full(32, -100.0, dtype=np.float32)
    y[16:25] = [-28.0, -35.0, -34.0, -32.0, -30.0, -33.0, -31.0, -34.0, -29.0]
    y[5] = -10.0
    pk, _, pr = ao.marker_find_peaks(y, prominence=0.0)
    assert list(pk) == [5, 16, 20, 24] and list(pr) == [90.0, 72.0, 4.0, 6.0]      # 22 fell to the distance rule
    assert list(ao.marker_find_peaks(y, prominence=6.0)[0]) == [5, 16, 24]          # `>=`: exactly 6 dB qualifies
    assert ao.snap_to_next_peak_bin(y, 24, prominence=6.0) == 5          # wraps to the first peak
    assert ao.snap_to_next_peak_bin(y, 5, prominence=6.0) == 16
    z = x.copy()
    z[3] = np.nan                                             # NaN: no comparison holds, walks stop there
    assert list(ao.marker_find_peaks(z)[0]) == [11, 20]
