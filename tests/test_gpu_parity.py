"""GPU parity: HIP path (through the C-ABI) vs the float64 gold oracle on identical IQ.

Tolerances (BASELINE.json north_star: 1e-4 relative float32):
  * linear power error <= 1e-4 * frame maximum on every bin  (REL_TOL)
  * |dB error| <= 1e-3 dB on every bin within 100 dB of the frame maximum (DB_TOL), where the allowance
    of a deep bin (from about 60 dB down) is TWO float32 rounding units of the frame's largest AMPLITUDE
    instead (2 * 2^-24 = 2^-23 * A_max, `_check` below passes amp_floor = 2 * so.AMP_FLOOR; no float32 FFT
    resolves 1e-3 dB a hundred dB under a tone; oracle/spectrum_oracle.py::parity_metrics states the rule and
    DESIGN.md section 2 the measurements it rests on).  SURVEY.md 8(d)'s literal bound - 1e-3 dB over the whole
    100 dB with no allowance - is NOT met by this path (nor by numpy's own float32 FFT): bench.py reports it
    separately as parity.survey_8d_strict_pass.
"""
import os
import time

import numpy as np
import pytest

from oracle import spectrum_oracle as so

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4
DB_TOL = 1e-3


@pytest.fixture(scope="module")
def pkg():
    import topdogspectrumanalyser_amd as p
    return p


def _check(db_gpu, db_gold, what="", floor_units=None, rows_gold=None):
    """Allowance 1e-3 dB, or two float32 rounding units of the frame's largest amplitude (2^-23 A_max) where that
    is worth more: the bound that held over the 3000-configuration soak of tools/parity_soak.py (median 0.14 units,
    worst 1.38: the bin N/2 away from an on-bin full-scale tone; long frames worst 1.03).
    rows_gold: db_gpu / db_gold are a max / min HOLD trace over these gold rows - its allowance is what follows from
    theirs, bin by bin (so.HoldAllowance: |max a - max b| <= max |a - b|), not that of a single row."""
    units = floor_units if floor_units is not None else 2
    if rows_gold is not None:
        rel, ddb = so.HoldAllowance(amp_floor=units * so.AMP_FLOOR).update(rows_gold).metrics(db_gpu, db_gold)
    else:
        rel, ddb = so.parity_metrics(db_gpu, db_gold, amp_floor=units * so.AMP_FLOOR)
    assert rel <= REL_TOL and ddb <= DB_TOL, f"{what}: rel={rel:.3e} ddb={ddb:.3e}"
    return rel, ddb


def _two_transforms(nfft):
    """sizes that run as a chirp-z convolution (two transforms of M >= 2 N points: about 1.6 x the error variance of one
    transform - profiles/r05_parity_soak.txt): neither a power of two nor made of the factors 2, 3, 5"""
    n = int(nfft)
    for f in (2, 3, 5):
        while n % f == 0:
            n //= f
    return n != 1


def _hackrf_engine(pkg, nfft, nf, **cfg):
    e = pkg.SpectrumEngine(nfft, max_frames=nf)
    e.set_window(so.hackrf_window(nfft))
    base = dict(db_mode="mag", log_floor=so.LOG_FLOOR, dc_alpha=1.0)
    base.update(cfg)
    e.configure(**base)
    return e


# ------------------------------------------------------------------------------------------------
# every size, both input formats
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nfft", [64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384])
def test_hackrf_plain_int8_all_sizes(pkg, nfft):
    nf = 5
    hop = nfft // 2
    iq = so.synth_iq_int8(hop * (nf - 1) + nfft, nfft, seed=nfft)
    gold, gmax, gmin = so.hackrf_batch(iq, nfft, hop, 20e6, precision="gold")
    with _hackrf_engine(pkg, nfft, nf, hold_max=True, hold_min=True) as e:
        out = e.process(iq, hop=hop)
        mx, mn = e.hold()
    assert out.shape == gold.shape and out.dtype == np.float32
    _check(out, gold, f"N={nfft}")
    _check(mx, gmax, "max hold", rows_gold=gold)
    _check(mn, gmin, "min hold", rows_gold=gold)
    assert np.array_equal(mx, out.max(axis=0)) and np.array_equal(mn, out.min(axis=0))


@pytest.mark.parametrize("nfft", [3, 7, 12, 32, 33, 65, 100, 1000, 1009, 1536, 3000, 4095, 4097, 6000, 8000, 8191])
def test_hackrf_plain_int8_sizes_that_are_not_a_power_of_two(pkg, nfft):
    """HackrfSamplesDataSource.set_num_samples takes any positive size and np.fft.fft any N
    (hackrf_samples.py:392-405, :370): every size up to 8192 has a device path (chirp-z on the frame kernel)."""
    nf = 5
    hop = max(1, nfft // 2)
    iq = so.synth_iq_int8(hop * (nf - 1) + nfft, nfft, seed=nfft)
    gold, gmax, gmin = so.hackrf_batch(iq, nfft, hop, 20e6, precision="gold")
    with _hackrf_engine(pkg, nfft, nf, hold_max=True, hold_min=True) as e:
        out = e.process(iq, hop=hop)
        mx, mn = e.hold()
        assert e.info().nfft == nfft
    assert out.shape == gold.shape and out.dtype == np.float32
    _check(out, gold, f"N={nfft}")
    _check(mx, gmax, "max hold", rows_gold=gold)
    _check(mn, gmin, "min hold", rows_gold=gold)
    assert np.array_equal(mx, out.max(axis=0)) and np.array_equal(mn, out.min(axis=0))


@pytest.mark.parametrize("nfft", [8193, 10000, 12000, 16383, 16385, 20000, 30011, 65537, 100000, 262145, 500000, 524287,
                                  524289, 600000, 777777, 1000000, 1048575])
@pytest.mark.parametrize("branch", ["hackrf", "rtl_exp"])
def test_long_frames_that_are_not_a_power_of_two(pkg, nfft, branch):
    """np.fft.fft / scipy.fft.fft take any N and the sources' size setters any positive size (hackrf_samples.py:370,
    :392-405; rtl_samples.py:170, :208-214): sizes above 8192 that are not a power of two run as a chirp-z convolution
    whose two M-point transforms (M = 2^ceil(log2(2N-1)) = 2^15 .. 2^20) go through the long-frame kernels - the second
    one transposed, rows first (tdsa_big.hip).  Above 2^19 points M would be 2^21: those frames run as four half-length
    sub-convolutions of 2^20 points (two half-rows per frame, three filter segments: tdsa_chirp.hip).  HackRF branch with
    both hold traces, and the RTL branch with exponential averaging (one frame per call on the float64 state)."""
    if nfft > 70000 and branch == "rtl_exp" and nfft != 600000:
        pytest.skip("covered by the HackRF branch at this size")
    nf = 3
    hop = nfft // 2 if branch == "hackrf" else nfft
    iq = so.synth_iq_int8(hop * (nf - 1) + nfft, 4096, seed=nfft % 1000)
    if branch == "hackrf":
        gold, gmax, gmin = so.hackrf_batch(iq, nfft, hop, 20e6, precision="gold")
        with _hackrf_engine(pkg, nfft, nf, hold_max=True, hold_min=True) as e:
            assert e.info().nfft == nfft
            out = e.process(iq, hop=hop)
            mx, mn = e.hold()
        assert out.shape == gold.shape and out.dtype == np.float32
        _check(out, gold, f"N={nfft}")
        _check(mx, gmax, "max hold", rows_gold=gold)
        assert np.array_equal(mx, out.max(axis=0)) and np.array_equal(mn, out.min(axis=0))
    else:
        gold, _, _ = so.rtl_batch(iq, nfft, hop, 2e6, precision="gold", avg=("exp", 3))
        with pkg.SpectrumEngine(nfft, max_frames=nf) as e:
            e.set_window(so.rtl_window("hanning", nfft).astype(np.float32))
            e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=-1.0, avg=("exp", 3))
            out = e.process(iq, hop=hop)
        _check(out, gold, f"RTL exp N={nfft}")


@pytest.mark.parametrize("nfft", [4, 6, 12, 20, 60, 96, 100, 250, 360, 625, 1000, 1500, 2000, 2187, 3000, 3125, 4050, 6000,
                                  6561, 7776, 8000, 8100, 9000, 10000])
def test_sizes_made_of_the_factors_2_3_5(pkg, nfft):
    """Frame lengths 2^a 3^b 5^c up to 10 000 points - what a user types into set_num_samples / set_fft_size
    (hackrf_samples.py:392-405, rtl_samples.py:208-214) - run as a mixed-radix Stockham transform of exactly N points
    (tdsa_smooth.hip; radices 4, 2, 3, 5) instead of the chirp-z convolution: HackRF branch with both hold traces from byte
    and complex64 samples, RTL branch (uint8) with linear averaging, against the float64 gold; and the chirp-z path of
    the same plan (tdsa_debug_knob smooth 0) must agree with it within the same bounds."""
    nf = 5
    hop = max(1, (2 * nfft) // 3)
    n = hop * (nf - 1) + nfft
    iq = so.synth_iq_int8(n, max(64, nfft), seed=nfft % 997)
    x = so.unpack_iq_int8(iq)
    gold, gmax, gmin = so.hackrf_batch(iq, nfft, hop, 20e6, precision="gold")
    for feed in (iq, x):
        for smooth in (1, 0):
            with _hackrf_engine(pkg, nfft, nf, hold_max=True, hold_min=True) as e:
                e.debug_knob("smooth", smooth)
                out = e.process(feed, hop=hop, n_frames=nf)
                mx, mn = e.hold()
            _check(out, gold, f"N={nfft} {feed.dtype} smooth={smooth}")
            _check(mx, gmax, "max hold", rows_gold=gold)
            _check(mn, gmin, "min hold", rows_gold=gold)
            assert np.array_equal(mx, out.max(axis=0)) and np.array_equal(mn, out.min(axis=0))
    # calls of more than eight frames take the frame means from the sums kernel, smaller ones form them in the transform's
    # kernel (byte samples): the rows of the first frames must be the same bits either way
    nf2 = 12
    iq2 = so.synth_iq_int8(hop * (nf2 - 1) + nfft, max(64, nfft), seed=nfft % 997)
    gold2, _, _ = so.hackrf_batch(iq2, nfft, hop, 20e6, precision="gold")
    with _hackrf_engine(pkg, nfft, nf2) as e:
        out2 = e.process(iq2, hop=hop, n_frames=nf2)
        dc2 = e.dc_estimate
    _check(out2, gold2, f"N={nfft} 12 frames")
    with _hackrf_engine(pkg, nfft, nf2) as e:
        out3 = np.concatenate([e.process(iq2[: 2 * (hop * 5 + nfft)], hop=hop, n_frames=6),
                               e.process(iq2[2 * hop * 6:], hop=hop, n_frames=6)])
        assert e.dc_estimate == dc2
    assert np.array_equal(out2, out3)
    rng = np.random.default_rng(nfft)
    u8 = rng.integers(0, 256, size=2 * n, dtype=np.uint8)
    xr = so.unpack_iq_uint8_rtl(u8)
    br = so.RtlBranchOracle(nfft, 2e6, "hamming", precision="gold")
    br.averager.set_mode("lin", 4)
    gavg = np.stack([np.array(br.power_levels(xr[k * hop:k * hop + nfft])) for k in range(nf)])
    with pkg.SpectrumEngine(nfft, max_frames=nf) as e:
        e.set_window(so.rtl_window("hamming", nfft).astype(np.float32))
        e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=-1.0, avg=("lin", 4))
        out = e.process(u8, hop=hop, n_frames=nf)
    _check(out, gavg, f"N={nfft} RTL lin 4")


@pytest.mark.parametrize("nfft", [10240 + 2560, 12000, 20000, 30000, 50000, 100000, 250000, 600000, 1000000, 1048576 - 65536])
def test_long_sizes_made_of_the_factors_2_3_5(pkg, nfft):
    """2^a 3^b 5^c above the LDS limit (10 000 points) up to 2^20: the mixed-radix transform in two passes (n1-point transforms
    of adjacent columns, times W_N^(n2 k1), through a complex64 buffer, n2-point transforms of adjacent rows: tdsa_smooth.hip)
    instead of the chirp-z convolution on the long-frame kernels; both paths against the gold, byte and complex64 samples,
    hold traces, and the RTL branch with exponential averaging."""
    nf = 3
    hop = nfft // 2
    iq = so.synth_iq_int8(hop * (nf - 1) + nfft, 4096, seed=nfft % 1000)
    gold, gmax, gmin = so.hackrf_batch(iq, nfft, hop, 20e6, precision="gold")
    for feed, smooth in ((iq, 1), (iq, 0), (so.unpack_iq_int8(iq), 1)):
        with _hackrf_engine(pkg, nfft, nf, hold_max=True, hold_min=True) as e:
            e.debug_knob("smooth", smooth)
            out = e.process(feed, hop=hop, n_frames=nf)
            mx, mn = e.hold()
        _check(out, gold, f"N={nfft} {feed.dtype} smooth={smooth}")
        _check(mx, gmax, "max hold", rows_gold=gold)
        assert np.array_equal(mx, out.max(axis=0)) and np.array_equal(mn, out.min(axis=0))
    if nfft <= 100000:
        golda, _, _ = so.rtl_batch(iq, nfft, nfft, 2e6, precision="gold", avg=("exp", 3))
        with pkg.SpectrumEngine(nfft, max_frames=nf) as e:
            e.set_window(so.rtl_window("hanning", nfft).astype(np.float32))
            e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=-1.0, avg=("exp", 3))
            out = e.process(iq, hop=nfft)
        _check(out, golda, f"RTL exp N={nfft}")


@pytest.mark.parametrize("nfft", [1000, 6000, 20000, 100003, 600000])
@pytest.mark.parametrize("fmt", ["i8", "u8", "c64"])
def test_chirp_plans_every_input_format_and_both_code_paths(pkg, nfft, fmt):
    """The chirp-z plans carry their element-wise passes inside the transforms (M <= 16384: one launch) or inside the
    long-frame column passes (M > 16384) - raw frames of each of the three input formats are unpacked by those loads.
    The alternatives the debug knobs keep for A/B (two launches; the passes as kernels of their own) must give rows
    within the same bounds, and the hold traces must be the folds of the rows whichever path made them."""
    nf = 3
    hop = (2 * nfft) // 3
    rng = np.random.default_rng(nfft)
    n = hop * (nf - 1) + nfft
    if fmt == "i8":
        iq = so.synth_iq_int8(n, 4096, seed=nfft % 997)
        x = so.unpack_iq_int8(iq)
        feed = iq
    elif fmt == "u8":
        feed = rng.integers(0, 256, size=2 * n, dtype=np.uint8)
        t = np.arange(n)
        feed[0::2] = np.clip(127.5 + 100 * np.cos(2 * np.pi * 0.123 * t) + rng.normal(0, 3, n), 0, 255).astype(np.uint8)
        feed[1::2] = np.clip(127.5 + 100 * np.sin(2 * np.pi * 0.123 * t) + rng.normal(0, 3, n), 0, 255).astype(np.uint8)
        x = so.unpack_iq_uint8_rtl(feed)
    else:
        iq = so.synth_iq_int8(n, 4096, seed=nfft % 991)
        x = so.unpack_iq_int8(iq)
        feed = x
    br = so.HackrfBranchOracle(nfft, 20e6, precision="gold")
    gold = np.stack([br.power_levels(x[k * hop:k * hop + nfft]) for k in range(nf)])
    # (1000 and 6000 are 2^a 5^b 3^c: by default a mixed-radix transform of exactly N points, tdsa_smooth.hip; "smooth" 0
    # sends them through the chirp-z convolution like the other sizes)
    for knobs in ({}, {"smooth": 0}, {"smooth": 0, "chirp_single": 0}, {"chirp_fuse_big": 0}):
        with _hackrf_engine(pkg, nfft, nf, hold_max=True, hold_min=True) as e:
            for k, v in knobs.items():
                e.debug_knob(k, v)
            out = e.process(feed, hop=hop, n_frames=nf)
            mx, mn = e.hold()
        _check(out, gold, f"N={nfft} {fmt} {knobs}")
        assert np.array_equal(mx, out.max(axis=0)) and np.array_equal(mn, out.min(axis=0))
        # hold traces only (no rows asked for): the same traces
        with _hackrf_engine(pkg, nfft, nf, hold_max=True, hold_min=True) as e:
            for k, v in knobs.items():
                e.debug_knob(k, v)
            assert e.process(feed, hop=hop, n_frames=nf, want_db=False) is None
            mx2, mn2 = e.hold()
        assert np.array_equal(mx2, mx) and np.array_equal(mn2, mn)
        # linear averaging over the three frames (the averager's scan on the power rows the transforms leave)
        ga = so.HackrfBranchOracle(nfft, 20e6, precision="gold")
        ga.averager.set_mode("lin", 3)
        gavg = np.stack([np.array(ga.power_levels(x[k * hop:k * hop + nfft])) for k in range(nf)])
        with _hackrf_engine(pkg, nfft, nf, avg=("lin", 3)) as e:
            for k, v in knobs.items():
                e.debug_knob(k, v)
            out = e.process(feed, hop=hop, n_frames=nf)
        _check(out, gavg, f"N={nfft} {fmt} {knobs} lin avg")


@pytest.mark.parametrize("nfft", [64, 1024, 4096, 16384])
def test_hackrf_plain_c64(pkg, nfft):
    nf = 3
    iq = so.synth_iq_int8(nfft * nf, nfft, seed=7 + nfft)
    x = so.unpack_iq_int8(iq)
    gold, _, _ = so.hackrf_batch(iq, nfft, nfft, 20e6, precision="gold", hold=False)
    with _hackrf_engine(pkg, nfft, nf) as e:
        out = e.process(x, hop=nfft)
    _check(out, gold, f"c64 N={nfft}")


def test_uint8_rtl_unpack_convention(pkg):
    """TDSA_IN_U8: x = u/127.5 - 1 (pyrtlsdr's packed_bytes_to_iq)."""
    nfft, nf = 2048, 3
    rng = np.random.default_rng(9)
    iq = rng.integers(0, 256, size=2 * nfft * nf, dtype=np.uint8)
    x = so.unpack_iq_uint8_rtl(iq)
    br = so.RtlBranchOracle(nfft, 2e6, "hanning", precision="gold")
    gold = np.stack([br.power_levels(x[k * nfft:(k + 1) * nfft]) for k in range(nf)])
    with pkg.SpectrumEngine(nfft, max_frames=nf) as e:
        e.set_window(so.rtl_window("hanning", nfft).astype(np.float32))
        e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=-1.0)
        out = e.process(iq, hop=nfft)
    _check(out, gold, "uint8")


# ------------------------------------------------------------------------------------------------
# golden vectors captured from the imported reference (tests/golden), batch API
# ------------------------------------------------------------------------------------------------
HACKRF_MODES = {
    "plain": dict(),
    "psd": dict(db_mode="pow", power_scale=None, log_floor=so.LOG_FLOOR),
    "exp4": dict(db_mode="pow", log_floor=so.POWER_LOG_FLOOR, avg=("exp", 4)),
    "lin3": dict(db_mode="pow", log_floor=so.POWER_LOG_FLOOR, avg=("lin", 3)),
    "psd_exp2": dict(db_mode="pow", power_scale=None, log_floor=so.LOG_FLOOR, avg=("exp", 2)),
    "dc_alpha_0p25": dict(dc_alpha=0.25),
}


@pytest.mark.parametrize("nfft", [1000, 1024, 4096, 16384])
@pytest.mark.parametrize("mode", sorted(HACKRF_MODES))
def test_hackrf_golden_batch(pkg, golden_dir, nfft, mode):
    g = np.load(os.path.join(golden_dir, f"hackrf_{nfft}.npz"))
    n, hop, nf, fs = int(g["nfft"]), int(g["hop"]), int(g["n_frames"]), float(g["sample_rate"])
    cfg = dict(HACKRF_MODES[mode])
    if "power_scale" in cfg and cfg["power_scale"] is None:
        cfg["power_scale"] = 1.0 / (fs * n)
    with pkg.SpectrumEngine(n, max_frames=nf) as e:
        e.set_window(g["window"])
        base = dict(db_mode="mag", log_floor=so.LOG_FLOOR, dc_alpha=1.0)
        base.update(cfg)
        e.configure(**base)
        out = e.process(g["iq_i8"], hop=hop)
        assert out.shape == (nf, n)
        _check(out, g[mode], f"golden hackrf_{n} {mode}")
        if mode == "dc_alpha_0p25":
            x = so.unpack_iq_int8(g["iq_i8"])
            br = so.HackrfBranchOracle(n, fs, dc_alpha=0.25, precision="gold")
            for k in range(nf):
                br.power_levels(so.frame(x, n, hop, k))
            assert abs(e.dc_estimate - complex(br.dc_estimate)) < 1e-6


RTL_MODES = {
    "hanning": dict(window="hanning"),
    "hamming": dict(window="hamming"),
    "rectangle": dict(window="rectangle"),
    "psd": dict(window="hanning", psd=True),
    "lin3": dict(window="hanning", avg=("lin", 3)),
    "exp4": dict(window="hanning", avg=("exp", 4)),
}


@pytest.mark.parametrize("nfft", [1024, 1500, 4096])
@pytest.mark.parametrize("mode", sorted(RTL_MODES))
def test_rtl_golden_batch(pkg, golden_dir, nfft, mode):
    g = np.load(os.path.join(golden_dir, f"rtl_{nfft}.npz"))
    n, nf, fs = int(g["nfft"]), int(g["n_frames"]), float(g["sample_rate"])
    m = RTL_MODES[mode]
    with pkg.SpectrumEngine(n, max_frames=nf) as e:
        e.set_window(so.rtl_window(m["window"], n).astype(np.float32))
        psd = m.get("psd", False)
        e.configure(db_mode="pow", power_scale=1.0 / (fs * n) if psd else 1.0,
                    log_floor=so.LOG_FLOOR if psd else so.POWER_LOG_FLOOR, dc_alpha=-1.0,
                    avg=m.get("avg", ("off", 1)))
        out = e.process(g["iq_i8"], hop=n)
    _check(out, g[mode], f"golden rtl_{n} {mode}")


# ------------------------------------------------------------------------------------------------
# the reference's own source classes' API, frame by frame (drop-in boundary)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nfft", [4096, 1000])                      # 1000: a size that is not a power of two
@pytest.mark.parametrize("mode", ["plain", "psd", "exp4", "lin3", "dc_alpha_0p25"])
def test_hackrf_source_class_golden(pkg, golden_dir, mode, nfft):
    g = np.load(os.path.join(golden_dir, f"hackrf_{nfft}.npz"))
    n, hop, nf = int(g["nfft"]), int(g["hop"]), int(g["n_frames"])
    x = so.unpack_iq_int8(g["iq_i8"])
    src = pkg.HackrfSamplesDataSource(sample_rate=int(g["sample_rate"]), centre_freq=int(g["centre_freq"]))
    src.num_samples = n
    src.running = True
    src._allocate_fft_resources()
    if mode == "psd":
        src.set_psd_mode(True)
    if mode == "exp4":
        src.set_averaging("exp", 4)
    if mode == "lin3":
        src.set_averaging("lin", 3)
    if mode == "dc_alpha_0p25":
        src.set_dc_alpha(0.25)
    for k in range(nf):
        src._reservoir = np.array(so.frame(x, n, hop, k), copy=True)
        p, fb = src.get_power_levels()
        assert p.dtype == g[mode].dtype                         # float64 when averaged, float32 plain
        _check(p, g[mode][k], f"{mode} frame {k}")
        assert np.array_equal(fb, g["freq_bins"])
        assert np.array_equal(src.get_raw_samples(), so.frame(x, n, hop, k))
    src.running = False


def test_hackrf_source_dc_estimate_across_resize_and_retune(pkg):
    """The tracked DC estimate belongs to the source: the reference keeps self._dc_estimate across set_num_samples
    (hackrf_samples.py:392-405).  Here a size change means a new plan: the estimate has to move with it.  A retune,
    on the other hand, ends in _flush_buffers in the reference too (its saved_dc is overwritten by the flush in
    _start_internal, :615): the estimate restarts from zero."""
    from topdogspectrumanalyser_amd.datasources.replay import ReplayHackRF
    iq = so.synth_iq_int8(1 << 18, 1024, seed=31)
    x = so.unpack_iq_int8(iq)
    src = pkg.HackrfSamplesDataSource(sample_rate=20_000_000, centre_freq=100_000_000)
    src.num_samples = 1024
    src.running = True
    src._allocate_fft_resources()
    src.set_dc_alpha(0.25)
    br = so.HackrfBranchOracle(1024, 20e6, dc_alpha=0.25, precision="gold")
    pos = 0
    for k in range(4):
        fr = x[pos: pos + 1024]
        pos += 1024
        src._reservoir = np.array(fr, copy=True)
        p, _ = src.get_power_levels()
        _check(p, br.power_levels(fr), f"1024 frame {k}")
    assert abs(src._dc_estimate - complex(br.dc_estimate)) < 1e-6
    src.set_num_samples(4096)                                   # new FFT size: a new plan on the GPU
    br2 = so.HackrfBranchOracle(4096, 20e6, dc_alpha=0.25, precision="gold")
    br2.dc_estimate = br.dc_estimate
    for k in range(3):
        fr = x[pos: pos + 4096]
        pos += 4096
        src._reservoir = np.array(fr, copy=True)
        p, _ = src.get_power_levels()
        _check(p, br2.power_levels(fr), f"4096 frame {k} after the size change")
    assert abs(src._dc_estimate - complex(br2.dc_estimate)) < 1e-6
    src.running = False
    # retune of a running source (injected replay device): buffers are flushed, the estimate is not
    live = pkg.HackrfSamplesDataSource(sample_rate=20_000_000, centre_freq=100_000_000,
                                       device_factory=lambda: ReplayHackRF(iq))
    live.set_dc_alpha(0.25)
    live.start()
    try:
        for _ in range(5):
            live.get_power_levels()
        assert abs(live._dc_estimate) > 1e-4
        live.update_centre_frequency(101_000_000)
        assert live._dc_estimate == 0j
    finally:
        live.stop()


@pytest.mark.parametrize("seed", range(int(os.environ.get("TDSA_SOURCE_CASES", "6"))))
def test_hackrf_source_random_knob_histories(pkg, seed):
    """Random histories of the HackRF source's DSP knobs between frames - FFT size, averaging mode / length / reset,
    PSD, DC alpha - against the float64 oracle of get_power_levels put through the same history."""
    rng = np.random.default_rng(6000 + seed)
    iq = so.synth_iq_int8(1 << 19, 1024, seed=int(rng.integers(1, 1 << 30)))
    x = so.unpack_iq_int8(iq)
    fs = 20_000_000
    n = 1024
    src = pkg.HackrfSamplesDataSource(sample_rate=fs, centre_freq=100_000_000)
    src.num_samples = n
    src.running = True
    src._allocate_fft_resources()
    br = so.HackrfBranchOracle(n, float(fs), dc_alpha=1.0, precision="gold")
    pos = 0
    for step in range(120):
        ev = rng.random()
        if ev < 0.08:
            new_n = int(rng.choice([256, 1024, 2048, 8192]))
            if new_n != n:
                n = new_n
                src.set_num_samples(n)
                nb = so.HackrfBranchOracle(n, float(fs), dc_alpha=br.dc_alpha, use_psd=br.use_psd, precision="gold")
                nb.dc_estimate = br.dc_estimate                  # survives the size change (hackrf_samples.py:392-405)
                nb.averager.set_mode(br.averager.mode, br.averager.n)   # set_num_samples resets the averager
                br = nb
        elif ev < 0.16:
            mode = [("off", 1), ("exp", int(rng.integers(2, 9))), ("lin", int(rng.integers(2, 12)))][int(rng.integers(0, 3))]
            src.set_averaging(*mode)
            br.averager.set_mode(*mode)
        elif ev < 0.20:
            src.reset_averaging()
            br.averager.reset()
        elif ev < 0.26:
            psd = bool(rng.integers(0, 2))
            src.set_psd_mode(psd)
            br.use_psd = psd
        elif ev < 0.32:
            a = float(rng.choice([1.0, 0.25, 0.05]))
            src.set_dc_alpha(a)
            br.dc_alpha = a
        if pos + n > len(x):
            pos = 0
        fr = x[pos: pos + n]
        pos += n
        src._reservoir = np.array(fr, copy=True)
        p, fb = src.get_power_levels()
        gold = np.asarray(br.power_levels(fr), dtype=np.float64)
        _check(p, gold, f"seed {seed} step {step}: n {n} avg {br.averager.mode},{br.averager.n} psd {br.use_psd} alpha {br.dc_alpha}")
        assert p.shape == (n,) and fb.shape == (n,)
    src.running = False


def test_hackrf_source_streaming_front_end(pkg):
    """start() with an injected device: reader thread + freshest-chunk framing + hold-last-good."""
    from topdogspectrumanalyser_amd.datasources.replay import ReplayHackRF
    n = 1024
    iq = so.synth_iq_int8(65536 * 2, n, seed=13)
    src = pkg.HackrfSamplesDataSource(sample_rate=20_000_000, centre_freq=2_450_000_000,
                                      device_factory=lambda: ReplayHackRF(iq))
    src.start()
    try:
        p, fb = src.get_power_levels()
        assert p.shape == (n,) and fb.shape == (n,) and np.isfinite(p).all()
        raw = src.get_raw_samples()
        gold = so.HackrfBranchOracle(n, 20e6, precision="gold").power_levels(raw)
        _check(p, gold, "streamed frame")
        st = src.get_stats()
        assert st["is_running"] and st["num_samples"] == n
    finally:
        src.stop()
    assert not src.is_running
    z, _ = src.get_power_levels()
    assert not z.any()


class _ScriptedHackRF:
    """Replay device whose stream can be switched to silence or to a stall (reader gets nothing)."""

    def __init__(self, iq):
        from topdogspectrumanalyser_amd.datasources.replay import ReplayHackRF
        self._inner = ReplayHackRF(iq)
        self.mode = "play"

    def read_samples(self, n):
        if self.mode == "stall":
            time.sleep(0.005)
            return np.array([], dtype=np.complex64)
        x = self._inner.read_samples(n)
        return np.zeros_like(x) if self.mode == "silent" else x

    def __getattr__(self, name):
        return getattr(self._inner, name)


def test_hackrf_source_concurrent_setters_and_reads(pkg):
    """The GUI thread turns knobs (FFT size incl. a long frame, averaging, PSD, DC alpha, retune) while the display
    timer keeps asking for frames and the reader thread keeps feeding: nothing raises, nothing hangs, every answer
    has the size that was current when it was produced."""
    import threading
    from topdogspectrumanalyser_amd.datasources.replay import ReplayHackRF
    iq = so.synth_iq_int8(1 << 18, 1024, seed=23)
    src = pkg.HackrfSamplesDataSource(sample_rate=20_000_000, centre_freq=100_000_000,
                                      device_factory=lambda: ReplayHackRF(iq))
    src.CONSUME_TIMEOUT = 0.2
    src.start()
    stop = threading.Event()
    problems, answers = [], [0]

    def timer():
        try:
            while not stop.is_set():
                p, fb = src.get_power_levels()
                if p.shape != fb.shape or p.ndim != 1:
                    problems.append(("shape", p.shape, fb.shape))
                answers[0] += 1
        except Exception as exc:                      # get_power_levels never raises
            problems.append(exc)

    th = threading.Thread(target=timer, daemon=True)
    th.start()
    rng = np.random.default_rng(5)
    t_end = time.time() + 2.0
    try:
        while time.time() < t_end:
            op = int(rng.integers(0, 6))
            if op == 0:
                src.set_num_samples(int(rng.choice([256, 1024, 4096, 16384, 32768])))
            elif op == 1:
                src.set_averaging(str(rng.choice(["off", "exp", "lin"])), int(rng.integers(1, 9)))
            elif op == 2:
                src.set_psd_mode(bool(rng.integers(0, 2)))
            elif op == 3:
                src.set_dc_alpha(float(rng.choice([1.0, 0.25])))
            elif op == 4:
                src.update_centre_frequency(int(100_000_000 + rng.integers(0, 5) * 1_000_000))
            else:
                src.reset_averaging()
            time.sleep(0.01)
    finally:
        stop.set()
        th.join(timeout=5.0)
        src.stop()
    assert not th.is_alive(), "display-timer thread hung"
    assert not problems, problems[:3]
    assert answers[0] > 10


def test_hackrf_source_holds_last_good_frame(pkg):
    """a3 (hackrf_samples.py:351-355): a silent frame and an underrun both return the PREVIOUS trace
    object; before any good frame they return zeros; a good frame afterwards replaces it."""
    n = 1024
    iq = so.synth_iq_int8(65536 * 2, n, seed=17)
    dev = _ScriptedHackRF(iq)
    src = pkg.HackrfSamplesDataSource(sample_rate=20_000_000, centre_freq=2_450_000_000,
                                      device_factory=lambda: dev)
    src.CONSUME_TIMEOUT = 0.1

    def switch(mode):
        dev.mode = mode
        time.sleep(0.05)                       # every chunk read from now on is of the new kind
        with src._lock:
            src._inbox.clear()

    dev.mode = "silent"
    src.start()
    try:
        switch("silent")
        z, fb = src.get_power_levels()         # silence before any good frame: zeros + axis
        assert z.shape == (n,) and not z.any() and fb.shape == (n,)
        switch("play")
        good, _ = src.get_power_levels()
        assert np.isfinite(good).all() and good.any()
        gold = so.HackrfBranchOracle(n, 20e6, precision="gold").power_levels(src.get_raw_samples())
        _check(good, gold, "good frame")
        keep = good.copy()
        switch("silent")
        held, fb2 = src.get_power_levels()
        assert held is good and np.array_equal(held, keep) and fb2 is fb
        switch("stall")
        t0 = time.time()
        held2, _ = src.get_power_levels()      # underrun: _consume_samples gives up -> None
        assert held2 is good and time.time() - t0 >= 0.09
        switch("play")
        fresh, _ = src.get_power_levels()
        assert fresh is not good and np.isfinite(fresh).all()
        assert src._last_good_power is fresh
    finally:
        src.stop()


@pytest.mark.parametrize("nfft", [1024, 1500])                      # 1500: a size that is not a power of two
@pytest.mark.parametrize("mode", ["hanning", "hamming", "rectangle", "psd", "lin3"])
def test_rtl_source_class_golden(pkg, golden_dir, mode, nfft):
    from topdogspectrumanalyser_amd.datasources.replay import ReplayRtlSdr
    g = np.load(os.path.join(golden_dir, f"rtl_{nfft}.npz"))
    n, nf, fs, fc = int(g["nfft"]), int(g["n_frames"]), float(g["sample_rate"]), float(g["centre_freq"])
    src = pkg.RtlSamplesDataSource(sample_rate=int(fs), centre_freq=int(fc),
                                   device_factory=lambda: ReplayRtlSdr(g["iq_i8"], fs, fc))
    src.start()
    src.set_fft_size(n)
    if mode in ("hanning", "hamming", "rectangle"):
        src.set_window_type(mode)
    if mode == "psd":
        src.set_psd_mode(True)
    if mode == "lin3":
        src.set_averaging("lin", 3)
    for k in range(nf):
        p, fb = src.get_power_levels()
        assert p.dtype == np.float64
        _check(p, g[mode][k], f"rtl {mode} frame {k}")
        assert np.array_equal(fb, g["freq_bins"])
    src.stop()


@pytest.mark.parametrize("seed", range(int(os.environ.get("TDSA_SOURCE_CASES", "6"))))
def test_rtl_source_random_knob_histories(pkg, seed):
    """Random histories of the RTL source's knobs between frames - FFT size (which silently reverts the window to
    Hanning, rtl_samples.py:213), window type, PSD, averaging - against the float64 oracle through the same history."""
    from topdogspectrumanalyser_amd.datasources.replay import ReplayRtlSdr
    rng = np.random.default_rng(6500 + seed)
    iq = so.synth_iq_int8(1 << 18, 1024, seed=int(rng.integers(1, 1 << 30)))
    x = so.unpack_iq_int8(iq)
    fs, fc = 2_048_000.0, 100e6
    src = pkg.RtlSamplesDataSource(sample_rate=int(fs), centre_freq=int(fc), device_factory=lambda: ReplayRtlSdr(iq, fs, fc))
    src.start()
    n = 1024
    src.set_fft_size(n)
    br = so.RtlBranchOracle(n, fs, "hanning", precision="gold")
    pos = 0
    try:
        for step in range(100):
            ev = rng.random()
            if ev < 0.08:
                new_n = int(rng.choice([256, 1024, 4096]))
                if new_n != n:
                    n = new_n
                    src.set_fft_size(n)
                    nb = so.RtlBranchOracle(n, fs, "hanning", use_psd=br.use_psd, precision="gold")   # (sic) Hanning again
                    nb.averager.set_mode(br.averager.mode, br.averager.n)
                    br = nb
            elif ev < 0.16:
                w = str(rng.choice(["hanning", "hamming", "rectangle"]))
                src.set_window_type(w)
                br.window = so.rtl_window(w, n)
            elif ev < 0.24:
                mode = [("off", 1), ("exp", int(rng.integers(2, 9))), ("lin", int(rng.integers(2, 12)))][int(rng.integers(0, 3))]
                src.set_averaging(*mode)
                br.averager.set_mode(*mode)
            elif ev < 0.30:
                psd = bool(rng.integers(0, 2))
                src.set_psd_mode(psd)
                br.use_psd = psd
            idx = (pos + np.arange(n)) % len(x)
            pos = (pos + n) % len(x)
            p, fb = src.get_power_levels()
            gold = np.asarray(br.power_levels(x[idx]), dtype=np.float64)
            _check(p, gold, f"seed {seed} step {step}: n {n} avg {br.averager.mode},{br.averager.n} psd {br.use_psd}")
            assert p.dtype == np.float64 and fb.shape == (n,)
    finally:
        src.stop()


# ------------------------------------------------------------------------------------------------
# DataProcessor: cal offset, 32-frame tare, max / min hold (golden sequence from the reference)
# ------------------------------------------------------------------------------------------------
class _NS:
    pass


class _Label:
    text = ""

    def setText(self, s):
        self.text = s


def _make_processor(pkg, src, cal_offset, max_on, min_on, **dp_options):
    mw, dm = _NS(), _NS()
    mw.current_source = src
    mw.calibration_manager = _NS()
    mw.calibration_manager.get_offset = lambda source_type: cal_offset
    mw.source_manager = _NS()
    mw.source_manager.last_source_type = "hackrf_samples"
    mw.status_label = _Label()
    mw.tare_active, mw.baseline_power_levels = False, None
    mw.live_power_levels = mw.max_power_levels = mw.min_power_levels = mw.frequency_bins = None
    mw.min_hold_enabled = min_on
    dm.tare_state = pkg.TareState()
    dm.max_peak_search_enabled = max_on
    dm.duty_cycle_enabled = dm.peak_list_enabled = False
    dm._update_tare_button_label = lambda s: None

    def _clear():
        mw.tare_active, mw.baseline_power_levels = False, None
    dm._clear_tare = _clear
    return pkg.DataProcessor(mw, dm, **dp_options), mw, dm


@pytest.mark.parametrize("holds", ["max", "min", "both", "both_as_reference"])
def test_data_processor_sequence_golden(pkg, golden_dir, holds):
    g = np.load(os.path.join(golden_dir, "processor_1024.npz"))
    n = int(g["nfft"])
    src = pkg.HackrfSamplesDataSource(sample_rate=int(g["sample_rate"]), centre_freq=int(g["centre_freq"]))
    src.num_samples = n
    src.running = True
    src._allocate_fft_resources()
    as_ref = holds == "both_as_reference"
    if as_ref:
        holds = "both"
    dp, mw, dm = _make_processor(pkg, src, float(g["cal_offset"]), holds in ("max", "both"),
                                 holds in ("min", "both"), reference_hold_alias=as_ref)
    # default: independent max / min traces - what the reference computes with ONE hold enabled, and what
    # the oracle computes with alias_quirk=False when both are.  reference_hold_alias=True: the reference's
    # own both-holds sequence (one shared ndarray: golden max_hold_both / min_hold_both), SURVEY 8(a) quirk ii
    ref_max = g["max_hold_both"] if as_ref else g["max_hold"]
    ref_min = g["min_hold_both"] if as_ref else g["min_hold"]
    for k, fr in enumerate(g["frames_c64"]):
        if k == int(g["tare_start"]):
            dm.tare_state = pkg.TareState(collecting=True)
        src._reservoir = np.array(fr, copy=True)
        dp._process_sample_data()
        live = mw.live_power_levels
        # both sides are float32 computations here (GPU vs the reference's own numpy float32 vectors), so
        # the allowance of parity_metrics applies twice: 1e-3 dB, or 2 x 2^-24 of the frame's largest
        # amplitude where that is worth more (depth judged before the tare subtraction)
        base = g["baseline"] if mw.tare_active else 0.0
        untared = g["live"][k] + base
        top = untared.max()

        def allowance(level_db):
            return np.maximum(DB_TOL, (20.0 / np.log(10.0)) * 2 * so.AMP_FLOOR * 10.0 ** ((top - level_db) / 20.0))

        d = np.abs(live - g["live"][k])
        assert np.all(d <= allowance(untared)), (k, float((d / allowance(untared)).max()))
        if holds in ("max", "both"):
            dm_ = np.abs(mw.max_power_levels - ref_max[k])
            assert np.all(dm_ <= allowance(ref_max[k] + base)), (k, float(dm_.max()))
        if holds in ("min", "both"):
            dn_ = np.abs(mw.min_power_levels - ref_min[k])
            assert np.all(dn_ <= allowance(ref_min[k] + base)), (k, float(dn_.max()))
        if as_ref:
            assert mw.max_power_levels is mw.min_power_levels        # one buffer, as in the reference
    assert mw.tare_active == bool(g["tare_active_at_end"])
    assert np.abs(mw.baseline_power_levels - g["baseline"]).max() < DB_TOL
    assert dm.tare_state.collecting is False
    src.running = False


class ProcessorModel:
    """float64 model of display_data_processor.py:317-395 of the reference: + calibration offset, tare collect /
    subtract (32 frames, cleared on a length change), fmax / fmin hold with the first frame adopted (NaN -> -500 /
    +500) and a disabled hold dropped when the trace length changes.  Validated against the imported reference in
    tests/test_reference_differential.py."""

    def __init__(self, alias_quirk=False):
        # alias_quirk: like the reference, a hold ADOPTS the frame object itself when it holds no NaN (_nan_safe
        # returns its argument), so two holds adopting on the same frame share one array and from then on both
        # just follow the live trace (SURVEY 8(a) quirk ii); False: independent running fmax / fmin (product default)
        self.alias_quirk = alias_quirk
        self.collect, self.buf, self.cnt, self.active, self.base = False, None, 0, False, None
        self.max = self.min = None

    def start_tare(self):
        self.collect, self.buf, self.cnt = True, None, 0

    def clear_tare(self):
        self.collect, self.buf, self.cnt, self.active, self.base = False, None, 0, False, None

    def frame(self, x, cal, max_on, min_on):
        lv = np.asarray(x, dtype=np.float64) + (cal if cal != 0.0 else 0.0)
        if self.collect:
            lin = 10.0 ** (lv / 10.0)
            if self.buf is None or self.buf.shape != lin.shape:
                self.buf, self.cnt = lin.copy(), 1
            else:
                self.buf += lin
                self.cnt += 1
            if self.cnt >= 32:
                self.base = 10.0 * np.log10(np.maximum(self.buf / self.cnt, 1e-30))
                self.active, self.collect, self.buf, self.cnt = True, False, None, 0
        if self.active and self.base is not None:
            if lv.shape != self.base.shape:
                self.active, self.base = False, None
            else:
                lv = lv - self.base
        for which, on in (("max", max_on), ("min", min_on)):
            cur = getattr(self, which)
            if not on:
                if cur is not None and cur.shape != lv.shape:
                    cur = None
            elif cur is None or cur.shape != lv.shape:
                if self.alias_quirk and not np.isnan(lv).any():
                    cur = lv                                  # the very object (and the other hold's, if it adopts too)
                else:
                    cur = np.where(np.isnan(lv), -500.0 if which == "max" else 500.0, lv)
            elif which == "max":
                np.fmax(cur, lv, out=cur)                     # in place, like the reference: an aliased partner sees it
            else:
                np.fmin(cur, lv, out=cur)
            setattr(self, which, cur)
        return lv


def processor_history(rng, ticks=150):
    """seeded random GUI history: yields (event, trace) per tick; events are ('max',), ('min',), ('tare',), ('clear',),
    ('cal', value) or None"""
    n = int(rng.choice([128, 256]))
    for _ in range(ticks):
        ev = rng.random()
        event = None
        if ev < 0.05:
            event = ("max",)
        elif ev < 0.10:
            event = ("min",)
        elif ev < 0.14:
            event = ("tare",)
        elif ev < 0.17:
            event = ("clear",)
        elif ev < 0.22:
            event = ("cal", float(rng.choice([0.0, -0.8087, 3.5])))
        elif ev < 0.25:
            n = int(rng.choice([128, 256, 512]))
        x = rng.normal(-70.0, 10.0, n).astype(np.float32)
        if rng.random() < 0.1:
            x[rng.integers(0, n)] = np.nan
        yield event, x


class _TraceSource:
    """stands in for a sample source: hands out prepared dB traces"""

    def __init__(self, pkg):
        self.base = pkg.SampleDataSource
        self.trace, self.axis = None, None
        self.last_data_time = 0.0

    def get_power_levels(self):
        return self.trace, self.axis


@pytest.mark.parametrize("as_reference", [False, True])
@pytest.mark.parametrize("seed", range(int(os.environ.get("TDSA_PROCESSOR_CASES", "6"))))
def test_data_processor_random_events(pkg, seed, as_reference):
    """Random GUI histories through DataProcessor._process_sample_data - holds switched on and off, tare runs
    started and cleared, calibration offset changes, trace length changes, NaN bins - against a float64 model of
    display_data_processor.py:317-395 of the reference (cal offset, tare, max / min hold)."""
    rng = np.random.default_rng(8000 + seed)
    src = _TraceSource(pkg)
    cal = {"v": 0.0}
    mw, dm = _NS(), _NS()
    mw.current_source = src
    mw.calibration_manager = _NS()
    mw.calibration_manager.get_offset = lambda source_type: cal["v"]
    mw.source_manager = _NS()
    mw.source_manager.last_source_type = "hackrf_samples"
    mw.status_label = _Label()
    mw.tare_active, mw.baseline_power_levels = False, None
    mw.live_power_levels = mw.max_power_levels = mw.min_power_levels = mw.frequency_bins = None
    mw.min_hold_enabled = bool(rng.integers(0, 2))
    dm.tare_state = pkg.TareState()
    dm.max_peak_search_enabled = bool(rng.integers(0, 2))
    dm.duty_cycle_enabled = dm.peak_list_enabled = False
    dm._update_tare_button_label = lambda s: None

    def _clear():
        mw.tare_active, mw.baseline_power_levels = False, None
        dm.tare_state = pkg.TareState()
    dm._clear_tare = _clear
    dp = pkg.DataProcessor(mw, dm, reference_hold_alias=as_reference)      # True: quirk ii reproduced on request

    model = ProcessorModel(alias_quirk=as_reference)
    for tick, (event, x) in enumerate(processor_history(rng)):
        if event == ("max",):
            dm.max_peak_search_enabled = not dm.max_peak_search_enabled
        elif event == ("min",):
            mw.min_hold_enabled = not mw.min_hold_enabled
        elif event == ("tare",):
            dm.tare_state = pkg.TareState(collecting=True)
            model.start_tare()
        elif event == ("clear",):
            dm._clear_tare()
            model.clear_tare()
        elif event is not None:
            cal["v"] = event[1]
        n = len(x)
        src.trace, src.axis = x, np.arange(n, dtype=np.float64)
        dp._process_sample_data()
        lv = model.frame(x, cal["v"], dm.max_peak_search_enabled, mw.min_hold_enabled)
        m_active, m_base, m_max, m_min = model.active, model.base, model.max, model.min
        what = f"seed {seed} tick {tick}"
        assert mw.tare_active == m_active, what
        dlive = np.abs(np.asarray(mw.live_power_levels, dtype=np.float64) - lv)
        bad = int(np.nanargmax(np.where(np.isnan(dlive), -1.0, dlive)))
        assert np.allclose(mw.live_power_levels, lv, rtol=0, atol=2e-4, equal_nan=True), \
            f"{what}: bin {bad} got {mw.live_power_levels[bad]} want {lv[bad]} x {x[bad]} base {None if m_base is None else m_base[bad]} nan mismatch {int((np.isnan(mw.live_power_levels) != np.isnan(lv)).sum())}"
        for got, want, name in ((mw.max_power_levels, m_max, "max"), (mw.min_power_levels, m_min, "min")):
            assert (got is None) == (want is None), f"{what}: {name} hold present {got is not None} vs {want is not None}"
            if want is not None:
                assert np.allclose(got, want, rtol=0, atol=2e-4, equal_nan=True), f"{what}: {name} hold"


def test_trace_averager_golden(pkg, golden_dir):
    g = np.load(os.path.join(golden_dir, "averager.npz"))
    for name, (mode, n) in {"off": ("off", 1), "exp8": ("exp", 8), "lin4": ("lin", 4), "lin64": ("lin", 64),
                            "exp1": ("exp", 1)}.items():
        av = pkg.TraceAverager()
        av.set_mode(mode, n)
        for k, f in enumerate(g["frames"]):
            out = av.process(f)
            assert np.allclose(out, g[name][k], rtol=1e-6, atol=0), (name, k)
        av.reset()
        assert np.allclose(av.process(g["frames"][3]), g["frames"][3])   # first frame after reset == input


def test_trace_averager_as_the_reference_smoke_checks_it(pkg):
    """The three TraceAverager checks of the reference's test_smoke.py:137-175 (passthrough when inactive, first
    frame == input and the second blends towards it, reset clears a NaN buffer), on the device-backed class.
    (tests/test_reference_own_tests.py runs the reference's file itself for everything that needs no GPU.)"""
    ta = pkg.TraceAverager()
    ones = np.ones(128, dtype=np.float64)
    assert ta.is_active is False
    assert np.allclose(ta.process(ones), ones)
    ta = pkg.TraceAverager()
    ta.set_mode("exp", 4)
    assert ta.is_active
    low = ta.process(np.ones(64) * 10.0).copy()[0]
    assert low == 10.0
    high = ta.process(np.ones(64) * 20.0).copy()[0]
    assert high > low and high == 10.0 * 0.75 + 20.0 / 4
    ta = pkg.TraceAverager()
    ta.set_mode("exp", 4)
    ta.process(np.full(64, np.nan))
    ta.reset()
    assert not np.any(np.isnan(ta.process(np.ones(64) * 5.0)))


@pytest.mark.parametrize("seed", range(int(os.environ.get("TDSA_AVERAGER_CASES", "8"))))
def test_trace_averager_random_histories(pkg, seed):
    """TraceAverager (host-array API of utils/signal_processing.py:5-73) through random histories - mode / length
    changes, resets, shape changes, 2-D inputs, huge and tiny values, NaN bins - against the reference's arithmetic
    restated in float64 numpy."""
    rng = np.random.default_rng(4400 + seed)
    av, ref = pkg.TraceAverager(), so.TraceAveragerOracle()
    shape = (256,)
    for step in range(80):
        ev = rng.random()
        if ev < 0.10:
            mode = [("off", 1), ("exp", int(rng.integers(1, 10))), ("lin", int(rng.integers(1, 12)))][int(rng.integers(0, 3))]
            av.set_mode(*mode)
            ref.set_mode(*mode)
        elif ev < 0.15:
            av.reset()
            ref.reset()
        elif ev < 0.20:
            shape = [(256,), (1000,), (3, 64), (1,)][int(rng.integers(0, 4))]
        x = (10.0 ** rng.uniform(-12, 3, size=shape)).astype(np.float32)
        if rng.random() < 0.05:
            x.flat[int(rng.integers(0, x.size))] = np.nan
        got = av.process(x)
        want = ref.process(x)
        assert np.shape(got) == np.shape(want), (seed, step)
        assert av.is_active == ref.is_active
        assert np.allclose(got, want, rtol=1e-6, atol=0, equal_nan=True), (seed, step, ref.mode, ref.n)


# ------------------------------------------------------------------------------------------------
# size-independent properties at BASELINE.json's full shapes
# ------------------------------------------------------------------------------------------------
def test_c3_full_second_properties(pkg):
    """C3: N=16384, hop=N/2, 20e6 samples -> 2440 frames, HackRF branch + peak hold."""
    nfft, hop, ns = 16384, 8192, 20_000_000
    nf = (ns - nfft) // hop + 1
    assert nf == 2440
    iq = so.synth_iq_int8(ns, nfft, seed=3)
    with _hackrf_engine(pkg, nfft, nf, hold_max=True, hold_min=True) as e:
        out = e.process(iq, hop=hop)
        mx, mn = e.hold()
        # (a) hold traces are exactly the column max / min of what was written
        assert np.array_equal(mx, out.max(axis=0)) and np.array_equal(mn, out.min(axis=0))
        # (b) frames are independent: any frame processed alone is bit-identical to its row in the batch
        e.reset()
        for k in (0, 1, 1219, 2439):
            alone = e.process(iq[2 * k * hop: 2 * (k * hop + nfft)], hop=hop, n_frames=1)
            assert np.array_equal(alone[0], out[k]), k
    # (c) sampled rows against the gold oracle
    x = so.unpack_iq_int8(iq)
    br = so.HackrfBranchOracle(nfft, 20e6, precision="gold")
    for k in (0, 7, 1219, 2439):
        _check(out[k], br.power_levels(so.frame(x, nfft, hop, k)), f"C3 frame {k}")
    # (d) the three synthetic tones sit where SURVEY.md 8(d) puts them (fftshift-ed bins)
    peak = int(np.argmax(out[5]))
    assert peak == nfft // 2 + nfft // 8


def test_c4_shape_independence_and_parseval(pkg):
    """C4 shape (N=8192, hop=N) on a slice of the waterfall: Parseval + batch == per-frame."""
    nfft, nf = 8192, 1024
    iq = so.synth_iq_int8(nfft * nf, nfft, seed=4)
    w = so.hackrf_window(nfft)
    with pkg.SpectrumEngine(nfft, max_frames=nf) as e:
        e.set_window(w)
        e.configure(db_mode="pow", power_scale=1.0, log_floor=0.0, dc_alpha=1.0)
        out = e.process(iq, hop=nfft)
        k = 517
        alone = e.process(iq[2 * k * nfft: 2 * (k + 1) * nfft], hop=nfft, n_frames=1)
        assert np.array_equal(alone[0], out[k])
    x = so.unpack_iq_int8(iq).astype(np.complex128).reshape(nf, nfft)
    xw = (x - x.mean(axis=1, keepdims=True)) * w.astype(np.float64)
    energy_time = nfft * (np.abs(xw) ** 2).sum(axis=1)               # sum_k |X_k|^2 = N sum_n |x_n|^2
    energy_freq = (10.0 ** (out.astype(np.float64) / 10.0)).sum(axis=1)
    assert np.max(np.abs(energy_freq / energy_time - 1.0)) < 1e-5


_C4_SHARE = {}


def _c4_share_iq():
    """the 8192 frames x 8192 points one GPU of eight gets of BASELINE config 4 (synthesised once per session: 25 s)"""
    if "iq" not in _C4_SHARE:
        nfft, chunk = 8192, 1024
        _C4_SHARE["iq"] = np.concatenate([so.synth_iq_int8(nfft * chunk, nfft, seed=40 + c) for c in range(8)])
    return _C4_SHARE["iq"]


def test_c4_full_waterfall_properties(pkg):
    """BASELINE config 4 as named, on ONE GPU: the whole waterfall of 65 536 frames x 8192 points in one call (1 GiB of int8
    IQ in, 2 GiB of dB rows out: what bench.py --config c4 times per step; displays/waterfall.py:163-180 is the layout the
    rows have).  The capture is the 8192-frame share eight times, each copy rotated by another odd number of samples (as
    bench.py builds it).  Max hold == column maximum of all 65 536 rows bit for bit, Parseval on the frames of every
    eighth block of 1024, frames at the copies' seams and ends against the float64 gold and against the same frame
    processed alone."""
    nfft, nf, chunk = 8192, 65536, 1024
    share = _c4_share_iq()
    iq = np.concatenate([share if k == 0 else np.roll(share, 2 * 977 * k) for k in range(8)])
    assert iq.size == 2 * nfft * nf
    w = so.hackrf_window(nfft)
    with pkg.SpectrumEngine(nfft, max_frames=nf) as e:
        e.set_window(w)
        e.configure(db_mode="pow", power_scale=1.0, log_floor=0.0, dc_alpha=1.0, hold_max=True)
        out = e.process(iq, hop=nfft)
        mx, _ = e.hold()
        assert out.shape == (nf, nfft)
        assert np.array_equal(mx, out.max(axis=0))
        e.reset()
        picks = (0, 8191, 8192, 8193, 30000, 57343, 57344, nf - 1)
        for k in picks:
            alone = e.process(iq[2 * k * nfft: 2 * (k + 1) * nfft], hop=nfft, n_frames=1)
            assert np.array_equal(alone[0], out[k]), k
    w64 = w.astype(np.float64)
    worst = 0.0
    for c in range(0, nf // chunk, 8):                # sum_k |X_k|^2 = N sum_n |w_n (x_n - mean)|^2, frame by frame
        x = so.unpack_iq_int8(iq[2 * c * chunk * nfft: 2 * (c + 1) * chunk * nfft]).astype(np.complex128).reshape(chunk, nfft)
        xw = (x - x.mean(axis=1, keepdims=True)) * w64
        energy_time = nfft * (xw.real ** 2 + xw.imag ** 2).sum(axis=1)
        energy_freq = (10.0 ** (out[c * chunk:(c + 1) * chunk].astype(np.float64) / 10.0)).sum(axis=1)
        worst = max(worst, float(np.max(np.abs(energy_freq / energy_time - 1.0))))
    assert worst < 1e-5, worst
    br = so.HackrfBranchOracle(nfft, 20e6, precision="gold")
    for k in picks:
        gold = br.power_levels(so.unpack_iq_int8(iq[2 * k * nfft: 2 * (k + 1) * nfft]))
        _check(out[k], gold, f"C4 waterfall frame {k}")


def test_c4_full_share_properties(pkg):
    """C4 at the size one GPU of eight gets: 64k frames / 8 GPUs = 8192 frames x 8192 points (hop = N), one launch.
    Parseval on every frame, frame independence, max hold == column max, sampled rows against the gold oracle."""
    nfft, nf, chunk = 8192, 8192, 1024
    iq = _c4_share_iq()
    w = so.hackrf_window(nfft)
    with pkg.SpectrumEngine(nfft, max_frames=nf) as e:
        e.set_window(w)
        e.configure(db_mode="pow", power_scale=1.0, log_floor=0.0, dc_alpha=1.0, hold_max=True)
        out = e.process(iq, hop=nfft)
        mx, _ = e.hold()
        assert out.shape == (nf, nfft)
        assert np.array_equal(mx, out.max(axis=0))
        e.reset()
        for k in (0, 1023, 1024, 4777, nf - 1):
            alone = e.process(iq[2 * k * nfft: 2 * (k + 1) * nfft], hop=nfft, n_frames=1)
            assert np.array_equal(alone[0], out[k]), k
    w64 = w.astype(np.float64)
    worst = 0.0
    for c in range(nf // chunk):                      # sum_k |X_k|^2 = N sum_n |w_n (x_n - mean)|^2, frame by frame
        x = so.unpack_iq_int8(iq[2 * c * chunk * nfft: 2 * (c + 1) * chunk * nfft]).astype(np.complex128).reshape(chunk, nfft)
        xw = (x - x.mean(axis=1, keepdims=True)) * w64
        energy_time = nfft * (xw.real ** 2 + xw.imag ** 2).sum(axis=1)
        energy_freq = (10.0 ** (out[c * chunk:(c + 1) * chunk].astype(np.float64) / 10.0)).sum(axis=1)
        worst = max(worst, float(np.max(np.abs(energy_freq / energy_time - 1.0))))
    assert worst < 1e-5, worst
    br = so.HackrfBranchOracle(nfft, 20e6, precision="gold")
    for k in (0, 2048, 5000, nf - 1):
        x = so.unpack_iq_int8(iq[2 * k * nfft: 2 * (k + 1) * nfft])
        gold = br.power_levels(x)                     # 20 log10(|X| + 1e-12) == 10 log10 |X|^2 at these levels
        _check(out[k], gold, f"C4 frame {k}")


def test_linearity_in_db(pkg):
    """Scaling the input by 2 moves every dB value by 20*log10(2) (DC removal and window are linear)."""
    nfft, nf = 4096, 4
    x = so.unpack_iq_int8(so.synth_iq_int8(nfft * nf, nfft, seed=21))
    with _hackrf_engine(pkg, nfft, nf) as e:
        a = e.process(x, hop=nfft)
        b = e.process((2 * x).astype(np.complex64), hop=nfft)
    assert np.max(np.abs((b - a) - 20 * np.log10(2.0))) < 2e-4


# ------------------------------------------------------------------------------------------------
# edge cases and error conventions
# ------------------------------------------------------------------------------------------------
def test_silence_hits_the_log_floor(pkg):
    """All-zero IQ: 20*log10(0 + 1e-12) = -240 dB (HackRF branch), 10*log10(0 + 1e-10) = -100 dB (RTL)."""
    nfft = 1024
    z = np.zeros(2 * nfft * 2, dtype=np.int8)
    with _hackrf_engine(pkg, nfft, 2) as e:
        out = e.process(z, hop=nfft)
    assert np.allclose(out, -240.0, atol=1e-3)
    with pkg.SpectrumEngine(nfft, max_frames=2) as e:
        e.set_window(np.hanning(nfft).astype(np.float32))
        e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=-1.0)
        out = e.process(z, hop=nfft)
    assert np.allclose(out, -100.0, atol=1e-3)


def test_weak_signal_uses_exact_mag_path(pkg):
    """|X| around 1e-6: the 1e-12 floor is still invisible but the exact DB_MAG branch runs."""
    nfft, nf = 1024, 2
    rng = np.random.default_rng(2)
    x = (1e-8 * (rng.standard_normal(nfft * nf) + 1j * rng.standard_normal(nfft * nf))).astype(np.complex64)
    gold = np.stack([so.HackrfBranchOracle(nfft, 20e6, precision="gold").power_levels(x[k * nfft:(k + 1) * nfft])
                     for k in range(nf)])
    with _hackrf_engine(pkg, nfft, nf) as e:
        out = e.process(x, hop=nfft)
    assert np.max(np.abs(out - gold)) < 5e-3


@pytest.mark.parametrize("hop", [1, 3, 1001, 2047, 4096, 5000])
def test_ragged_and_unaligned_hops(pkg, hop):
    """Frame starts are only sample (2-byte) aligned; trailing samples that do not fill a frame are ignored."""
    nfft = 4096
    ns = nfft + 6 * hop + 123
    iq = so.synth_iq_int8(ns, nfft, seed=hop)
    nf = so.num_frames(ns, nfft, hop)
    gold, _, _ = so.hackrf_batch(iq, nfft, hop, 20e6, precision="gold", hold=False)
    with _hackrf_engine(pkg, nfft, nf) as e:
        out = e.process(iq, hop=hop)
    assert out.shape == (nf, nfft)
    _check(out, gold, f"hop={hop}")


def test_empty_and_error_paths(pkg):
    from topdogspectrumanalyser_amd import _native as nat
    with _hackrf_engine(pkg, 1024, 4) as e:
        assert e.process(np.zeros(0, dtype=np.int8)).shape == (0, 1024)         # empty input
        assert e.process(np.zeros(2 * 1000, dtype=np.int8)).shape == (0, 1024)  # shorter than one frame
        assert e.hold() == (None, None)                                          # nothing held yet
        with pytest.raises(nat.TdsaError):
            e.process(np.zeros(2 * 1024 * 5, dtype=np.int8))                     # more frames than capacity
        with pytest.raises(nat.TdsaError):
            e.set_window(np.ones(512, dtype=np.float32))
        with pytest.raises(TypeError):
            e.process(np.zeros(2048, dtype=np.float64))
    with pytest.raises(nat.TdsaError):
        pkg.SpectrumEngine(1500000)                                              # above 2^20
    with pytest.raises(nat.TdsaError):
        pkg.SpectrumEngine(1 << 21)                                              # beyond the largest plan (2^20)
    with pytest.raises(nat.TdsaError):
        pkg.SpectrumEngine(1)                                                    # below the smallest (2)
    with pytest.raises(nat.TdsaError):
        pkg.SpectrumEngine(0)
    e = pkg.SpectrumEngine(1024)
    with pytest.raises(nat.TdsaError):
        e.process(np.zeros(2048, dtype=np.int8))                                 # window never set
    e.close()


def test_hold_and_averager_state_across_calls(pkg):
    """State persists across calls like mw.max_power_levels / TraceAverager._buffer, and resets clear it."""
    nfft, hop = 2048, 2048
    iq = so.synth_iq_int8(nfft * 8, nfft, seed=33)
    gold, gmax, _ = so.hackrf_batch(iq, nfft, hop, 20e6, precision="gold", avg=("lin", 5))
    with pkg.SpectrumEngine(nfft, max_frames=8) as e:
        e.set_window(so.hackrf_window(nfft))
        e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=1.0,
                    avg=("lin", 5), hold_max=True)
        a = e.process(iq[:2 * nfft * 3], hop=hop)
        b = e.process(iq[2 * nfft * 3:], hop=hop)
        out = np.concatenate([a, b])
        _check(out, gold, "split batch, lin avg")
        mx, _ = e.hold()
        _check(mx, gmax, "hold across calls", rows_gold=gold)
        buf, cnt = e.averaged()
        assert cnt == 5
        _check(10 * np.log10(buf + so.POWER_LOG_FLOOR), gold[-1], "averager state read-back")
        e.reset()
        assert e.hold() == (None, None) and e.averaged()[1] == 0
        again = e.process(iq[:2 * nfft], hop=hop)
        _check(again, gold[:1], "after reset")


def test_tare_baseline_and_cal_offset_in_batch(pkg):
    nfft, nf = 1024, 6
    iq = so.synth_iq_int8(nfft * nf, nfft, seed=44)
    gold, _, _ = so.hackrf_batch(iq, nfft, nfft, 20e6, precision="gold", cal_offset_db=-0.8087, hold=False)
    base = np.linspace(-3, 3, nfft).astype(np.float32)
    with _hackrf_engine(pkg, nfft, nf, cal_offset_db=-0.8087, hold_max=True) as e:
        e.set_tare_baseline(base)
        out = e.process(iq, hop=nfft)
        mx, _ = e.hold()
    _check(out + base, gold, "tare baseline + cal offset in the batch kernel")
    assert np.array_equal(mx, out.max(axis=0))


def test_c64_nan_first_frame_hold_semantics(pkg):
    """_nan_safe: a NaN first frame seeds the hold with -500 / +500; later NaNs are ignored (np.fmax)."""
    nfft = 256
    x = so.unpack_iq_int8(so.synth_iq_int8(nfft * 3, nfft, seed=5)).copy()
    x[5] = np.nan
    with _hackrf_engine(pkg, nfft, 3, hold_max=True, hold_min=True) as e:
        out = e.process(x, hop=nfft)
        mx, mn = e.hold()
    assert np.isnan(out[0]).all() and np.isfinite(out[1:]).all()
    assert np.array_equal(mx, np.fmax(np.fmax(np.full(nfft, -500.0, np.float32), out[1]), out[2]))
    assert np.array_equal(mn, np.fmin(np.fmin(np.full(nfft, 500.0, np.float32), out[1]), out[2]))


# ------------------------------------------------------------------------------------------------
# C5: 2^20-point FFT, Welch averaging over K segments, calibration offset (four-step kernels)
# ------------------------------------------------------------------------------------------------
CAL = -0.8087054556396822       # calibration.json:3 of the reference (rtl_samples), BASELINE config 5's offset


def _welch_gold(iq, nfft, k, cal, window="hanning", dc=False):
    x = so.unpack_iq_int8(iq).astype(np.complex128)
    w = so.rtl_window(window, nfft)
    acc = np.zeros(nfft)
    for s in range(k):
        seg = x[s * nfft:(s + 1) * nfft]
        if dc:
            seg = seg - seg.mean()
        acc += np.abs(np.fft.fftshift(np.fft.fft(seg * w))) ** 2
    return 10 * np.log10(acc / k + so.POWER_LOG_FLOOR) + cal, acc / k


def test_c5_million_point_welch(pkg):
    nfft, k, cal = 1 << 20, 4, -0.8087054556396822
    iq = so.synth_iq_int8(nfft * k, nfft, seed=5)
    gold, gold_mean = _welch_gold(iq, nfft, k, cal)
    with pkg.SpectrumEngine(nfft, max_frames=k) as e:
        e.set_window(so.rtl_window("hanning", nfft).astype(np.float32))
        e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=-1.0,
                    avg=("lin", k), cal_offset_db=cal, hold_max=True)
        out = e.process(iq, hop=nfft)
        assert out.shape == (1, nfft)
        _check(out[0], gold, "C5 one call")
        mean, cnt = e.averaged()
        assert cnt == k and np.max(np.abs(mean - gold_mean) / gold_mean.max()) < 1e-5
        mx, _ = e.hold()
        assert np.array_equal(mx, out[0])
        # the same Welch average fed in two calls (state persists like TraceAverager._buffer)
        e.reset()
        e.process(iq[:2 * nfft * 1], hop=nfft)
        out2 = e.process(iq[2 * nfft * 1:], hop=nfft)
        assert np.max(np.abs(out2[0] - out[0])) < 1e-4
        # peak where SURVEY.md 8(d) puts the strongest tone
        assert int(np.argmax(out[0])) == nfft // 2 + nfft // 8


def test_c5_reference_fixture(pkg, golden_dir):
    """C5 against vectors from the imported reference (tests/golden/c5_million.npz): 2^20 points, RTL branch,
    Welch mean of 8 segments + the calibration offset of calibration.json; comb of every 257th bin and the 64
    strongest bins, after the first segment and after the eighth."""
    g = np.load(os.path.join(golden_dir, "c5_million.npz"))
    n, k, cal = int(g["nfft"]), int(g["k"]), -0.8087054556396822
    iq = so.synth_iq_int8(n * k, n, seed=int(g["seed"]))
    with pkg.SpectrumEngine(n, max_frames=k) as e:
        e.set_window(so.rtl_window("hanning", n).astype(np.float32))
        e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=-1.0,
                    avg=("lin", k), cal_offset_db=cal)
        first = e.process(iq[: 2 * n], hop=n)[0]
        mean = e.process(iq[2 * n:], hop=n)[0]               # the other seven: state carried like TraceAverager
    for tag, got in (("first", first), ("mean", mean)):
        for kind in ("comb", "top"):
            bins, want = g[f"{tag}_{kind}_bins"], g[f"{tag}_{kind}_db"] + cal
            top = g[f"{tag}_top_db"].max() + cal
            allow = np.maximum(DB_TOL, (20.0 / np.log(10.0)) * 2 * so.AMP_FLOOR * 10.0 ** ((top - want) / 20.0))
            d = np.abs(got[bins] - want)
            assert np.all(d[want >= top - 100.0] <= allow[want >= top - 100.0]), (tag, kind, float(d.max()))
        assert np.array_equal(np.sort(np.argsort(got)[-64:]), g[f"{tag}_top_bins"]) or tag == "first"
    assert int(np.argmax(mean)) == n // 2 + n // 8


def test_c5_full_size_properties(pkg):
    """K = 64 segments of 2^20 points (the BASELINE.json size) through properties that do not need a CPU FFT of
    the whole batch: Parseval against the time-domain energy, state carried across calls, tone position, hold."""
    n, k = 1 << 20, 64
    iq = so.synth_iq_int8(n * k, n, seed=11)
    w = so.rtl_window("hanning", n)
    with pkg.SpectrumEngine(n, max_frames=k) as e:
        e.set_window(w.astype(np.float32))
        e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=-1.0, avg=("lin", k),
                    hold_max=True)
        one = e.process(iq, hop=n)[0]
        mean, cnt = e.averaged()
        mx, _ = e.hold()
        assert cnt == k and np.array_equal(mx, one)
        e.reset()
        e.process(iq[: 2 * n * 24], hop=n)                                   # 24 + 40 segments, two calls
        two = e.process(iq[2 * n * 24:], hop=n)[0]
        assert e.averaged()[1] == k
    assert np.max(np.abs(two - one)) < 1e-4
    assert int(np.argmax(one)) == n // 2 + n // 8
    # Parseval: sum_k |X[k]|^2 = N sum_n |w x|^2, averaged over the segments
    x = so.unpack_iq_int8(iq).astype(np.complex128).reshape(k, n)
    energy = float(np.mean(np.sum(np.abs(x * w) ** 2, axis=1))) * n
    assert abs(float(mean.sum()) - energy) <= 2e-6 * energy
    # the strongest tone (40 LSB at bin N/8 exactly): |X|^2 = (A/128 * sum(w))^2 plus noise far below
    peak = 10 * np.log10((40.0 / 128.0 * w.sum()) ** 2)
    assert abs(one[n // 2 + n // 8] - peak) < 0.05


def test_c5_welch_is_reproducible_bit_for_bit(pkg):
    """Six runs of the same 16 segments of 2^20 points (1024 column-pass workgroups in flight: more than the 256 it took
    to expose the gfx950 store-data hazard of the column pass's 16-byte row stores, tdsa_big.hip - its padding depends on
    the compiler keeping the row offset in the VGPR): the Welch row and the float64 mean must come out identical bit for
    bit every time (per-workgroup partial rows summed in a fixed order; a hazard shows as run-to-run differences of
    1e-6 .. 6e-4 of the frame maximum) and right."""
    n, k = 1 << 20, 16
    iq = so.synth_iq_int8(n * k, n, seed=19)
    rows, means = [], []
    with pkg.SpectrumEngine(n, max_frames=k) as e:
        e.set_window(so.rtl_window("hanning", n).astype(np.float32))
        e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=-1.0, avg=("lin", k))
        for _ in range(6):
            e.reset()
            rows.append(e.process(iq, hop=n)[0].copy())
            means.append(e.averaged()[0].copy())
    for r, m in zip(rows[1:], means[1:]):
        assert np.array_equal(r, rows[0]) and np.array_equal(m, means[0])
    gold, _ = _welch_gold(iq, n, k, 0.0)
    _check(rows[0], gold, "Welch of 16 segments of 2^20 points")


@pytest.mark.parametrize("log2n,k", [(15, 5), (17, 12), (20, 8), (20, 64)])
def test_c5_row_pass_and_gather_as_one_launch(pkg, log2n, k):
    """tdsa_debug_knob big_fuse_gather 1: the row pass of a Welch capture and gather + finish run as ONE launch whose
    workgroups draw tickets from a dependency-counted queue (tdsa_big.hip) - the same partial rows summed in the same order:
    the dB row, the float64 state and the hold trace must be the bits of the three-launch chain, call after call (the
    queue's counters only grow), also with the state carried across calls; no workgroup ever gave up waiting."""
    n = 1 << log2n
    iq = so.synth_iq_int8(n * k, min(n, 1 << 16), seed=log2n + k)
    outs = []
    for knob in (0, 1):
        with pkg.SpectrumEngine(n, max_frames=k) as e:
            e.set_window(so.rtl_window("hanning", n).astype(np.float32))
            e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=-1.0, avg=("lin", 3 * k),
                        cal_offset_db=-0.8087, hold_max=True)
            e.debug_knob("big_fuse_gather", knob)
            got = []
            for rep in range(3):                               # the averager's state is carried: 3 k segments in all
                got.append(e.process(iq, hop=n)[0].copy())
            got.append(e.averaged()[0].copy())
            got.append(e.hold()[0].copy())
            e.reset()
            got.append(e.process(iq, hop=n)[0].copy())         # and from a fresh state again
            e.debug_knob("big_queue_gave_up", 0)               # raises if a workgroup of a fused launch ever gave up
            outs.append(got)
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_c5_fused_launches_of_two_plans_share_the_gpu(pkg):
    """Two plans on their own streams, both with the fused row + gather launch, captures queued alternately without a
    synchronize in between: a device-wide barrier would deadlock here (each launch's waiting workgroups holding the CUs
    the other's unstarted ones need); with the ticket queue a waiting workgroup only waits for workgroups that are running."""
    import ctypes as C
    nat = pkg._native
    n, k = 1 << 20, 16
    iq = so.synth_iq_int8(n * k, 1 << 16, seed=77)
    d_in = C.c_void_p()
    nat.check(nat.lib.tdsa_dev_alloc(0, iq.nbytes, C.byref(d_in)))
    try:
        nat.check(nat.lib.tdsa_memcpy_h2d(0, d_in, iq.ctypes.data_as(C.c_void_p), iq.nbytes))
        plans = []
        for knob in (1, 1, 0):
            e = pkg.SpectrumEngine(n, max_frames=k)
            e.set_window(so.rtl_window("hanning", n).astype(np.float32))
            e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=-1.0, avg=("lin", k))
            e.debug_knob("big_fuse_gather", knob)
            plans.append(e)
        for rep in range(40):
            for e in plans:
                e.reset(nat.RESET_AVG)
                e.process_device(nat.IN_I8, d_in.value, n * k, n, k, None)
        means = []
        for e in plans:
            e.synchronize()
            e.debug_knob("big_queue_gave_up", 0)
            means.append(e.averaged()[0].copy())
            e.close()
        assert np.array_equal(means[0], means[2]) and np.array_equal(means[1], means[2])
    finally:
        nat.lib.tdsa_dev_free(0, d_in)


def test_c5_single_frame_with_dc_removal(pkg):
    nfft = 1 << 20
    iq = so.synth_iq_int8(nfft, nfft, seed=6)
    gold, _ = _welch_gold(iq, nfft, 1, 0.0, dc=True)
    with pkg.SpectrumEngine(nfft, max_frames=1) as e:
        e.set_window(so.rtl_window("hanning", nfft).astype(np.float32))
        e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=1.0)
        out = e.process(iq, hop=nfft)
    _check(out[0], gold, "C5 single frame, DC removed")


def test_c5_sharded_welch_combines_on_host(pkg):
    """each GPU averages its share of the segments; the host merges mean + count."""
    from topdogspectrumanalyser_amd import sharding
    nfft, k = 1 << 20, 4
    iq = so.synth_iq_int8(nfft * k, nfft, seed=8)
    gold, gold_mean = _welch_gold(iq, nfft, k, 0.0)
    means, counts = [], []
    for rank in range(2):                              # two "GPUs" = two plans on the one device here
        f0, f1 = sharding.shard_frames(k, rank, 2)
        s0, s1 = sharding.shard_samples(f0, f1, nfft, nfft)
        with pkg.SpectrumEngine(nfft, max_frames=f1 - f0) as e:
            e.set_window(so.rtl_window("hanning", nfft).astype(np.float32))
            e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=-1.0,
                        avg=("lin", k))
            e.process(iq[2 * s0:2 * s1], hop=nfft, want_db=False)
            m, c = e.averaged()
            means.append(m)
            counts.append(c)
    mean, total = sharding.combine_welch(means, counts)
    assert total == k
    _check(10 * np.log10(mean + so.POWER_LOG_FLOOR), gold, "sharded Welch")


@pytest.mark.parametrize("nfft,k,dtype", [(1 << 20, 5, np.float32), (1 << 20, 5, np.float64), (1 << 16, 7, np.float32),
                                          (4096, 300, np.float32), (4096, 300, np.float64)])
def test_welch_partials_combined_on_one_device(pkg, nfft, k, dtype):
    """tdsa_welch_export / tdsa_welch_combine (what bench.py --config c5 --gpus N does inside every timed
    step): three plans average unequal shares of a capture's segments (one of them none at all), hand out their running
    means as float32 / float64, and a fourth plan reassembles the overall mean ON ITS DEVICE.  The dB row must match the
    float64 gold of the whole capture, float64 partials reproduce the host combine (sharding.combine_welch) to 1e-15 and, like float32 ones, the row of ONE
    plan that saw every segment to 1e-5 dB (the plans group their float32 power sums differently);
    afterwards the combining plan carries the state: one more segment continues the running mean
    (utils/signal_processing.py:56-59)."""
    from topdogspectrumanalyser_amd import sharding
    iq = so.synth_iq_int8(nfft * (k + 1), nfft, seed=11)
    gold, _ = _welch_gold(iq[: 2 * nfft * k], nfft, k, CAL)
    gold_next, _ = _welch_gold(iq, nfft, k + 1, CAL)

    def plan(frames):
        e = pkg.SpectrumEngine(nfft, max_frames=max(1, frames))
        e.set_window(so.rtl_window("hanning", nfft).astype(np.float32))
        e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=-1.0, avg=("lin", k + 1),
                    cal_offset_db=CAL)
        return e
    shares = [(0, k // 2), (k // 2, k // 2), (k // 2, k)]      # the middle "GPU" gets nothing
    parts = np.zeros((len(shares), nfft + 16), dtype=dtype)[:, :nfft]   # strided rows, as in a shared slab
    counts = []
    for r, (f0, f1) in enumerate(shares):
        with plan(f1 - f0) as e:
            if f1 > f0:
                e.process(iq[2 * f0 * nfft: 2 * f1 * nfft], hop=nfft, want_db=False)
            counts.append(e.welch_export(parts[r]))
    assert counts == [f1 - f0 for f0, f1 in shares]
    # the same partials left in device buffers (tdsa_peer_alloc) and combined in place: the same bits as through the host
    import ctypes as C
    nat = pkg._native
    bufs, handle = [], (C.c_ubyte * 64)()
    try:
        for r, (f0, f1) in enumerate(shares):
            ptr = C.c_void_p()
            nat.check(nat.lib.tdsa_peer_alloc(0, nfft * np.dtype(dtype).itemsize, C.byref(ptr), handle))
            bufs.append(int(ptr.value))
            with plan(f1 - f0) as e:
                if f1 > f0:
                    e.process(iq[2 * f0 * nfft: 2 * f1 * nfft], hop=nfft, want_db=False)
                assert e.welch_export_dev(bufs[r], as_f32=dtype == np.float32) == f1 - f0
        with plan(1) as comb:
            row_dev = comb.welch_combine_dev(bufs, counts, as_f32=dtype == np.float32, want_host=True)
            mean_dev, _ = comb.averaged()
        with plan(1) as comb:
            assert np.array_equal(row_dev, comb.welch_combine(parts, counts, want_host=True))
            assert np.array_equal(mean_dev, comb.averaged()[0])
            with pytest.raises(nat.TdsaError):                   # a part that counts needs a pointer
                comb.welch_combine_dev([bufs[0], None, None], counts[:1] + [1, 1], as_f32=dtype == np.float32)
    finally:
        for b in bufs:
            nat.lib.tdsa_peer_free(0, C.c_void_p(b))
    with plan(k) as one:                                         # the reference: one plan, every segment
        row_one = one.process(iq[: 2 * nfft * k], hop=nfft)
        row_one = row_one[-1] if row_one.ndim == 2 else row_one
        mean_one, _ = one.averaged()
    with plan(1) as comb:
        row = comb.welch_combine(parts, counts, want_host=True)
        _check(row, gold, f"Welch partials combined on the device ({np.dtype(dtype).name})")
        mean, cnt = comb.averaged()
        assert cnt == k
        host_mean, _ = sharding.combine_welch([p.astype(np.float64) for p in parts], counts)
        assert np.max(np.abs(mean - host_mean)) <= 1e-15 * np.max(host_mean) * k
        # against ONE plan that saw every segment: that plan adds the segments' power in float32 inside its row pass
        # (the LDS-resident sizes: float32 chunk aggregates), the partial plans in other groupings - the states agree to
        # float32 summation rounding, the rows to a few float32 units of the dB value; float32 partials add 2.6e-7 dB
        assert np.max(np.abs(mean - mean_one)) <= 1e-6 * np.max(mean_one)
        d = np.abs(row - row_one)
        assert np.all(d <= 1e-5), (float(d.max()), float(row_one[np.argmax(d)]))
        # the combining plan carries the state on: segment k + 1 joins the running mean
        nxt = comb.process(iq[2 * nfft * k:], hop=nfft)
        _check(nxt[-1], gold_next, "running mean continued after the combine")
        assert comb.averaged()[1] == k + 1
    with plan(1) as bad:                                         # a capped mean cannot take partials
        bad.configure(avg=("lin", k - 1))
        with pytest.raises(pkg._native.TdsaError):
            bad.welch_combine(parts, counts, want_host=True)


def test_welch_partials_read_in_place_from_another_process(pkg):
    """Ranks on GPUs of one node leave their partial means in device buffers (tdsa_peer_alloc), rank 0 maps them through
    HIP IPC handles (tdsa_peer_open) and tdsa_welch_combine_dev reads them in place - here: two bench.py ranks sharing this
    box's one GPU, the 64 segments of the C5 capture split 32 / 32; the combined dB row passes the bench's parity block
    against the float64 Welch gold and the line says which exchange ran."""
    import json
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29547", "bench.py", "--gpus", "2", "--config", "c5",
                          "--c5-combine", "peer", "--steps", "10", "--warmup", "2", "--reps", "2", "--min-region-s", "0.1"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["welch"]["exchange"] == "peer" and d["welch"]["segments_per_rank"] == [32, 32]
    assert d["parity"]["pass"] is True, d["parity"]
    assert d["welch"]["combine_ms"] < 5.0


def test_debug_knobs_named_in_the_header_exist_and_unknown_names_are_errors(pkg):
    """tdsa_debug_knob: the library reads nothing from the environment; every name the header's developer section lists is
    accepted, anything else - the developer-build-only "big_pre_wgs" included - is an error, not a silent no-op."""
    import re
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    txt = open(os.path.join(root, "include", "tdsa_hip.h")).read()
    sec = txt[txt.index("developer section"):txt.index("int tdsa_debug_knob")]
    names = re.findall(r'"([a-z0-9_]+)"', sec)
    shipped = [n for n in names if n not in ("big_pre_wgs", "cu_mask")]
    assert set(shipped) >= {"num_cu", "avg_wg_min", "avg_f64_chunks", "overlap_share", "big_group", "chirp_single", "chirp_fuse_big"}
    values = {"num_cu": 128, "overlap_share": 50, "big_group": 16}
    with pkg.SpectrumEngine(1 << 15, max_frames=4) as e:
        for n in shipped:
            if n == "smooth_n1":                       # (only a two-pass mixed-radix plan has a split to move)
                with pytest.raises(pkg._native.TdsaError):
                    e.debug_knob(n, 100)
                continue
            e.debug_knob(n, values.get(n, 1))
    with pkg.SpectrumEngine(20000, max_frames=2) as e:
        e.debug_knob("smooth_n1", 100)
        with pytest.raises(pkg._native.TdsaError):
            e.debug_knob("smooth_n1", 7)                # not a divisor
        with pytest.raises(pkg._native.TdsaError):
            e.debug_knob("smooth_n1", 1)                # cofactor 20 000 above the LDS limit
        for bad in ("big_pre_wgs", "cu_mask", "no_such_knob", ""):
            with pytest.raises(pkg._native.TdsaError):
                e.debug_knob(bad, 1)


def test_shader_clock_is_plausible(pkg):
    with pkg.SpectrumEngine(1024, max_frames=1) as e:
        mhz, ns = e.shader_clock()
    assert 800.0 < mhz < 3000.0 and abs(2e3 / ns - mhz) < 1.0


@pytest.mark.parametrize("log2n", [15, 16, 17, 18, 19, 20])
def test_long_frames_all_sizes(pkg, log2n):
    """N = 2^15 .. 2^20 = N1 x 16384: column DFT kernel + the frame kernel as row pass.  HackRF branch
    (per-frame DC removal, normalised Hann, 20log10|X|) on int8 and on complex64 input, hold trace."""
    nfft = 1 << log2n
    iq = so.synth_iq_int8(nfft, nfft, seed=70 + log2n)
    gold, gmax, _ = so.hackrf_batch(iq, nfft, nfft, 20e6, precision="gold")
    with _hackrf_engine(pkg, nfft, 1, hold_max=True) as e:
        out = e.process(iq, hop=nfft)
        assert out.shape == (1, nfft)
        _check(out, gold, f"N=2^{log2n} int8")
        mx, _ = e.hold()
        assert np.array_equal(mx, out[0])
        assert int(np.argmax(out[0])) == nfft // 2 + nfft // 8
        out_c = e.process(so.unpack_iq_int8(iq), hop=nfft)
        _check(out_c, gold, f"N=2^{log2n} complex64")
        assert abs(e.dc_estimate - so.unpack_iq_int8(iq).astype(np.complex128).mean()) < 1e-6


@pytest.mark.parametrize("log2n,avg", [(15, ("exp", 4)), (16, ("lin", 3)), (17, ("lin", 6))])
def test_long_frames_trace_averager(pkg, log2n, avg):
    """TraceAverager exp / capped lin on long frames, one frame per call, RTL branch (power dB)."""
    nfft, nf = 1 << log2n, 5
    iq_i8 = so.synth_iq_int8(nfft * nf, nfft, seed=81)
    gold, gmax, gmin = so.rtl_batch(iq_i8, nfft, nfft, 2e6, precision="gold", avg=avg)
    with pkg.SpectrumEngine(nfft, max_frames=1) as e:
        e.set_window(so.rtl_window("hanning", nfft).astype(np.float32))
        e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=-1.0, avg=avg,
                    hold_max=True, hold_min=True)
        rows = [e.process(iq_i8[2 * nfft * k: 2 * nfft * (k + 1)], hop=nfft)[0] for k in range(nf)]
        _check(np.stack(rows), gold, f"N=2^{log2n} {avg}")
        mx, mn = e.hold()
        assert np.array_equal(mx, np.max(rows, axis=0)) and np.array_equal(mn, np.min(rows, axis=0))
        buf, cnt = e.averaged()
        assert cnt == (1 if avg[0] == "exp" else min(avg[1], nf))
        _check(10 * np.log10(buf + so.POWER_LOG_FLOOR), gold[-1], "averager state")
        with pytest.raises(Exception):                  # several frames per call only in the Welch regime
            e.process(iq_i8[: 2 * nfft * 2], hop=nfft)


def test_long_frame_welch_groups_and_overlap(pkg):
    """Welch at 2^16 with more segments than one column/row round (group of 8) and hop = N/2."""
    nfft, hop, k = 1 << 16, 1 << 15, 19
    iq = so.synth_iq_int8(hop * (k - 1) + nfft, nfft, seed=91)
    x = so.unpack_iq_int8(iq).astype(np.complex128)
    w = so.rtl_window("hanning", nfft)
    acc = np.zeros(nfft)
    for s_ in range(k):
        acc += np.abs(np.fft.fftshift(np.fft.fft(x[s_ * hop: s_ * hop + nfft] * w))) ** 2
    gold = 10 * np.log10(acc / k + so.POWER_LOG_FLOOR)
    with pkg.SpectrumEngine(nfft, max_frames=k) as e:
        e.set_window(w.astype(np.float32))
        e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=-1.0, avg=("lin", k))
        out = e.process(iq, hop=hop)
        _check(out[0], gold, "Welch 19 x 2^16, 50 % overlap")
        mean, cnt = e.averaged()
        assert cnt == k and np.max(np.abs(mean - acc / k) / (acc / k).max()) < 1e-5


@pytest.mark.parametrize("seed", range(int(os.environ.get("TDSA_LONG_CASES", "6"))))
def test_long_frames_random(pkg, seed):
    """Seeded random long-frame cases (2^15 .. 2^18 points; int8 / uint8 / complex64; DC off, per frame or tracked;
    plain frames, TraceAverager exp / capped lin, or a Welch mean fed in random portions with 0-75 % overlap; calibration
    offset, hold traces) against the float64 arithmetic of the reference."""
    rng = np.random.default_rng(7000 + seed)
    log2n = int(rng.integers(15, 19))
    nfft = 1 << log2n
    fmt = str(rng.choice(["i8", "u8", "c64"]))
    dc_alpha = float(rng.choice([-1.0, 1.0, 0.3]))
    kind = str(rng.choice(["plain", "exp", "lin", "welch"]))
    cal = float(rng.choice([0.0, -0.8087]))
    k = int(rng.integers(2, 9))
    hop = nfft if kind != "welch" else int(nfft * float(rng.choice([1.0, 0.5, 0.25])))
    win = so.rtl_window(str(rng.choice(["hanning", "hamming", "rectangle"])), nfft)
    i8 = so.synth_iq_int8(hop * (k - 1) + nfft, nfft, seed=int(rng.integers(1, 1 << 30)))
    if fmt == "i8":
        raw, x = i8, so.unpack_iq_int8(i8).astype(np.complex128)
    elif fmt == "u8":
        raw = (i8.astype(np.int16) + 128).astype(np.uint8)
        x = so.unpack_iq_uint8_rtl(raw).astype(np.complex128)
    else:
        raw = so.unpack_iq_int8(i8)
        x = raw.astype(np.complex128)
    per = 2 if fmt != "c64" else 1
    avg = {"plain": ("off", 1), "exp": ("exp", int(rng.integers(2, 6))), "lin": ("lin", int(rng.integers(2, 5))),
           "welch": ("lin", 64)}[kind]
    # float64 restatement: per segment (x - dc) * w -> |fftshift(fft)|^2, then the averager, then dB + offset
    dc = 0.0 + 0.0j
    powers = []
    for s_ in range(k):
        seg = x[s_ * hop: s_ * hop + nfft]
        if dc_alpha >= 0.0:
            dc = (1.0 - dc_alpha) * dc + dc_alpha * seg.mean()
            seg = seg - dc
        powers.append(np.abs(np.fft.fftshift(np.fft.fft(seg * win))) ** 2)
    av = so.TraceAveragerOracle()
    av.set_mode(*avg)
    with pkg.SpectrumEngine(nfft, max_frames=k) as e:
        e.set_window(win.astype(np.float32))
        e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=dc_alpha, avg=avg,
                    cal_offset_db=cal, hold_max=True, hold_min=True)
        what = f"seed {seed}: 2^{log2n} {fmt} dc {dc_alpha} {kind} {avg} hop {hop} k {k}"
        rows = []
        if kind == "welch":
            done = 0
            while done < k:
                take = int(rng.integers(1, k - done + 1))
                part = raw[per * hop * done: per * (hop * (done + take - 1) + nfft)]
                out = e.process(part, hop=hop, n_frames=take)
                assert out.shape == (1, nfft)
                done += take
                gold = 10 * np.log10(np.mean(powers[:done], axis=0) + so.POWER_LOG_FLOOR) + cal
                _check(out[0], gold, what + f" after {done} segments")
                rows.append(out[0])
        else:
            for s_ in range(k):
                out = e.process(raw[per * hop * s_: per * (hop * s_ + nfft)], hop=nfft, n_frames=1)
                gold = 10 * np.log10(np.asarray(av.process(powers[s_]), dtype=np.float64) + so.POWER_LOG_FLOOR) + cal
                _check(out[0], gold, what + f" frame {s_}")
                rows.append(out[0])
        mx, mn = e.hold()
        assert np.array_equal(mx, np.max(rows, axis=0)) and np.array_equal(mn, np.min(rows, axis=0)), what


def test_hackrf_source_long_fft_size(pkg):
    """set_num_samples above 16384 (unbounded in the reference, hackrf_samples.py:392-405) runs on the GPU."""
    from topdogspectrumanalyser_amd.datasources.replay import ReplayHackRF
    n = 32768
    iq = so.synth_iq_int8(65536 * 2, n, seed=19)
    src = pkg.HackrfSamplesDataSource(sample_rate=20_000_000, centre_freq=100_000_000,
                                      device_factory=lambda: ReplayHackRF(iq))
    src.start()
    try:
        src.set_num_samples(n)
        p, fb = src.get_power_levels()
        assert p.shape == (n,) and fb.shape == (n,) and p.any()
        gold = so.HackrfBranchOracle(n, 20e6, precision="gold").power_levels(src.get_raw_samples())
        _check(p, gold, "HackRF source at 32768 points")
    finally:
        src.stop()


@pytest.mark.parametrize("avg", [("exp", 8), ("lin", 16), ("lin", 5000)])
def test_long_batch_averaging_uses_chunked_scan(pkg, avg):
    """> 128 frames: the averager runs as a three-pass chunked scan (frames x bins parallel); it must
    reproduce the order-dependent recurrence of TraceAverager, also when the batch is split in two calls."""
    nfft, nf = 1024, 700
    iq = so.synth_iq_int8(nfft * nf, nfft, seed=61)
    gold, gmax, gmin = so.hackrf_batch(iq, nfft, nfft, 20e6, precision="gold", avg=avg)
    with pkg.SpectrumEngine(nfft, max_frames=nf) as e:
        e.set_window(so.hackrf_window(nfft))
        e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=1.0, avg=avg,
                    hold_max=True, hold_min=True)
        out = e.process(iq, hop=nfft)
        _check(out, gold, f"chunked {avg}")
        mx, mn = e.hold()
        assert np.array_equal(mx, out.max(axis=0)) and np.array_equal(mn, out.min(axis=0))
        buf, cnt = e.averaged()
        assert cnt == (1 if avg[0] == "exp" else min(avg[1], nf))
        e.reset()
        a = e.process(iq[:2 * nfft * 300], hop=nfft)          # 300 frames (chunked), then 400 more
        b = e.process(iq[2 * nfft * 300:], hop=nfft)
        _check(np.concatenate([a, b]), gold, f"chunked split {avg}")


@pytest.mark.parametrize("avg", [("exp", 4), ("lin", 16), ("lin", 5000), ("exp", 2)])
@pytest.mark.parametrize("nfft,nf", [(4096, 1500), (16384, 700), (8192, 513), (1024, 2500), (512, 17000), (2048, 1100), (1000, 1300), (256, 45000)])
def test_batch_averaging_with_workgroup_chunks(pkg, monkeypatch, avg, nfft, nf):
    """N >= 4096, > 128 frames: the chunks of the averager's chained scan are the frame ranges of the frame kernel's
    workgroups, which form their chunk's aggregate themselves (float32 dot products; chain and re-scan in float64).
    Below 4096 points and for chirp-z sizes, > 1024 frames: up to 256 equal ranges of the batch, aggregates by a pass of
    the scan's own, the same chain.  Rows,
    hold traces and the averager state must follow the order-dependent recurrence of TraceAverager
    (utils/signal_processing.py:35-61) - against the float64 gold, against the three-pass scan with float64 aggregates
    (tdsa_debug_knob avg_f64_chunks) to within what a float32 aggregate can move a row (1e-5 dB), and across a split batch."""
    hop = nfft // 2
    iq = so.synth_iq_int8(hop * (nf - 1) + nfft, nfft, seed=71)
    gold, gmax, gmin = so.hackrf_batch(iq, nfft, hop, 20e6, precision="gold", avg=avg)
    gold_src = so.HackrfBranchOracle(nfft, 20e6, precision="gold")
    gold_src.averager.set_mode(*avg)
    x = so.unpack_iq_int8(iq)
    for k in range(nf):
        gold_src.power_levels(so.frame(x, nfft, hop, k))
    gold_state = np.asarray(gold_src.averager.buffer, dtype=np.float64)

    def run(split, f64_chunks=False):
        with pkg.SpectrumEngine(nfft, max_frames=nf) as e:
            e.set_window(so.hackrf_window(nfft))
            e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=1.0, avg=avg,
                        hold_max=True, hold_min=True)
            if f64_chunks:
                e.debug_knob("avg_f64_chunks", 1)
            if split:
                k0 = 300
                a = e.process(iq[:2 * (hop * (k0 - 1) + nfft)], hop=hop)
                b = e.process(iq[2 * hop * k0:], hop=hop)
                out = np.concatenate([a, b])
            else:
                out = e.process(iq, hop=hop)
            mx, mn = e.hold()
            buf, cnt = e.averaged()
        return out, mx, mn, buf, cnt

    out, mx, mn, buf, cnt = run(False)
    assert out.shape == (nf, nfft)
    _check(out, gold, f"workgroup chunks {avg}")
    assert np.array_equal(mx, out.max(axis=0)) and np.array_equal(mn, out.min(axis=0))
    assert cnt == (1 if avg[0] == "exp" else min(avg[1], nf))
    assert np.max(np.abs(buf - gold_state)) <= 2e-5 * gold_state.max()        # float32 transform under a float64 average
    out_s, _, _, buf_s, _ = run(True)
    _check(out_s, gold, f"workgroup chunks, split batch {avg}")
    out_o, mx_o, _, buf_o, _ = run(False, f64_chunks=True)
    assert np.max(np.abs(out - out_o)) <= 1e-5, np.max(np.abs(out - out_o))
    assert np.max(np.abs(buf - buf_o)) <= 3e-7 * buf_o.max()


@pytest.mark.parametrize("avg", [("exp", 4), ("lin", 16)])
@pytest.mark.parametrize("nfft,nf", [(1024, 100), (4096, 60), (512, 49)])
def test_short_averaged_batches_take_the_chunked_scan_too(pkg, monkeypatch, avg, nfft, nf):
    """Up to 4096 points the workgroup-chunk scan takes over from the one-thread-per-bin kernel at 49 frames already
    (tdsa_debug_knob avg_wg_min): same rows as the sequential kernel to within a float32 aggregate, and the float64 gold."""
    iq = so.synth_iq_int8(nfft * nf, nfft, seed=91)
    gold, _, _ = so.hackrf_batch(iq, nfft, nfft, 20e6, precision="gold", avg=avg)

    def run(wg_min=None):
        with pkg.SpectrumEngine(nfft, max_frames=nf) as e:
            e.set_window(so.hackrf_window(nfft))
            e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=1.0, avg=avg)
            if wg_min is not None:
                e.debug_knob("avg_wg_min", wg_min)
            out = e.process(iq, hop=nfft)
            return out, e.averaged()[0]
    out, buf = run()
    _check(out, gold, f"short averaged batch {avg}")
    out_s, buf_s = run(wg_min=100000)
    assert np.max(np.abs(out - out_s)) <= 1e-5
    assert np.max(np.abs(buf - buf_s)) <= 3e-7 * buf_s.max()


@pytest.mark.parametrize("avg", [("lin", 100000), ("exp", 4), ("lin", 16)])
@pytest.mark.parametrize("nfft,nf", [(1024, 900), (4096, 600), (16384, 400), (2048, 100)])
def test_averaging_for_the_state_alone(pkg, avg, nfft, nf):
    """A batch whose dB rows and hold traces are not wanted (out = NULL, hold off) - the averaged spectrum of a capture, e.g.
    Welch at a native size (utils/signal_processing.py:35-61 with lin n >= frames): the frame kernel forms the chunk
    aggregates without writing the linear rows and the scan ends with its chain.  The state must be the one the full path
    leaves, bit for bit, and continue correctly into a following call that does want rows."""
    import ctypes as C
    nat = pkg._native
    hop = nfft // 2
    ns = hop * (nf - 1) + nfft
    iq = so.synth_iq_int8(ns, nfft, seed=83)
    iq2 = so.synth_iq_int8(ns, nfft, seed=84)

    def configured():
        e = pkg.SpectrumEngine(nfft, max_frames=nf)
        e.set_window(so.hackrf_window(nfft))
        e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=1.0, avg=avg)
        return e
    with configured() as e:
        e.process(iq, hop=hop)
        want, want_cnt = e.averaged()
        rows2 = e.process(iq2, hop=hop)
        want2, _ = e.averaged()
    d_in = C.c_void_p()
    nat.check(nat.lib.tdsa_dev_alloc(0, iq.nbytes, C.byref(d_in)))
    nat.check(nat.lib.tdsa_memcpy_h2d(0, d_in, iq.ctypes.data_as(C.c_void_p), iq.nbytes))
    try:
        with configured() as e:
            e.process_device(nat.IN_I8, d_in.value, ns, hop, nf, None)
            got, got_cnt = e.averaged()
            assert got_cnt == want_cnt and np.array_equal(got, want)
            assert np.array_equal(e.process(iq2, hop=hop), rows2)
            assert np.array_equal(e.averaged()[0], want2)
    finally:
        nat.check(nat.lib.tdsa_dev_free(0, d_in))
    gold_src = so.HackrfBranchOracle(nfft, 20e6, precision="gold")
    gold_src.averager.set_mode(*avg)
    x = so.unpack_iq_int8(iq)
    for k in range(nf):
        gold_src.power_levels(so.frame(x, nfft, hop, k))
    gold_state = np.asarray(gold_src.averager.buffer, dtype=np.float64)
    assert np.max(np.abs(got - gold_state)) <= 2e-5 * gold_state.max()


# ------------------------------------------------------------------------------------------------
# real-input (audio) path: two real channels in one complex FFT
# ------------------------------------------------------------------------------------------------
class _FakeStream:
    def __init__(self, blocks):
        self._blocks, self._i = list(blocks), 0

    def start(self): pass
    def stop(self): pass
    def close(self): pass

    def read(self, n):
        b = self._blocks[self._i]
        self._i += 1
        assert len(b) == n
        return np.array(b, copy=True), False


@pytest.mark.parametrize("mode,chan,psd", [("mono", "mono", False), ("left", "left", False), ("mono_psd", "mono", True)])
def test_audio_source_golden(pkg, golden_dir, mode, chan, psd):
    g = np.load(os.path.join(golden_dir, "audio_1024.npz"))
    n, nf, fs = int(g["nfft"]), int(g["n_frames"]), int(g["sample_rate"])
    st = g["stereo_f32"]
    blocks = [st[k * n:(k + 1) * n] for k in range(nf)]
    src = pkg.MicrophoneSamplesDataSource(sample_rate=fs, stream_factory=lambda rate, block: _FakeStream(blocks))
    src.set_fft_size(n)
    src.set_channel_mode(chan)
    src.set_psd_mode(psd)
    src.start(None)
    src._audio_block = n
    win = so.rtl_window("hanning", n)
    for k in range(nf):
        p, fb = src.get_power_levels()
        assert p.shape == (n // 2 + 1,) and np.array_equal(fb, g["freq_bins"])
        blk = blocks[k].astype(np.float64)
        sig = (blk[:, 0] + blk[:, 1]) * 0.5 if chan == "mono" else blk[:, 0]
        gold = so.audio_db(so.audio_compute_power(sig, win, n, fs, psd, precision="gold"), psd)
        _check(p, gold, f"audio {mode} frame {k} vs gold")
        _check(p, g[mode][k], f"audio {mode} frame {k} vs reference")
    src.stop()


def test_audio_stereo_and_averaging(pkg, golden_dir):
    g = np.load(os.path.join(golden_dir, "audio_1024.npz"))
    n, nf, fs = int(g["nfft"]), int(g["n_frames"]), int(g["sample_rate"])
    st = g["stereo_f32"]
    win = so.rtl_window("hanning", n)
    blocks = [st[k * n:(k + 1) * n] for k in range(nf)]
    src = pkg.MicrophoneSamplesDataSource(sample_rate=fs, stream_factory=lambda rate, block: _FakeStream(blocks))
    src.set_fft_size(n)
    src.set_channel_mode("stereo")
    src.set_averaging("lin", 3)
    src.start(None)
    src._audio_block = n
    av = so.TraceAveragerOracle()
    av.set_mode("lin", 3)
    for k in range(nf):
        (left, right), _ = src.get_power_levels()
        blk = blocks[k].astype(np.float64)
        pl = av.process(so.audio_compute_power(blk[:, 0], win, n, fs, False, precision="gold"))
        pr = so.audio_compute_power(blk[:, 1], win, n, fs, False, precision="gold")
        assert left.dtype == np.float64
        _check(left, so.audio_db(np.array(pl), False), f"stereo left (averaged) {k}")
        _check(right, so.audio_db(pr, False), f"stereo right {k}")
    src.stop()


@pytest.mark.parametrize("avg", [("exp", 4), ("lin", 5000)])
@pytest.mark.parametrize("n,nf", [(1024, 1500), (2048, 300)])
def test_audio_long_averaged_batch(pkg, avg, n, nf):
    """Real-input path (audio_samples.py:121-131) over a long batch with the TraceAverager on (utils/signal_processing.py:35-61):
    above 128 frames the averager runs as the chunked scan of the complex path (above 1024: equal ranges + two-level chain)
    instead of one thread per bin walking every frame."""
    rng = np.random.default_rng(5)
    t = np.arange(n * nf)
    sig = (0.4 * np.sin(2 * np.pi * 0.0317 * t) + 0.01 * rng.standard_normal(n * nf) + 0.02).astype(np.float32)
    st = np.stack([sig, 0.5 * sig], axis=1).astype(np.float32)
    win = so.rtl_window("hanning", n)
    fs = 44100
    with pkg.SpectrumEngine(n, max_frames=nf) as e:
        e.set_window(win.astype(np.float32))
        e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=1.0, avg=avg)
        out = e.process_real2(st, "left")
    av = so.TraceAveragerOracle()
    av.set_mode(*avg)
    for k in range(nf):
        blk = st[k * n:(k + 1) * n, 0].astype(np.float64)
        gold = so.audio_db(np.array(av.process(so.audio_compute_power(blk, win, n, fs, False, precision="gold"))), False)
        if k % 37 == 0 or k == nf - 1:
            _check(out[k], gold, f"audio averaged batch row {k} {avg}")


def test_audio_quiet_channel_is_not_polluted(pkg):
    """Each channel is transformed on its own (as the reference does): a channel 80 dB below the other, or silent,
    keeps the same parity as a loud one.  (A packed z = L + iR transform leaves ~1e-7 of the louder channel's
    amplitude in the quieter one: 5.7 dB of error at -80 dB.)"""
    n, fs = 4096, 48000
    rng = np.random.default_rng(1)
    t = np.arange(n)
    win = so.rtl_window("hanning", n)
    for ratio in (1.0, 1e-2, 1e-4, 0.0):
        left = 0.5 * np.sin(2 * np.pi * 1000.3 * t / fs)
        right = ratio * 0.5 * np.sin(2 * np.pi * 3333.1 * t / fs) + 1e-6 * rng.standard_normal(n)
        st = np.stack([left, right], axis=1).astype(np.float32)
        with pkg.SpectrumEngine(n, max_frames=1) as e:
            e.set_window(win.astype(np.float32))
            e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=1.0)
            gold_r = so.audio_db(so.audio_compute_power(st[:, 1].astype(np.float64), win, n, fs, False, precision="gold"), False)
            gold_l = so.audio_db(so.audio_compute_power(st[:, 0].astype(np.float64), win, n, fs, False, precision="gold"), False)
            _check(e.process_real2(st, "right")[0], gold_r, f"right alone, level ratio {ratio}")
            both = e.process_real2(st, "stereo")[0]
            _check(both[1], gold_r, f"stereo right, level ratio {ratio}")
            _check(both[0], gold_l, f"stereo left, level ratio {ratio}")


@pytest.mark.parametrize("n", [10000, 44100, 48000])
def test_audio_long_frames_that_are_not_a_power_of_two(pkg, n):
    """MicrophoneSamplesDataSource.set_fft_size takes any size (audio_samples.py:208-214) and scipy.fft.rfft any n (:125):
    one second of audio as ONE frame (44100 / 48000 points) rides the chirp-z path of the long-frame kernels like any
    other size above 8192 - mono mix, left and stereo, one-sided power with the inner bins doubled."""
    fs = 48000
    rng = np.random.default_rng(n)
    t = np.arange(2 * n)
    st = np.stack([0.4 * np.sin(2 * np.pi * 997.3 * t / fs) + 0.01 * rng.standard_normal(2 * n) + 0.02,
                   0.1 * np.sin(2 * np.pi * 5003.1 * t / fs) + 0.02 * rng.standard_normal(2 * n)], axis=1).astype(np.float32)
    win = so.rtl_window("hanning", n)
    with pkg.SpectrumEngine(n, max_frames=2) as e:
        e.set_window(win.astype(np.float32))
        e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=1.0)
        mono = e.process_real2(st, "mono")
        both = e.process_real2(st, "stereo")
    assert mono.shape == (2, n // 2 + 1) and both.shape == (2, 2, n // 2 + 1)
    for k in range(2):
        blk = st[k * n:(k + 1) * n].astype(np.float64)
        gold_m = so.audio_db(so.audio_compute_power((blk[:, 0] + blk[:, 1]) * 0.5, win, n, fs, False, precision="gold"), False)
        gold_l = so.audio_db(so.audio_compute_power(blk[:, 0], win, n, fs, False, precision="gold"), False)
        gold_r = so.audio_db(so.audio_compute_power(blk[:, 1], win, n, fs, False, precision="gold"), False)
        _check(mono[k], gold_m, f"mono frame {k}")
        _check(both[k, 0], gold_l, f"left frame {k}")
        _check(both[k, 1], gold_r, f"right frame {k}")


class _EndlessStream:
    """a stereo float32 stream that hands out consecutive blocks of whatever length is asked for"""

    def __init__(self, data):
        self._d, self._pos = data, 0

    def start(self): pass
    def stop(self): pass
    def close(self): pass

    def read(self, n):
        idx = (self._pos + np.arange(n)) % len(self._d)
        self._pos = (self._pos + n) % len(self._d)
        return np.array(self._d[idx], copy=True), False


@pytest.mark.parametrize("seed", range(int(os.environ.get("TDSA_SOURCE_CASES", "6"))))
def test_audio_source_random_histories(pkg, seed):
    """Microphone source with short blocks and the rolling window of audio_samples.py:149-156 (low sample rate: a
    30 ms read is shorter than the FFT), channel mode / PSD / averaging changed between ticks: against the float64
    restatement fed the same rolling buffer."""
    rng = np.random.default_rng(6800 + seed)
    fs = int(rng.choice([8000, 16000, 48000]))
    n = int(rng.choice([256, 1024, 2048, 300, 1000, 1501]))      # the last three: not powers of two (chirp-z path)
    t = np.arange(1 << 16)
    data = np.stack([0.3 * np.sin(2 * np.pi * 440.0 * t / fs) + 0.01 * rng.standard_normal(len(t)) + 0.02,
                     0.1 * np.sin(2 * np.pi * 1234.5 * t / fs) + 0.02 * rng.standard_normal(len(t))], axis=1).astype(np.float32)
    src = pkg.MicrophoneSamplesDataSource(sample_rate=fs, stream_factory=lambda rate, block: _EndlessStream(data))
    src.set_fft_size(n)
    src.start(None)
    block = src._audio_block
    assert block == min(n, max(64, int(fs * 30 / 1000)))
    win = so.rtl_window("hanning", n)
    av = so.TraceAveragerOracle()
    chan, psd = "mono", False
    buf = np.zeros((n, 2), dtype=np.float64)
    pos = 0
    try:
        for step in range(60):
            ev = rng.random()
            if ev < 0.12:
                chan = str(rng.choice(["mono", "left", "right", "stereo"]))
                src.set_channel_mode(chan)
            elif ev < 0.20:
                psd = bool(rng.integers(0, 2))
                src.set_psd_mode(psd)
            elif ev < 0.30:
                mode = [("off", 1), ("exp", int(rng.integers(2, 7))), ("lin", int(rng.integers(2, 9)))][int(rng.integers(0, 3))]
                src.set_averaging(*mode)
                av.set_mode(*mode)
            idx = (pos + np.arange(block)) % len(data)
            pos = (pos + block) % len(data)
            raw = data[idx].astype(np.float64)
            buf = np.concatenate([buf[block:], raw]) if block < n else raw
            res, axis = src.get_power_levels()
            sig = {"mono": (buf[:, 0] + buf[:, 1]) * 0.5, "left": buf[:, 0], "right": buf[:, 1], "stereo": buf[:, 0]}[chan]
            gold = so.audio_db(np.asarray(av.process(so.audio_compute_power(sig, win, n, fs, psd, precision="gold"))), psd)
            what = f"seed {seed} step {step}: fs {fs} n {n} block {block} {chan} psd {psd} avg {av.mode},{av.n}"
            if chan == "stereo":
                left, right = res
                _check(left, gold, what + " left")
                _check(right, so.audio_db(so.audio_compute_power(buf[:, 1], win, n, fs, psd, precision="gold"), psd), what + " right")
            else:
                _check(res, gold, what)
            assert axis.shape == (n // 2 + 1,)
    finally:
        src.stop()


def test_real2_batch_matches_per_frame(pkg):
    n, nf = 2048, 6
    rng = np.random.default_rng(3)
    st = (0.1 * rng.standard_normal((n * nf, 2))).astype(np.float32)
    with pkg.SpectrumEngine(n, max_frames=nf) as e:
        e.set_window(np.hamming(n).astype(np.float32))
        e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=1.0)
        both = e.process_real2(st, "stereo")
        assert both.shape == (nf, 2, n // 2 + 1)
        for ch, name in ((0, "left"), (1, "right")):
            one = e.process_real2(st, name)
            assert np.array_equal(one, both[:, ch])
            for k in (0, nf - 1):
                gold = so.audio_db(so.audio_compute_power(st[k * n:(k + 1) * n, ch].astype(np.float64),
                                                          np.hamming(n), n, 44100, False, precision="gold"), False)
                _check(one[k], gold, f"real2 {name} {k}")


# ------------------------------------------------------------------------------------------------
# overlapped launches (tdsa_set_overlap): same bits as serial execution, state ops stay ordered
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("streams", [2, 3, 4])
def test_overlapped_launches_match_serial(pkg, streams):
    import ctypes as C
    nat = pkg._native
    nfft, hop, nf, calls = 4096, 2048, 300, 7
    ns = hop * (nf - 1) + nfft
    batches = [so.synth_iq_int8(ns, nfft, seed=100 + i) for i in range(calls)]
    serial, smax = [], None
    with _hackrf_engine(pkg, nfft, nf, hold_max=True, hold_min=True) as e:
        for iq in batches:
            serial.append(e.process(iq, hop=hop))
        smax, smin = e.hold()

    d_in, d_out = [], []
    for iq in batches:
        a, b = C.c_void_p(), C.c_void_p()
        nat.check(nat.lib.tdsa_dev_alloc(0, iq.nbytes, C.byref(a)))
        nat.check(nat.lib.tdsa_dev_alloc(0, nf * nfft * 4, C.byref(b)))
        nat.check(nat.lib.tdsa_memcpy_h2d(0, a, iq.ctypes.data_as(C.c_void_p), iq.nbytes))
        d_in.append(a)
        d_out.append(b)
    try:
        with _hackrf_engine(pkg, nfft, nf, hold_max=True, hold_min=True) as e:
            e.set_overlap(streams)
            for rep in range(2):                      # second round exercises reset + re-dirtied state
                for a, b in zip(d_in, d_out):
                    e.process_device(nat.IN_I8, a.value, ns, hop, nf, b.value)
                mx, mn = e.hold()                     # joins the auxiliary streams
                assert np.array_equal(mx, smax) and np.array_equal(mn, smin)
                for i, b in enumerate(d_out):
                    got = np.empty((nf, nfft), dtype=np.float32)
                    nat.check(nat.lib.tdsa_memcpy_d2h(0, got.ctypes.data_as(C.c_void_p), b, got.nbytes))
                    assert np.array_equal(got, serial[i]), f"call {i} differs with {streams} streams"
                e.reset()
                assert e.hold() == (None, None)
            # an order-dependent mode must fall back to the main stream and still be right
            e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=1.0, avg=("exp", 4))
            gold, _, _ = so.hackrf_batch(batches[0], nfft, hop, 20e6, precision="gold", avg=("exp", 4))
            out = e.process(batches[0], hop=hop)
            _check(out, gold, "exp averaging with overlap enabled")
    finally:
        for a, b in zip(d_in, d_out):
            nat.lib.tdsa_dev_free(0, a)
            nat.lib.tdsa_dev_free(0, b)


def test_averager_scan_row_alignment(pkg):
    """Chunked averager scan (> 128 frames): the 4-bins-per-thread kernels need 16-byte aligned rows; an output
    pointer that is only float aligned must fall back to the one-bin kernels with identical rows."""
    import ctypes as C
    nat = pkg._native
    nfft, hop, nf = 1024, 512, 300
    ns = hop * (nf - 1) + nfft
    iq = so.synth_iq_int8(ns, nfft, seed=77)
    d_in, d_out = C.c_void_p(), C.c_void_p()
    nat.check(nat.lib.tdsa_dev_alloc(0, iq.nbytes, C.byref(d_in)))
    nat.check(nat.lib.tdsa_dev_alloc(0, nf * nfft * 4 + 64, C.byref(d_out)))
    nat.check(nat.lib.tdsa_memcpy_h2d(0, d_in, iq.ctypes.data_as(C.c_void_p), iq.nbytes))
    try:
        rows = {}
        for off in (0, 4):
            with pkg.SpectrumEngine(nfft, max_frames=nf) as e:
                e.set_window(so.hackrf_window(nfft))
                e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=1.0, avg=("exp", 8),
                            hold_max=True)
                e.process_device(nat.IN_I8, d_in.value, ns, hop, nf, d_out.value + off)
                e.synchronize()
                got = np.empty((nf, nfft), dtype=np.float32)
                nat.check(nat.lib.tdsa_memcpy_d2h(0, got.ctypes.data_as(C.c_void_p), C.c_void_p(d_out.value + off),
                                                  got.nbytes))
                rows[off] = (got, e.hold()[0])
        assert np.array_equal(rows[0][0], rows[4][0]) and np.array_equal(rows[0][1], rows[4][1])
        gold, _, _ = so.hackrf_batch(iq, nfft, hop, 20e6, precision="gold", avg=("exp", 8))
        _check(rows[4][0], gold, "chunked exp scan, unaligned rows")
    finally:
        nat.lib.tdsa_dev_free(0, d_in)
        nat.lib.tdsa_dev_free(0, d_out)


@pytest.mark.parametrize("streams", [2, 3, 4])
def test_host_entry_points_stay_consistent_with_overlap(pkg, streams):
    """tdsa_process_i8 after tdsa_set_overlap(n > 1): the frame kernel may run on an auxiliary stream, the
    read-back must still wait for it (include/tdsa_hip.h: every other entry point stays sequentially
    consistent).  Large batches in an order-free mode, rows compared bit for bit with a serial plan."""
    nfft, hop, nf, calls = 16384, 8192, 400, 8
    ns = hop * (nf - 1) + nfft
    batches = [so.synth_iq_int8(ns, nfft, seed=300 + i) for i in range(calls)]
    with _hackrf_engine(pkg, nfft, nf, hold_max=True) as e:
        serial = [e.process(iq, hop=hop) for iq in batches]
        smax, _ = e.hold()
    with _hackrf_engine(pkg, nfft, nf, hold_max=True) as e:
        e.set_overlap(streams)
        for i, iq in enumerate(batches):
            got = e.process(iq, hop=hop)
            assert np.array_equal(got, serial[i]), f"host call {i} returned stale rows with {streams} streams"
        mx, _ = e.hold()
        assert np.array_equal(mx, smax)


def test_host_calls_of_every_staging_class_agree(pkg):
    """The host entry points move small calls through pinned, device-visible buffers (read / written in place),
    medium ones through the same buffers with DMA copies and large ones straight from the caller's memory
    (tdsa_capi.cpp: kZeroCopyMax, kPinnedBounceMax): the rows are the same bit for bit whichever way they went."""
    nfft, hop = 1024, 512
    total = 700                                            # 700 frames: 1.4 MB of int8 in, 2.8 MB of rows out
    iq = so.synth_iq_int8(hop * (total - 1) + nfft, nfft, seed=77)
    with _hackrf_engine(pkg, nfft, total) as e:
        whole = e.process(iq, hop=hop)                     # large: no staging
    with _hackrf_engine(pkg, nfft, total) as e:
        parts, pos = [], 0
        for k in (1, 3, 40, 1, 200, 7, 448):               # 1..40 frames: in place; 200: bounce; 448: large
            parts.append(e.process(iq[2 * hop * pos: 2 * (hop * (pos + k - 1) + nfft)], hop=hop, n_frames=k))
            pos += k
        assert pos == total
    assert np.array_equal(np.concatenate(parts), whole)
    x = so.unpack_iq_int8(iq)
    with _hackrf_engine(pkg, nfft, total) as e:
        c = np.concatenate([e.process(x[hop * p: hop * (p + k - 1) + nfft], hop=hop, n_frames=k)
                            for p, k in ((0, 1), (1, 30), (31, 669))])
    _check(c, whole.astype(np.float64), "complex64 through the three staging classes", floor_units=2)


def test_set_overlap_rejects_bad_counts(pkg):
    with _hackrf_engine(pkg, 1024, 4) as e:
        for bad in (0, -1, 5):
            with pytest.raises(Exception):
                e.set_overlap(bad)
        e.set_overlap(1)


# ------------------------------------------------------------------------------------------------
# host pipeline (tdsa_pipe_*): pinned ring, async H2D / kernel / D2H; same results as tdsa_process_i8
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,streams", [(True, 1), (True, 3), (False, 2)])
def test_host_pipe_matches_synchronous_calls(pkg, rows, streams):
    nfft, hop, nf, chunks = 2048, 1024, 97, 9
    ns = hop * (nf - 1) + nfft
    batches = [so.synth_iq_int8(ns, nfft, seed=500 + i) for i in range(chunks)]
    with _hackrf_engine(pkg, nfft, nf, hold_max=True) as e:
        ref = [e.process(iq, hop=hop) for iq in batches]
        ref_max, _ = e.hold()
    with _hackrf_engine(pkg, nfft, nf, hold_max=True) as e:
        e.set_overlap(streams)
        with e.pipe(ns, n_slots=3, rows=rows) as q:
            got, sub = [], 0
            for iq in batches:
                if q.pending == 3:                                   # ring full: drain the oldest first
                    r = q.collect()
                    got.append(None if r is None else r.copy())
                slot = q.acquire()
                assert slot.dtype == np.int8 and slot.size == 2 * ns
                slot[: iq.size] = iq
                q.submit(ns, hop, nf)
                sub += 1
            while q.pending:
                r = q.collect()
                got.append(None if r is None else r.copy())
            assert len(got) == chunks
            if rows:
                for i, (g, r) in enumerate(zip(got, ref)):
                    assert g.shape == r.shape and np.array_equal(g, r), f"chunk {i}"
            else:
                assert all(g is None for g in got)
            mx, _ = e.hold()
            assert np.array_equal(mx, ref_max)


def test_host_pipe_rows_as_uint8_levels(pkg):
    """rows="u8": the read-back leg carries what ImageItem.setImage(rows, levels=(min_db, max_db)) makes of the dB rows
    (displays/waterfall.py:353-356) - byte for byte the numpy expression on the rows a synchronous call returns; levels
    changed between submissions apply to the slots submitted afterwards; the slot's float32 rows stay on the device."""
    import ctypes as C
    nfft, hop, nf, chunks = 4096, 2048, 61, 7
    ns = hop * (nf - 1) + nfft
    batches = [so.synth_iq_int8(ns, nfft, seed=900 + i) for i in range(chunks)]
    levels = [(-120.0, 0.0)] * 2 + [(-60.0, -10.0)] * 3 + [(-37.5, 12.25)] * 2

    def as_u8(rows, lo, hi):
        with np.errstate(invalid="ignore"):
            t = np.clip((rows - lo) / (hi - lo) * 255, 0, 255)         # float32 rows against Python floats: float32
        return np.where(np.isnan(t), 0, t).astype(np.uint8)

    with _hackrf_engine(pkg, nfft, nf, hold_max=True) as e:
        ref = [e.process(iq, hop=hop) for iq in batches]
        ref_max, _ = e.hold()
    with _hackrf_engine(pkg, nfft, nf, hold_max=True) as e:
        e.set_overlap(2)
        with e.pipe(ns, n_slots=3, rows="u8") as q:                     # (-120, 0) until set_levels is called
            got = []
            for i, iq in enumerate(batches):
                if q.pending == 3:
                    got.append(q.collect_u8().copy())
                if i > 0 and levels[i] != levels[i - 1]:
                    q.set_levels(*levels[i])
                q.acquire()[: iq.size] = iq
                q.submit(ns, hop, nf)
            dev_ptr, dev_nf = None, None
            while q.pending:
                if q.pending == 1:                                       # the float32 rows of the slot are still there
                    rows_dev, dev_nf = q.collect_device()
                    last = np.empty((dev_nf, nfft), dtype=np.float32)
                    nat = pkg._native
                    nat.check(nat.lib.tdsa_memcpy_d2h(0, last.ctypes.data_as(C.c_void_p), C.c_void_p(rows_dev), last.nbytes))
                    assert np.array_equal(last, ref[-1])
                    break
                got.append(q.collect_u8().copy())
            with pytest.raises(Exception):
                q.set_levels(0.0, 0.0)
        for i, g in enumerate(got):
            assert g.dtype == np.uint8 and g.shape == ref[i].shape
            assert np.array_equal(g, as_u8(ref[i], *levels[i])), f"chunk {i}"
        assert len(got) == chunks - 1
        mx, _ = e.hold()
        assert np.array_equal(mx, ref_max)
        with e.pipe(ns, n_slots=2, rows=True) as q2:
            q2.acquire()[: batches[0].size] = batches[0]
            q2.submit(ns, hop, nf)
            with pytest.raises(Exception):
                q2.collect_u8()                                          # not a byte-row pipe
            q2.collect()


@pytest.mark.parametrize("seed", range(int(os.environ.get("TDSA_PIPE_CASES", "8"))))
def test_host_pipe_random(pkg, seed):
    """Seeded random pipelines (size incl. a long frame, slots, frames per submit, overlap streams, averaging, tracked
    DC, holds): every slot's rows and the plan state bit for bit what synchronous tdsa_process_i8 calls give."""
    rng = np.random.default_rng(3000 + seed)
    nfft = int(2 ** rng.choice([6, 8, 10, 11, 12, 13, 14, 15]))
    if rng.integers(0, 4) == 0:
        nfft = _any_size(rng)                      # one case in four: a size that is not a power of two
    long_frame = nfft > 16384
    hop = nfft if long_frame else int(rng.choice([nfft, nfft // 2, int(rng.integers(1, nfft + 1))]))
    max_nf = 1 if long_frame else int(rng.integers(1, 40))
    n_slots = int(rng.integers(1, 5))
    streams = int(rng.integers(1, 4))
    avg = [("off", 1), ("exp", 4), ("lin", 7)][int(rng.integers(0, 3))]
    dc_alpha = float(rng.choice([1.0, 0.3, -1.0]))
    averaging = avg[0] != "off"
    mode = dict(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR) if averaging else \
        dict(db_mode="mag", log_floor=so.LOG_FLOOR)
    slot_samples = hop * (max_nf - 1) + nfft
    chunks = []
    for i in range(int(rng.integers(3, 10))):
        nf = int(rng.integers(1, max_nf + 1))
        chunks.append((nf, so.synth_iq_int8(hop * (nf - 1) + nfft, nfft, seed=int(rng.integers(1, 1 << 30)))))

    def engine():
        e = pkg.SpectrumEngine(nfft, max_frames=max_nf)
        e.set_window(so.hackrf_window(nfft))
        e.configure(dc_alpha=dc_alpha, avg=avg, cal_offset_db=-0.8087, hold_max=True, hold_min=True, **mode)
        return e

    with engine() as e:
        ref = [e.process(iq, hop=hop, n_frames=nf) for nf, iq in chunks]
        ref_hold = e.hold()
    with engine() as e:
        e.set_overlap(streams)
        with e.pipe(slot_samples, n_slots=n_slots, rows=True) as q:
            got = []
            for nf, iq in chunks:
                if q.pending == n_slots:
                    got.append(q.collect().copy())
                q.acquire()[: iq.size] = iq
                q.submit(iq.size // 2, hop, nf)
            while q.pending:
                got.append(q.collect().copy())
        hold = e.hold()
    assert len(got) == len(ref)
    for i, (g, r) in enumerate(zip(got, ref)):
        assert g.shape == r.shape and np.array_equal(g, r), (seed, i, nfft, hop, avg, dc_alpha, streams, n_slots)
    for a, b in zip(hold, ref_hold):
        assert (a is None and b is None) or np.array_equal(a, b)


def test_host_pipe_state_modes_and_errors(pkg):
    nfft, hop, nf = 1024, 1024, 16
    ns = nfft * nf
    iqs = [so.synth_iq_int8(ns, nfft, seed=700 + i) for i in range(4)]
    gold, _, _ = so.hackrf_batch(np.concatenate(iqs), nfft, hop, 20e6, precision="gold", avg=("exp", 6))
    with pkg.SpectrumEngine(nfft, max_frames=nf) as e:
        e.set_window(so.hackrf_window(nfft))
        e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=1.0, avg=("exp", 6))
        with e.pipe(ns, n_slots=2) as q:
            with pytest.raises(Exception):
                q.collect()                                   # nothing submitted
            with pytest.raises(Exception):
                q.submit(ns, hop, nf)                         # nothing acquired
            out = []
            for iq in iqs:                                    # order-dependent mode: slots stay in order
                if q.pending == 2:
                    out.append(q.collect().copy())
                q.acquire()[: iq.size] = iq
                q.submit(ns, hop, nf)
            with pytest.raises(Exception):
                q.acquire()                                   # ring full: both slots in flight
            out.append(q.collect().copy())
            q.acquire()
            with pytest.raises(Exception):
                q.acquire()                                   # slot already handed out
            with pytest.raises(Exception):
                q.submit(ns + 1, hop, nf)                     # larger than the slot
            q.submit(ns, hop, nf)                             # (the slot still holds iqs[2])
            while q.pending:
                out.append(q.collect().copy())
        assert len(out) == 5
        _check(np.concatenate(out[:4]), gold, "exp averaging through the pipe")
    with pkg.SpectrumEngine(nfft, max_frames=nf) as e:
        with pytest.raises(Exception):
            e.pipe(nfft - 1)                                  # slot smaller than one frame
        with pytest.raises(Exception):
            e.pipe(ns, n_slots=0)


def test_stream_spectra_helper(pkg):
    from topdogspectrumanalyser_amd.utils.streaming import stream_spectra
    nfft, hop = 4096, 2048
    sizes = [40000, 40000, 17000, 4096, 39999]
    chunks = [so.synth_iq_int8(s, nfft, seed=900 + i) for i, s in enumerate(sizes)]
    nf_max = (sizes[0] - nfft) // hop + 1
    with _hackrf_engine(pkg, nfft, nf_max, hold_max=True) as e:
        ref = [e.process(c, hop=hop) for c in chunks]
        ref_max, _ = e.hold()
    with _hackrf_engine(pkg, nfft, nf_max, hold_max=True) as e:
        got = list(stream_spectra(e, chunks, hop=hop))
        mx, _ = e.hold()
    assert len(got) == len(ref)
    for g, r in zip(got, ref):
        assert g.shape == r.shape and np.array_equal(g, r)
    assert np.array_equal(mx, ref_max)
    with _hackrf_engine(pkg, nfft, nf_max) as e:                               # the rows as the display's bytes
        got8 = list(stream_spectra(e, chunks, hop=hop, rows="u8", levels=(-90.0, -5.0)))
    for g, r in zip(got8, ref):
        t = np.clip((r - (-90.0)) / (-5.0 - (-90.0)) * 255, 0, 255)
        assert g.dtype == np.uint8 and np.array_equal(g, t.astype(np.uint8))
    with _hackrf_engine(pkg, nfft, nf_max) as e:
        assert list(stream_spectra(e, [])) == []
        with pytest.raises(ValueError):
            list(stream_spectra(e, [np.zeros(2 * 100, dtype=np.int8)]))
        with pytest.raises(ValueError):
            list(stream_spectra(e, [chunks[2], chunks[0]], hop=hop))       # later chunk larger than the slot


# ------------------------------------------------------------------------------------------------
# seeded random sweep over sizes x formats x hops x modes: a wider net than the hand-picked cases
# ------------------------------------------------------------------------------------------------
def _any_size(rng):
    """A frame length the native kernels do not cover: anything in [3, 8192] that is not a power of two >= 64
    (small, prime, highly composite and just-below-the-limit sizes all get their share; the sizes above 8192 have
    their own sweep below)."""
    while True:
        kind = int(rng.integers(0, 4))
        n = int([rng.integers(3, 64), rng.integers(64, 1025), rng.integers(1025, 8193),
                 rng.choice([1000, 1009, 1536, 3000, 4095, 4097, 6000, 8000, 8191])][kind])
        if n < 64 or n & (n - 1):
            return n


def _random_case(rng, any_size=False):
    nfft = _any_size(rng) if any_size else int(2 ** rng.integers(6, 15))
    nf = int(rng.integers(1, 24))
    hop = int(rng.choice([nfft, nfft // 2, nfft // 4 + 1, int(rng.integers(1, 2 * nfft))]))
    branch = str(rng.choice(["hackrf", "rtl"]))
    avg = [("off", 1), ("exp", int(rng.integers(2, 9))), ("lin", int(rng.integers(2, 30)))][int(rng.integers(0, 3))]
    return dict(nfft=nfft, nf=nf, hop=hop, branch=branch, avg=avg, psd=bool(rng.integers(0, 2)),
                dc_alpha=float(rng.choice([1.0, 1.0, 0.25])), cal=float(rng.choice([0.0, -0.8087, 3.5])),
                window=str(rng.choice(["hanning", "hamming", "rectangle"])), seed=int(rng.integers(1, 1 << 30)))


@pytest.mark.parametrize("case_id", range(int(os.environ.get("TDSA_SWEEP_CASES", "48"))))
def test_random_configuration_sweep(pkg, case_id):
    _run_sweep_case(pkg, case_id, _random_case(np.random.default_rng(4242 + case_id)))


def _run_sweep_case(pkg, case_id, c):
    nfft, nf, hop = c["nfft"], c["nf"], c["hop"]
    iq = so.synth_iq_int8(hop * (nf - 1) + nfft, nfft, seed=c["seed"])
    fs = 20e6 if c["branch"] == "hackrf" else 2e6
    averaging = c["avg"][0] != "off" and c["avg"][1] > 1
    if c["branch"] == "hackrf":
        gold, gmax, gmin = so.hackrf_batch(iq, nfft, hop, fs, use_psd=c["psd"], avg=c["avg"], dc_alpha=c["dc_alpha"],
                                           cal_offset_db=c["cal"], precision="gold")
        window, dc = so.hackrf_window(nfft), c["dc_alpha"]
    else:
        gold, gmax, gmin = so.rtl_batch(iq, nfft, hop, fs, window=c["window"], use_psd=c["psd"], avg=c["avg"],
                                        cal_offset_db=c["cal"], precision="gold")
        window, dc = so.rtl_window(c["window"], nfft), -1.0
    if c["psd"]:
        mode = dict(db_mode="pow", power_scale=1.0 / (fs * nfft), log_floor=so.LOG_FLOOR)
    elif averaging or c["branch"] == "rtl":
        mode = dict(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR)
    else:
        mode = dict(db_mode="mag", log_floor=so.LOG_FLOOR)
    with pkg.SpectrumEngine(nfft, max_frames=nf) as e:
        e.set_window(window)
        e.configure(dc_alpha=dc, avg=c["avg"], cal_offset_db=c["cal"], hold_max=True, hold_min=True, **mode)
        cut = int(np.random.default_rng(case_id).integers(0, nf + 1))       # state must carry across calls
        parts = []
        if cut > 0:
            parts.append(e.process(iq[: 2 * (hop * (cut - 1) + nfft)], hop=hop, n_frames=cut))
        if cut < nf:
            parts.append(e.process(iq[2 * hop * cut:], hop=hop, n_frames=nf - cut))
        out = np.concatenate(parts)
        mx, mn = e.hold()
    what = f"case {case_id}: {c}"
    _check(out, gold, what)
    _check(mx, gmax, what + " max hold", rows_gold=gold)
    _check(mn, gmin, what + " min hold", rows_gold=gold)


@pytest.mark.parametrize("case_id", range(int(os.environ.get("TDSA_ANYSIZE_CASES", "24"))))
def test_random_configuration_sweep_any_size(pkg, case_id):
    """The same sweep over frame lengths that are NOT a power of two (np.fft.fft / scipy.fft.fft take any N,
    hackrf_samples.py:370, rtl_samples.py:170): chirp-z path of tdsa_chirp.hip, same bounds."""
    _run_sweep_case(pkg, case_id, _random_case(np.random.default_rng(777 + case_id), any_size=True))


@pytest.mark.parametrize("case_id", range(int(os.environ.get("TDSA_ANYSIZE_LONG_CASES", "16"))))
def test_random_configuration_sweep_any_long_size(pkg, case_id):
    """... and over the sizes above 8192 that are not a power of two (8193 .. 65536: M = 2^15 .. 2^17, the two M-point
    transforms of the chirp-z convolution on the long-frame kernels), every branch / window / averaging / PSD / DC mode
    of the sweep, same bounds."""
    rng = np.random.default_rng(31337 + case_id)
    c = _random_case(rng, any_size=True)
    while True:
        n = int(rng.choice([rng.integers(8193, 16384), rng.integers(16385, 32768), rng.integers(32769, 65536),
                            rng.choice([8193, 10000, 16383, 16385, 20000, 44100, 48000, 65535])]))
        if n & (n - 1):
            break
    c["nfft"], c["nf"] = n, int(rng.integers(1, 7))
    c["hop"] = int(rng.choice([n, n // 2, n // 4 + 1, int(rng.integers(1, 2 * n))]))
    _run_sweep_case(pkg, case_id, c)


# ------------------------------------------------------------------------------------------------
# random CALL SEQUENCES on one plan: process in pieces, averaging changes, resets, calibration offset, tare
# baseline - checked call by call against the float64 oracle driven through the same sequence
# ------------------------------------------------------------------------------------------------
def _mode_for(psd, averaging, fs, nfft):
    if psd:
        return dict(db_mode="pow", power_scale=1.0 / (fs * nfft), log_floor=so.LOG_FLOOR)
    if averaging:
        return dict(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR)
    return dict(db_mode="mag", power_scale=1.0, log_floor=so.LOG_FLOOR)


def call_sequence_trial(pkg, trial, report=None, any_size=False):
    """-> (worst dB error in allowance units of ONE rounding unit, worst relative power error, process calls)"""
    nat = pkg._native
    rng = np.random.default_rng((9000 if not any_size else 19000) + trial)
    nfft = _any_size(rng) if any_size else int(2 ** rng.integers(6, 14))
    hop = int(rng.choice([nfft, nfft // 2, int(rng.integers(1, 2 * nfft))]))
    fs, psd = 20e6, bool(rng.integers(0, 2))
    dc_alpha = float(rng.choice([1.0, 1.0, 0.3]))
    max_call, total = 12, 90
    iq = so.synth_iq_int8(hop * (total - 1) + nfft, nfft, seed=int(rng.integers(1, 1 << 30)))
    fmt = str(rng.choice(["i8", "u8", "c64"]))           # the three input formats of the C-ABI
    if fmt == "i8":
        raw, x, per = iq, so.unpack_iq_int8(iq), 2
    elif fmt == "u8":
        raw = (iq.astype(np.int16) + 128).astype(np.uint8)
        x, per = so.unpack_iq_uint8_rtl(raw), 2
    else:
        raw = so.unpack_iq_int8(iq)
        x, per = raw, 1
    br = so.HackrfBranchOracle(nfft, fs, dc_alpha, psd, "gold")
    hold, hold_allow = so.HoldOracle(True, True), so.HoldAllowance()
    cal, tare = 0.0, None
    worst_units, worst_rel, calls = 0.0, 0.0, 0
    with pkg.SpectrumEngine(nfft, max_frames=max_call) as e:
        e.set_window(so.hackrf_window(nfft))
        e.configure(dc_alpha=dc_alpha, avg=("off", 1), cal_offset_db=0.0, hold_max=True, hold_min=True,
                    **_mode_for(psd, False, fs, nfft))
        pos = 0
        while pos < total:
            op = int(rng.integers(0, 10))
            if op == 0:
                avg = [("off", 1), ("exp", int(rng.integers(2, 9))), ("lin", int(rng.integers(2, 20)))][int(rng.integers(0, 3))]
                # tdsa_set_mode restarts the average when mode or length CHANGE (include/tdsa_hip.h)
                if (avg[0], max(1, avg[1])) != (br.averager.mode, br.averager.n):
                    br.averager.set_mode(*avg)
                e.configure(avg=avg, **_mode_for(psd, br.averager.is_active, fs, nfft))
            elif op == 1:
                br.averager.reset()
                e.reset(nat.RESET_AVG)
            elif op == 2:
                hold, hold_allow = so.HoldOracle(True, True), so.HoldAllowance()
                e.reset(nat.RESET_HOLD_MAX | nat.RESET_HOLD_MIN)
            elif op == 3:
                cal = float(rng.choice([0.0, -0.8087, 3.5]))
                e.configure(cal_offset_db=cal)
            elif op == 4:
                tare = None if rng.integers(0, 2) else rng.uniform(-3, 3, nfft).astype(np.float32)
                e.set_tare_baseline(tare)
            else:
                k = int(min(rng.integers(1, max_call + 1), total - pos))
                out = e.process(raw[per * hop * pos: per * (hop * (pos + k - 1) + nfft)], hop=hop, n_frames=k)
                gold = np.empty((k, nfft))
                for j in range(k):
                    g = np.asarray(br.power_levels(so.frame(x, nfft, hop, pos + j)), dtype=np.float64) + cal
                    if tare is not None:
                        g = g - tare.astype(np.float64)
                    gold[j] = g
                    hold.update(g)
                mx, mn = e.hold()
                hold_allow.update(gold)          # the traces' allowance follows from that of every row they have seen
                pairs = (so.parity_metrics(out, gold), hold_allow.metrics(mx, hold.max), hold_allow.metrics(mn, hold.min))
                units = max(p[1] for p in pairs) / 1e-3
                rel = max(p[0] for p in pairs)
                if report is not None and (units > 2.0 or rel > REL_TOL):
                    report(f"trial {trial} pos {pos} k {k}: nfft {nfft} hop {hop} psd {psd} dc {dc_alpha} avg "
                           f"{br.averager.mode},{br.averager.n} cal {cal} tare {tare is not None}: {units:.2f} units, rel {rel:.1e} "
                           f"(rows {pairs[0][1] / 1e-3:.2f}, max hold {pairs[1][1] / 1e-3:.2f}, min hold {pairs[2][1] / 1e-3:.2f})")
                worst_units, worst_rel, calls = max(worst_units, units), max(worst_rel, rel), calls + 1
                pos += k
    return worst_units, worst_rel, calls


# trials of round 5's 1200-trial soak that stood above two units under the old rule (hold traces judged like single rows):
# 871 (N = 8192, lin 16, tare: MIN hold 2.49, rows 0.08), any-size 1031 (N = 5921: min hold 2.17, rows 1.17) and any-size 83
# (N = 4095 chirp-z, exp 8, tracked DC: a ROW at 2.03) - kept as fixed regression cases
SEQUENCE_REGRESSIONS = [(871, False), (1031, True), (83, True)]


def _sequence_bound(pkg, trial, any_size):
    """(bound in rounding units, nfft): two for one transform, three where a frame takes two (chirp-z sizes)"""
    rng = np.random.default_rng((9000 if not any_size else 19000) + trial)
    nfft = _any_size(rng) if any_size else int(2 ** rng.integers(6, 14))
    return (3.0 if _two_transforms(nfft) else 2.0), nfft


@pytest.mark.parametrize("trial,any_size", SEQUENCE_REGRESSIONS)
def test_call_sequences_that_crossed_the_old_allowance(pkg, trial, any_size):
    bound, nfft = _sequence_bound(pkg, trial, any_size)
    units, rel, calls = call_sequence_trial(pkg, trial, any_size=any_size)
    assert calls > 0 and rel <= REL_TOL and units <= bound, f"trial {trial} (N = {nfft}): {units:.2f} units, rel {rel:.2e}"


_TRIAL0 = int(os.environ.get("TDSA_TRIAL_OFFSET", "0"))      # soak runs: fresh seeds beyond those of earlier soaks


@pytest.mark.parametrize("trial", range(_TRIAL0, _TRIAL0 + int(os.environ.get("TDSA_SEQUENCE_TRIALS", "24"))))
def test_random_call_sequences(pkg, trial):
    units, rel, calls = call_sequence_trial(pkg, trial)
    assert calls > 0 and rel <= REL_TOL and units <= 2.0, f"trial {trial}: {units:.2f} units, rel {rel:.2e}"


@pytest.mark.parametrize("trial", range(_TRIAL0, _TRIAL0 + int(os.environ.get("TDSA_ANYSIZE_TRIALS", "12"))))
def test_random_call_sequences_any_size(pkg, trial):
    bound, nfft = _sequence_bound(pkg, trial, True)
    units, rel, calls = call_sequence_trial(pkg, trial, any_size=True)
    assert calls > 0 and rel <= REL_TOL and units <= bound, f"trial {trial} (N = {nfft}): {units:.2f} units, rel {rel:.2e}"


def test_engine_closes_its_pipes_first(pkg):
    """A pipe holds slots the plan's streams write to: closing the engine first (or leaving `with` blocks out
    of order) closes its live pipes before the plan goes, and a later pipe.close() is a no-op."""
    nfft, nf = 1024, 8
    iq = so.synth_iq_int8(nfft * nf, nfft, seed=23)
    e = _hackrf_engine(pkg, nfft, nf)
    q = e.pipe(nfft * nf, n_slots=2, rows=True)
    q.acquire()[: iq.size] = iq
    q.submit(nfft * nf, nfft, nf)
    rows = np.array(q.collect(), copy=True)
    q.acquire()[: iq.size] = iq
    q.submit(nfft * nf, nfft, nf)                     # one slot still in flight when the engine goes away
    e.close()
    assert not q._q and q not in e._pipes
    q.close()                                         # nothing left to free, must not touch the dead plan
    gold, _, _ = so.hackrf_batch(iq, nfft, nfft, 20e6, precision="gold", hold=False)
    _check(rows, gold, "pipe rows before the engine closed")


def test_host_pipe_device_rows_feed_analytics(pkg):
    """rows="device": the dB rows of every slot stay on the GPU and are handed to the analytics as a device
    pointer; results equal those computed from rows read back the ordinary way."""
    from topdogspectrumanalyser_amd import analytics as an
    import ctypes as C
    nat = pkg._native
    nfft, hop, nf, chunks = 4096, 2048, 60, 5
    ns = hop * (nf - 1) + nfft
    batches = [so.synth_iq_int8(ns, nfft, seed=1200 + i) for i in range(chunks)]
    with _hackrf_engine(pkg, nfft, nf) as e:
        ref = [e.process(iq, hop=hop) for iq in batches]
    with _hackrf_engine(pkg, nfft, nf) as e, e.pipe(ns, n_slots=2, rows="device") as q:
        with pytest.raises(Exception):
            q.collect_device()
        seen = 0

        def check():
            nonlocal seen
            ptr, n = q.collect_device()
            assert n == nf and ptr
            got = np.empty((nf, nfft), dtype=np.float32)
            nat.check(nat.lib.tdsa_memcpy_d2h(0, got.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), got.nbytes))
            assert np.array_equal(got, ref[seen])
            peak, pbin, _ = an.rows_stats(e, ptr, n)
            assert np.array_equal(peak, ref[seen].max(axis=1)) and np.array_equal(pbin, ref[seen].argmax(axis=1))
            seen += 1

        for iq in batches:
            if q.pending == 2:
                check()
            q.acquire()[: iq.size] = iq
            q.submit(ns, hop, nf)
        while q.pending:
            check()
        assert seen == chunks
    with _hackrf_engine(pkg, nfft, nf) as e, e.pipe(ns, rows=False) as q:
        q.acquire()[: batches[0].size] = batches[0]
        q.submit(ns, hop, nf)
        with pytest.raises(Exception):
            q.collect_device()                                  # this pipe keeps no rows
        q.collect()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_process_sharded_threads(pkg, world):
    """Frame sharding from one process: a thread + plan per shard (all on GPU 0 here), halo handled, rows
    and hold traces identical to the unsharded run."""
    from topdogspectrumanalyser_amd.sharding import process_sharded
    nfft, hop, nf = 2048, 768, 101
    iq = so.synth_iq_int8(hop * (nf - 1) + nfft, nfft, seed=77)
    w = so.hackrf_window(nfft)
    with _hackrf_engine(pkg, nfft, nf, hold_max=True, hold_min=True) as e:
        ref = e.process(iq, hop=hop)
        rmx, rmn = e.hold()
    rows, mx, mn = process_sharded(iq, nfft, hop, [0] * world, w, hold="maxmin", db_mode="mag",
                                   log_floor=so.LOG_FLOOR, dc_alpha=1.0)
    assert np.array_equal(rows, ref) and np.array_equal(mx, rmx) and np.array_equal(mn, rmn)
    with pytest.raises(ValueError):
        process_sharded(iq, nfft, hop, [0], w, avg=("exp", 4))
    short, _, _ = process_sharded(iq[: 2 * 100], nfft, hop, [0, 0], w)
    assert short.shape == (0, nfft)


@pytest.mark.parametrize("seed", range(int(os.environ.get("TDSA_SHARD_CASES", "6"))))
def test_process_sharded_random(pkg, seed):
    """Seeded random captures split over 1 .. 9 shards (more shards than frames included; int8 and complex64; any
    hop; DC per frame or off): rows and hold traces bit for bit the unsharded result."""
    from topdogspectrumanalyser_amd.sharding import process_sharded
    rng = np.random.default_rng(9500 + seed)
    nfft = int(2 ** rng.integers(6, 14)) if rng.integers(0, 4) else _any_size(rng)
    hop = int(rng.choice([nfft, max(1, nfft // 2), int(rng.integers(1, 2 * nfft))]))
    nf = int(rng.integers(1, 60))
    world = int(rng.integers(1, 10))
    dc_alpha = float(rng.choice([1.0, -1.0]))
    iq = so.synth_iq_int8(hop * (nf - 1) + nfft + int(rng.integers(0, hop)), nfft, seed=int(rng.integers(1, 1 << 30)))
    data = iq if rng.integers(0, 2) else so.unpack_iq_int8(iq)
    w = so.hackrf_window(nfft)
    with pkg.SpectrumEngine(nfft, max_frames=nf) as e:
        e.set_window(w)
        e.configure(db_mode="mag", log_floor=so.LOG_FLOOR, dc_alpha=dc_alpha, hold_max=True, hold_min=True)
        ref = e.process(data, hop=hop, n_frames=nf)
        rmx, rmn = e.hold()
    rows, mx, mn = process_sharded(data, nfft, hop, [0] * world, w, hold="maxmin", db_mode="mag",
                                   log_floor=so.LOG_FLOOR, dc_alpha=dc_alpha)
    assert rows.shape == ref.shape and np.array_equal(rows, ref), (seed, nfft, hop, nf, world)
    assert np.array_equal(mx, rmx) and np.array_equal(mn, rmn)


def test_no_device_memory_left_behind(pkg):
    """Plans, pipes, trace objects and display accumulators created, used through their lazily allocating paths
    (averaging, real input, long frames, analytics scratch) and closed, many times: the free device memory comes back."""
    if os.environ.get("PYTEST_XDIST_WORKER"):
        pytest.skip("free device memory is a property of the whole GPU: other xdist workers allocate beside this one")
    import torch
    from topdogspectrumanalyser_amd import analytics as an

    def cycle(i):
        rng = np.random.default_rng(i)
        nfft = int(2 ** rng.integers(6, 17))
        nf = 1 if nfft > 16384 else int(rng.integers(1, 40))
        iq = so.synth_iq_int8(nfft * nf, nfft, seed=i)
        with pkg.SpectrumEngine(nfft, max_frames=max(nf, 130)) as e:
            e.set_window(so.hackrf_window(nfft))
            e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=0.5, avg=("exp", 4),
                        hold_max=True)
            e.process(iq, hop=nfft, n_frames=nf if nfft <= 16384 else 1)
            if nfft <= 16384:
                e.configure(avg=("off", 1), dc_alpha=1.0)
                st = rng.standard_normal((nfft * 2, 2)).astype(np.float32)
                e.process_real2(st, "stereo")
                with e.pipe(nfft * nf, n_slots=2) as q:
                    q.acquire()[: iq.size] = iq
                    q.submit(nfft * nf, nfft, nf)
                    rows = q.collect()
                    assert rows.shape == (nf, nfft)
                e.set_overlap(3)
        with an.DensityHistogram(256, 0.9) as dh, an.WaterfallRing(10, 256, -120.0) as wf:
            dh.update(np.zeros(256, dtype=np.float32))
            wf.push(np.zeros(256, dtype=np.float32))
        av = pkg.TraceAverager()
        av.set_mode("exp", 4)
        av.process(np.ones(512, dtype=np.float32))

    for i in range(3):
        cycle(i)                                   # first-use allocations of the runtime itself
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for i in range(40):
        cycle(100 + i)
    import gc
    gc.collect()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, f"{(free0 - free1) / 2**20:.1f} MiB of device memory not returned"



# ------------------------------------------------------------------------------------------------
# several captures in one call (tdsa_process_dev_batch): the bits of consecutive tdsa_process_dev calls
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nfft,hop,nf,fmt", [(16384, 8192, 61, "i8"), (4096, 4096, 37, "u8"), (256, 100, 50, "i8"),
                                             (1024, 512, 33, "c64"), (1000, 500, 9, "i8"), (1 << 15, 1 << 15, 1, "i8")])
@pytest.mark.parametrize("mode", ["hold", "exp", "tracked_dc"])
def test_batched_captures_match_consecutive_calls(pkg, nfft, hop, nf, fmt, mode):
    """n_seg captures handed over at once: rows, hold traces, DC state and averager state are exactly those of n_seg
    consecutive calls - whether the captures leave as one persistent launch (order-free modes on LDS-resident sizes)
    or are run one after the other inside the call (averaging, tracked DC, chirp-z and long-frame plans).  The
    segment strides are larger than the captures and odd multiples of a sample, the output stride has a gap."""
    import ctypes as C
    nat = pkg._native
    if nfft > 16384 and mode == "exp":
        pytest.skip("one frame per call there")
    n_seg = 5
    ns = hop * (nf - 1) + nfft
    bps = 8 if fmt == "c64" else 2
    in_fmt = {"i8": nat.IN_I8, "u8": nat.IN_U8, "c64": nat.IN_C64}[fmt]
    seg_stride = ns * bps + 3 * bps * 7                       # gap of 21 samples between captures
    out_stride = nf * nfft + 64
    rng = np.random.default_rng(nfft + nf)
    caps = []
    for sgi in range(n_seg):
        iq = so.synth_iq_int8(ns, nfft if nfft >= 8 else 8, seed=500 + sgi)
        if fmt == "u8":
            iq = (iq.astype(np.int16) + 128).astype(np.uint8)
        elif fmt == "c64":
            iq = so.unpack_iq_int8(iq).astype(np.complex64)
        caps.append(iq)
    blob = np.zeros(seg_stride * n_seg, dtype=np.uint8)
    for sgi, iq in enumerate(caps):
        raw = iq.view(np.uint8)
        blob[sgi * seg_stride: sgi * seg_stride + raw.size] = raw
    cfg = dict(db_mode="mag", log_floor=so.LOG_FLOOR, dc_alpha=1.0, hold_max=True, hold_min=True)
    if mode == "exp":
        cfg = dict(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=1.0, avg=("exp", 4), hold_max=True)
    elif mode == "tracked_dc":
        cfg.update(dc_alpha=0.25)
    d_in, d_out = C.c_void_p(), C.c_void_p()
    nat.check(nat.lib.tdsa_dev_alloc(0, blob.nbytes, C.byref(d_in)))
    nat.check(nat.lib.tdsa_dev_alloc(0, out_stride * n_seg * 4, C.byref(d_out)))
    nat.check(nat.lib.tdsa_memcpy_h2d(0, d_in, blob.ctypes.data_as(C.c_void_p), blob.nbytes))
    rows_out = 1 if nfft > 16384 else nf

    def run(batched):
        with pkg.SpectrumEngine(nfft, max_frames=nf) as e:
            e.set_window(so.hackrf_window(nfft))
            e.configure(**cfg)
            for rep in range(2):                              # the second round starts from non-trivial state
                if batched:
                    e.process_device_batch(in_fmt, d_in.value, seg_stride, n_seg, ns, hop, nf, d_out.value, out_stride)
                else:
                    for sgi in range(n_seg):
                        e.process_device(in_fmt, d_in.value + sgi * seg_stride, ns, hop, nf,
                                         d_out.value + 4 * sgi * out_stride)
            e.synchronize()
            got = np.empty(out_stride * n_seg, dtype=np.float32)
            nat.check(nat.lib.tdsa_memcpy_d2h(0, got.ctypes.data_as(C.c_void_p), d_out, got.nbytes))
            rows = [got[sgi * out_stride: sgi * out_stride + rows_out * nfft].reshape(rows_out, nfft).copy()
                    for sgi in range(n_seg)]
            return rows, e.hold(), e.dc_estimate, e.info().frames_held_max

    try:
        nat.check(nat.lib.tdsa_memcpy_h2d(0, d_out, np.full(out_stride * n_seg, np.nan, np.float32).ctypes.data_as(C.c_void_p),
                                          out_stride * n_seg * 4))
        seq = run(False)
        nat.check(nat.lib.tdsa_memcpy_h2d(0, d_out, np.full(out_stride * n_seg, np.nan, np.float32).ctypes.data_as(C.c_void_p),
                                          out_stride * n_seg * 4))
        bat = run(True)
    finally:
        nat.lib.tdsa_dev_free(0, d_in)
        nat.lib.tdsa_dev_free(0, d_out)
    for sgi in range(n_seg):
        assert np.array_equal(seq[0][sgi], bat[0][sgi]), f"capture {sgi} differs"
    for a, b in zip(seq[1], bat[1]):
        assert (a is None and b is None) or np.array_equal(a, b)
    assert seq[2] == bat[2] and seq[3] == bat[3]
    if mode == "hold" and nfft <= 16384 and fmt == "i8":      # and the rows are right, not merely equal
        gold, _, _ = so.hackrf_batch(caps[2], nfft, hop, 20e6, precision="gold")
        _check(bat[0][2], gold, "capture 2 of the batch")


@pytest.mark.parametrize("nfft,fmt", [(4096, "i8"), (16384, "i8"), (512, "u8"), (2048, "c64")])
@pytest.mark.parametrize("out_gap", [0, 96])
def test_batched_single_frame_captures(pkg, nfft, fmt, out_gap):
    """ONE frame per capture - one get_power_levels() per queued chunk, the use tdsa_hip.h cites for the batch call
    (hackrf_samples.py:254-305).  ceil(2^32 / 1) does not fit the kernel's 32-bit segment divisor, so the library runs
    such a batch as one capture whose frames sit a segment stride apart (contiguous rows) or capture by capture
    (gapped rows): rows, hold traces and DC state must be those of consecutive calls either way, with gapped input."""
    import ctypes as C
    nat = pkg._native
    n_seg, nf, hop = 7, 1, nfft
    bps = 8 if fmt == "c64" else 2
    in_fmt = {"i8": nat.IN_I8, "u8": nat.IN_U8, "c64": nat.IN_C64}[fmt]
    seg_stride = nfft * bps + 5 * bps * 3                      # gap of 15 samples between captures
    out_stride = nfft + out_gap
    caps = []
    for sgi in range(n_seg):
        iq = so.synth_iq_int8(nfft, nfft, seed=700 + sgi)
        if fmt == "u8":
            iq = (iq.astype(np.int16) + 128).astype(np.uint8)
        elif fmt == "c64":
            iq = so.unpack_iq_int8(iq).astype(np.complex64)
        caps.append(iq)
    blob = np.zeros(seg_stride * n_seg, dtype=np.uint8)
    for sgi, iq in enumerate(caps):
        raw = iq.view(np.uint8)
        blob[sgi * seg_stride: sgi * seg_stride + raw.size] = raw
    d_in, d_out = C.c_void_p(), C.c_void_p()
    nat.check(nat.lib.tdsa_dev_alloc(0, blob.nbytes, C.byref(d_in)))
    nat.check(nat.lib.tdsa_dev_alloc(0, out_stride * n_seg * 4, C.byref(d_out)))
    nat.check(nat.lib.tdsa_memcpy_h2d(0, d_in, blob.ctypes.data_as(C.c_void_p), blob.nbytes))

    def run(batched):
        nan = np.full(out_stride * n_seg, np.nan, np.float32)
        nat.check(nat.lib.tdsa_memcpy_h2d(0, d_out, nan.ctypes.data_as(C.c_void_p), nan.nbytes))
        with _hackrf_engine(pkg, nfft, 4, hold_max=True, hold_min=True) as e:
            for rep in range(2):
                if batched:
                    e.process_device_batch(in_fmt, d_in.value, seg_stride, n_seg, nfft, hop, nf, d_out.value, out_stride)
                else:
                    for sgi in range(n_seg):
                        e.process_device(in_fmt, d_in.value + sgi * seg_stride, nfft, hop, nf,
                                         d_out.value + 4 * sgi * out_stride)
            e.synchronize()
            got = np.empty(out_stride * n_seg, dtype=np.float32)
            nat.check(nat.lib.tdsa_memcpy_d2h(0, got.ctypes.data_as(C.c_void_p), d_out, got.nbytes))
            return got, e.hold(), e.dc_estimate, e.info().frames_held_max

    try:
        seq = run(False)
        bat = run(True)
        with _hackrf_engine(pkg, nfft, 4) as e:               # rows that would overlap are refused, not raced
            with pytest.raises(Exception):
                e.process_device_batch(in_fmt, d_in.value, seg_stride, n_seg, nfft, hop, nf, d_out.value, nfft // 2)
    finally:
        nat.lib.tdsa_dev_free(0, d_in)
        nat.lib.tdsa_dev_free(0, d_out)
    assert np.array_equal(seq[0], bat[0], equal_nan=True)      # the gaps keep their NaN fill in both
    for a, b in zip(seq[1], bat[1]):
        assert np.array_equal(a, b)
    assert seq[2] == bat[2] and seq[3] == bat[3] == 2 * n_seg
    if fmt == "i8":
        gold_src = so.HackrfBranchOracle(nfft, 20e6, precision="gold")
        for sgi in (0, 3, n_seg - 1):
            _check(bat[0][sgi * out_stride: sgi * out_stride + nfft],
                   np.asarray(gold_src.power_levels(so.unpack_iq_int8(caps[sgi]))), f"capture {sgi}")


def test_batched_captures_full_c3_shape(pkg):
    """Four seconds of the C3 shape (4 x 2440 frames of 16384 points, hop N/2) in one launch: the hold trace equals
    the column maximum of all 9760 rows and sampled rows match the gold oracle."""
    import ctypes as C
    nat = pkg._native
    nfft, hop, ns, n_seg = 16384, 8192, 20_000_000, 4
    nf = (ns - nfft) // hop + 1
    base = so.synth_iq_int8(ns, nfft, seed=3)
    d_in, d_out = C.c_void_p(), C.c_void_p()
    nat.check(nat.lib.tdsa_dev_alloc(0, 2 * ns * n_seg, C.byref(d_in)))
    nat.check(nat.lib.tdsa_dev_alloc(0, nf * nfft * 4 * n_seg, C.byref(d_out)))
    try:
        hosts = [base if k == 0 else np.roll(base, 2 * 977 * k) for k in range(n_seg)]
        for k, hst in enumerate(hosts):
            nat.check(nat.lib.tdsa_memcpy_h2d(0, C.c_void_p(d_in.value + 2 * ns * k), hst.ctypes.data_as(C.c_void_p), hst.nbytes))
        with _hackrf_engine(pkg, nfft, nf, hold_max=True) as e:
            e.process_device_batch(nat.IN_I8, d_in.value, 2 * ns, n_seg, ns, hop, nf, d_out.value, nf * nfft)
            mx, _ = e.hold()
            assert e.info().frames_held_max == n_seg * nf
        colmax = None
        for k in range(n_seg):
            got = np.empty((nf, nfft), dtype=np.float32)
            nat.check(nat.lib.tdsa_memcpy_d2h(0, got.ctypes.data_as(C.c_void_p), C.c_void_p(d_out.value + 4 * nf * nfft * k),
                                              got.nbytes))
            colmax = got.max(axis=0) if colmax is None else np.maximum(colmax, got.max(axis=0))
            gold_src = so.HackrfBranchOracle(nfft, 20e6, precision="gold")
            for f in (0, nf - 1):
                x = so.unpack_iq_int8(hosts[k][2 * f * hop: 2 * (f * hop + nfft)])
                _check(got[f], np.asarray(gold_src.power_levels(x)), f"second {k} frame {f}")
        assert np.array_equal(mx, colmax)
    finally:
        nat.lib.tdsa_dev_free(0, d_in)
        nat.lib.tdsa_dev_free(0, d_out)


@pytest.mark.parametrize("group", [1, 3, 5])
def test_long_frame_welch_over_several_rounds(pkg, monkeypatch, group):
    """The column / row rounds of the long-frame path (tdsa_debug_knob big_group segments each, 64 by default: one round for the
    C5 capture): with small rounds the row pass's per-workgroup partial rows are ADDED to from round to round (rounds of
    unequal size, fewer workgroups per row in the last one) and the result must not depend on the round size."""
    nfft, k, cal = 1 << 16, 8, -0.8087054556396822
    iq = so.synth_iq_int8(nfft * k, nfft, seed=17)
    gold, gold_mean = _welch_gold(iq, nfft, k, cal)
    with pkg.SpectrumEngine(nfft, max_frames=k) as e:
        e.set_window(so.rtl_window("hanning", nfft).astype(np.float32))
        e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=-1.0, avg=("lin", k),
                    cal_offset_db=cal)
        e.debug_knob("big_group", group)
        out = e.process(iq, hop=nfft)
        mean, cnt = e.averaged()
    with pkg.SpectrumEngine(nfft, max_frames=k) as e:
        e.set_window(so.rtl_window("hanning", nfft).astype(np.float32))
        e.configure(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR, dc_alpha=-1.0, avg=("lin", k),
                    cal_offset_db=cal)
        one_round = e.process(iq, hop=nfft)
    assert cnt == k and np.max(np.abs(mean - gold_mean) / gold_mean.max()) < 1e-5
    _check(out[0], gold, f"Welch of {k} segments in rounds of {group}")
    assert np.max(np.abs(out[0] - one_round[0])) < 1e-4          # float32 partial sums are grouped differently, no more


def test_batched_captures_without_rows_and_with_tare(pkg):
    """tdsa_process_dev_batch with out_db_dev = NULL (only the hold traces are wanted) and with a tare baseline set:
    the hold traces of the one-launch batch equal those of consecutive calls, with and without rows."""
    import ctypes as C
    nat = pkg._native
    nfft, hop, nf, n_seg = 8192, 4096, 45, 4
    ns = hop * (nf - 1) + nfft
    caps = [so.synth_iq_int8(ns, nfft, seed=900 + s) for s in range(n_seg)]
    blob = np.concatenate(caps)
    base = (np.linspace(-3.0, 3.0, nfft)).astype(np.float32)
    d_in, d_out = C.c_void_p(), C.c_void_p()
    nat.check(nat.lib.tdsa_dev_alloc(0, blob.nbytes, C.byref(d_in)))
    nat.check(nat.lib.tdsa_dev_alloc(0, n_seg * nf * nfft * 4, C.byref(d_out)))
    nat.check(nat.lib.tdsa_memcpy_h2d(0, d_in, blob.ctypes.data_as(C.c_void_p), blob.nbytes))
    try:
        holds = []
        for batched, rows in ((False, True), (True, True), (True, False)):
            with _hackrf_engine(pkg, nfft, nf, hold_max=True, hold_min=True) as e:
                nat.check(nat.lib.tdsa_set_tare_baseline(e._h, base.ctypes.data_as(C.c_void_p), nfft))
                out = d_out.value if rows else None
                if batched:
                    e.process_device_batch(nat.IN_I8, d_in.value, 2 * ns, n_seg, ns, hop, nf, out, nf * nfft)
                else:
                    for s in range(n_seg):
                        e.process_device(nat.IN_I8, d_in.value + 2 * ns * s, ns, hop, nf, d_out.value + 4 * nf * nfft * s)
                holds.append(e.hold())
                if rows:
                    got = np.empty((n_seg * nf, nfft), dtype=np.float32)
                    nat.check(nat.lib.tdsa_memcpy_d2h(0, got.ctypes.data_as(C.c_void_p), d_out, got.nbytes))
                    assert np.array_equal(holds[-1][0], got.max(axis=0)) and np.array_equal(holds[-1][1], got.min(axis=0))
        for h in holds[1:]:
            assert np.array_equal(h[0], holds[0][0]) and np.array_equal(h[1], holds[0][1])
        gold, _, _ = so.hackrf_batch(caps[1], nfft, hop, 20e6, precision="gold")
        _check(got[nf: 2 * nf] + base[None, :], gold, "capture 1, tare added back")
    finally:
        nat.lib.tdsa_dev_free(0, d_in)
        nat.lib.tdsa_dev_free(0, d_out)
