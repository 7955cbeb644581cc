#!/usr/bin/env python3
"""Developer tool (round-4 verdict item 6): the parity allowance's worst case, hunted on purpose.

The worst bin every soak has found is the one N/2 away from a full-scale tone that falls exactly on a bin: in the last
radix-2 stage X[k0 + N/2] = E - W O with E = W O = X[k0] / 2, so what is left is the difference of the rounding errors
the two half-amplitude partial sums carry (profiles/HISTORY.md, appendix "(old) 2. Oracle", has the budget).  This tool draws ONLY such cases -
a complex exponential of 100 ... 127 LSB exactly on a random bin, random phase, optionally a little noise, every native
size 64 ... 16384 (and long frames 2^15 ... 2^17 with --long), both branches, per-frame / tracked / no DC removal, the
three windows - and reports, as |dB error| over the parity allowance with ONE float32 rounding unit of the frame's largest
amplitude as its floor (on the deep bins: the amplitude error in units of 2^-24 A_max; the tests allow two units):
  * the error AT the bin k0 + N/2 and the worst error of any bin within 100 dB of the frame maximum, per case;
  * their distribution, a Gaussian fit of the tail and the probability of exceeding the allowance of two units it implies.

python tools/parity_targeted.py [--cases 20000] [--long 200] [--seed 0]
"""
import argparse
import math
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle import spectrum_oracle as so  # noqa: E402
from topdogspectrumanalyser_amd import SpectrumEngine  # noqa: E402

UNIT = 2.0 ** -24


def tone_frames(rng, nfft, nf, amp, noise):
    """nf frames (hop = nfft) of an exactly-on-bin complex exponential, int8 interleaved; returns (iq, k0 per frame)"""
    out = np.empty((nf, 2 * nfft), dtype=np.int8)
    k0s = rng.integers(1, nfft, nf)                    # (not the DC bin: the HackRF branch's mean removal would take the tone out)
    n = np.arange(nfft, dtype=np.float64)
    for f in range(nf):
        ph = rng.uniform(0, 2 * np.pi)
        x = amp * np.exp(1j * (2 * np.pi * ((int(k0s[f]) * n) % nfft) / nfft + ph))
        if noise > 0:
            x = x + noise * (rng.standard_normal(nfft) + 1j * rng.standard_normal(nfft))
        out[f, 0::2] = np.clip(np.rint(x.real), -128, 127)
        out[f, 1::2] = np.clip(np.rint(x.imag), -128, 127)
    return out.reshape(-1), k0s


def err_over_allowance(db_gpu, db_gold):
    """per bin: |dB error| / allowance of oracle.parity_metrics with ONE rounding unit as the floor - 1e-3 dB, or the dB
    worth of 2^-24 of the frame's largest amplitude at that bin's depth where that is more (from 66 dB down).  On the deep
    bins this IS the amplitude error in units of 2^-24 A_max; the tests allow 2."""
    db_gpu, db_gold = np.asarray(db_gpu, np.float64), np.asarray(db_gold, np.float64)
    depth = db_gold.max(axis=-1, keepdims=True) - db_gold
    allowance = np.maximum(1e-3, (20.0 / np.log(10.0)) * UNIT * 10.0 ** (depth / 20.0))
    return np.abs(db_gpu - db_gold) / allowance


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=20000)
    ap.add_argument("--long", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(20260929 + a.seed)
    sizes = [1 << k for k in range(6, 15)]
    per_cfg = 8                                          # frames per call: one plan per configuration, several cases per call
    at_bin, worst, strict, meta = [], [], [], []
    done = 0
    while done < a.cases:
        nfft = int(rng.choice(sizes, p=np.array([1, 1, 1, 1, 2, 2, 3, 3, 6], float) / 20))
        branch = str(rng.choice(["hackrf", "hackrf", "rtl"]))
        dc = float(rng.choice([1.0, 1.0, 0.25])) if branch == "hackrf" else -1.0
        window = "hanning" if branch == "hackrf" else str(rng.choice(["hanning", "hamming", "rectangle"]))
        psd = bool(rng.integers(0, 4) == 0)
        amp = float(rng.choice([127.0, 127.0, 120.0, 100.0]))
        noise = float(rng.choice([0.0, 0.0, 0.3, 1.0]))
        nf = per_cfg
        iq, k0s = tone_frames(rng, nfft, nf, amp, noise)
        fs = 20e6 if branch == "hackrf" else 2e6
        if branch == "hackrf":
            gold, _, _ = so.hackrf_batch(iq, nfft, nfft, fs, use_psd=psd, dc_alpha=dc, precision="gold")
            w = so.hackrf_window(nfft)
            mode = dict(db_mode="pow", power_scale=1.0 / (fs * nfft), log_floor=so.LOG_FLOOR) if psd else \
                dict(db_mode="mag", log_floor=so.LOG_FLOOR)
        else:
            gold, _, _ = so.rtl_batch(iq, nfft, nfft, fs, window=window, use_psd=psd, precision="gold")
            w = so.rtl_window(window, nfft)
            mode = dict(db_mode="pow", power_scale=1.0 / (fs * nfft), log_floor=so.LOG_FLOOR) if psd else \
                dict(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR)
        with SpectrumEngine(nfft, max_frames=nf) as e:
            e.set_window(w)
            e.configure(dc_alpha=dc, **mode)
            out = e.process(iq, hop=nfft, n_frames=nf)
        units = err_over_allowance(out, gold)
        gold = np.asarray(gold, np.float64)
        depth = gold.max(axis=-1, keepdims=True) - gold
        for f in range(nf):
            kb = (int(k0s[f]) + nfft // 2 + nfft // 2) % nfft          # fftshift-ed position of bin k0 + N/2
            at_bin.append(units[f, kb] if depth[f, kb] <= 100.0 else 0.0)
            m = depth[f] <= 100.0
            worst.append(units[f][m].max())
            strict.append(np.abs(np.asarray(out[f], np.float64) - gold[f])[m].max())
            meta.append((nfft, branch, dc, window, amp, noise, float(depth[f, kb])))
        done += nf
    at_bin, worst, strict = np.array(at_bin), np.array(worst), np.array(strict)
    print(f"{len(worst)} on-bin full-scale tone frames, sizes 64 ... 16384 (|dB error| / allowance with a floor of ONE unit 2^-24 A_max; the tests allow 2)")
    for name, u in (("error at the bin k0 + N/2", at_bin), ("worst bin within 100 dB of the maximum", worst)):
        s = np.sort(u)
        q = lambda p: s[int(p * (len(s) - 1))]                  # noqa: E731
        print(f"  {name:40s} median {np.median(s):.3f}  90 % {q(0.9):.3f}  99 % {q(0.99):.3f}  99.9 % {q(0.999):.3f}  "
              f"99.99 % {q(0.9999):.3f}  worst {s[-1]:.3f}  (> 1: {(s > 1).sum()}, > 1.5: {(s > 1.5).sum()}, > 2: {(s > 2).sum()})")
    i = int(np.argmax(worst))
    print(f"  worst case: N = {meta[i][0]}, {meta[i][1]} branch, dc_alpha {meta[i][2]}, {meta[i][3]}, amplitude {meta[i][4]:.0f} LSB, "
          f"noise {meta[i][5]}, bin k0 + N/2 sits {meta[i][6]:.1f} dB below the maximum")
    # tail model: the error at that bin is a sum of many bounded, independent rounding errors -> Gaussian to a good
    # approximation; sigma from the upper quantiles of the worst-bin statistic of the 16384-point cases (the largest sigma)
    for nsel in (4096, 8192, 16384):
        sel = np.array([m[0] == nsel for m in meta])
        if sel.sum() < 100:
            continue
        u = np.sort(worst[sel])
        # half-normal: quantile p of |X| is sigma * sqrt(2) * erfinv(p)
        sig = np.median([u[int(p * (len(u) - 1))] / (math.sqrt(2.0) * _erfinv(p)) for p in (0.9, 0.95, 0.99)])
        p2 = math.erfc(2.0 / (sig * math.sqrt(2.0)))
        print(f"  N = {nsel:5d}: {sel.sum()} frames, worst {u[-1]:.3f}; Gaussian tail fit sigma = {sig:.3f} units -> "
              f"P(one frame's worst bin > 2 units) = {p2:.1e}")
    print(f"  SURVEY 8(d) read literally (|dB| <= 1e-3 on every bin within 100 dB, no allowance): worst plain |dB| error "
          f"{strict.max():.2e} dB, {100.0 * (strict > 1e-3).mean():.1f} % of these frames exceed it")
    # long frames
    if a.long:
        lw = []
        for _ in range(a.long):
            lg = int(rng.integers(15, 18))
            nfft = 1 << lg
            iq, k0s = tone_frames(rng, nfft, 1, 127.0, float(rng.choice([0.0, 0.3])))
            gold, _, _ = so.hackrf_batch(iq, nfft, nfft, 20e6, precision="gold")
            with SpectrumEngine(nfft, max_frames=1) as e:
                e.set_window(so.hackrf_window(nfft))
                e.configure(db_mode="mag", log_floor=so.LOG_FLOOR, dc_alpha=1.0)
                out = e.process(iq, hop=nfft, n_frames=1)
            units = err_over_allowance(out, gold)
            gold = np.asarray(gold, np.float64)
            m = (gold.max() - gold[0]) <= 100.0
            lw.append(units[0][m].max())
        lw = np.sort(np.array(lw))
        print(f"  long frames 2^15 ... 2^17, {len(lw)} frames: worst bin median {np.median(lw):.3f}  99 % {lw[int(0.99 * (len(lw) - 1))]:.3f}  worst {lw[-1]:.3f}")


def _erfinv(p):
    """inverse error function by bisection (scipy is not needed for three quantiles)"""
    lo, hi = 0.0, 6.0
    for _ in range(80):
        mid = 0.5 * (lo + hi)
        if math.erf(mid) < p:
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


if __name__ == "__main__":
    main()
