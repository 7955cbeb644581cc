#!/usr/bin/env python3
"""Developer tool: the seeded random configuration sweep of tests/test_gpu_parity.py over many more cases, plus long
frames; prints, per class, the distribution of the worst dB error of a case in units of the parity allowance of
oracle.spectrum_oracle.parity_metrics (1e-3 dB, or ONE float32 rounding unit 2^-24 * A_max where that is worth more).

python tools/parity_soak.py [--cases 1500] [--long 60]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import spectrum_oracle as so  # noqa: E402
from topdogspectrumanalyser_amd import SpectrumEngine  # noqa: E402
import test_gpu_parity as T  # noqa: E402


def run_case(c):
    nfft, nf, hop = c["nfft"], c["nf"], c["hop"]
    iq = so.synth_iq_int8(hop * (nf - 1) + nfft, nfft, seed=c["seed"])
    fs = 20e6 if c["branch"] == "hackrf" else 2e6
    averaging = c["avg"][0] != "off" and c["avg"][1] > 1
    if c["branch"] == "hackrf":
        gold, _, _ = so.hackrf_batch(iq, nfft, hop, fs, use_psd=c["psd"], avg=c["avg"], dc_alpha=c["dc_alpha"],
                                     cal_offset_db=c["cal"], precision="gold")
        window, dc = so.hackrf_window(nfft), c["dc_alpha"]
    else:
        gold, _, _ = so.rtl_batch(iq, nfft, hop, fs, window=c["window"], use_psd=c["psd"], avg=c["avg"],
                                  cal_offset_db=c["cal"], precision="gold")
        window, dc = so.rtl_window(c["window"], nfft), -1.0
    if c["psd"]:
        mode = dict(db_mode="pow", power_scale=1.0 / (fs * nfft), log_floor=so.LOG_FLOOR)
    elif averaging or c["branch"] == "rtl":
        mode = dict(db_mode="pow", power_scale=1.0, log_floor=so.POWER_LOG_FLOOR)
    else:
        mode = dict(db_mode="mag", log_floor=so.LOG_FLOOR)
    with SpectrumEngine(nfft, max_frames=nf) as e:
        e.set_window(window)
        e.configure(dc_alpha=dc, avg=c["avg"], cal_offset_db=c["cal"], **mode)
        if nfft > 16384 and nfft & (nfft - 1) == 0:       # native long-frame plans: one frame per call
            out = np.concatenate([e.process(iq[2 * hop * k: 2 * (hop * k + nfft)], hop=nfft, n_frames=1) for k in range(nf)])
        else:
            out = e.process(iq, hop=hop, n_frames=nf)
    rel, ddb = so.parity_metrics(out, gold)
    return rel, ddb / 1e-3, dc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=1500)
    ap.add_argument("--long", type=int, default=60)
    ap.add_argument("--longany", type=int, default=60, help="long frames that are not a power of two (8193 .. 300000 points)")
    ap.add_argument("--seed", type=int, default=0, help="offset added to every generator's seed (0: the seeds of the committed records)")
    a = ap.parse_args()
    classes = {"N <= 16384": [], "N <= 16384, tracked DC": [], "long frames": [], "long frames, not a power of two": []}
    worst_rel = 0.0
    for i in range(a.cases):
        c = T._random_case(np.random.default_rng(4242 + a.seed + i))
        rel, units, dc = run_case(c)
        worst_rel = max(worst_rel, rel)
        classes["N <= 16384, tracked DC" if 0.0 <= dc < 1.0 else "N <= 16384"].append(units)
    rng = np.random.default_rng(77 + a.seed)
    for i in range(a.long):
        lg = int(rng.integers(15, 19))
        c = dict(nfft=1 << lg, nf=int(rng.integers(1, 4)), hop=1 << lg, branch=str(rng.choice(["hackrf", "rtl"])),
                 avg=("off", 1), psd=bool(rng.integers(0, 2)), dc_alpha=float(rng.choice([1.0, 1.0, 0.25])),
                 cal=0.0, window=str(rng.choice(["hanning", "hamming", "rectangle"])), seed=int(rng.integers(1, 1 << 30)))
        rel, units, dc = run_case(c)
        worst_rel = max(worst_rel, rel)
        classes["long frames"].append(units)
    rng = np.random.default_rng(78 + a.seed)
    for i in range(a.longany):
        while True:
            n = int(rng.choice([rng.integers(8193, 20000), rng.integers(20000, 70000), rng.integers(70000, 300000)]))
            if n & (n - 1):
                break
        c = dict(nfft=n, nf=int(rng.integers(1, 4)), hop=int(rng.choice([n, n // 2])), branch=str(rng.choice(["hackrf", "rtl"])),
                 avg=[("off", 1), ("exp", 3)][int(rng.integers(0, 2))], psd=bool(rng.integers(0, 2)),
                 dc_alpha=float(rng.choice([1.0, 1.0, 0.25])), cal=0.0,
                 window=str(rng.choice(["hanning", "hamming", "rectangle"])), seed=int(rng.integers(1, 1 << 30)))
        rel, units, dc = run_case(c)
        worst_rel = max(worst_rel, rel)
        classes["long frames, not a power of two"].append(units)
    # real-input (audio) path: tdsa_process_real2 against the float64 restatement of audio_samples.py:121-131
    classes["real input (audio)"] = []
    rng = np.random.default_rng(99 + a.seed)
    for i in range(a.cases // 10):
        n = int(2 ** rng.integers(6, 15))
        nf = int(rng.integers(1, 9))
        chan = str(rng.choice(["mono", "left", "right", "stereo"]))
        psd = bool(rng.integers(0, 2))
        fs = 44100
        amp = float(10.0 ** rng.uniform(-3, 0))
        t = np.arange(n * nf)
        st = np.stack([amp * np.sin(2 * np.pi * (n / 7.3) * t / n) + 0.01 * amp * rng.standard_normal(n * nf) + 0.05 * amp,
                       0.5 * amp * np.sin(2 * np.pi * (n / 3.1) * t / n) + 0.02 * amp * rng.standard_normal(n * nf)],
                      axis=1).astype(np.float32)
        win = so.rtl_window(str(rng.choice(["hanning", "hamming", "rectangle"])), n)
        with SpectrumEngine(n, max_frames=nf) as e:
            e.set_window(win.astype(np.float32))
            e.configure(db_mode="pow", power_scale=(1.0 / (fs * n)) if psd else 1.0,
                        log_floor=so.LOG_FLOOR if psd else so.POWER_LOG_FLOOR, dc_alpha=1.0)
            out = e.process_real2(st, chan)
        worst = 0.0
        for k in range(nf):
            blk = st[k * n:(k + 1) * n].astype(np.float64)
            sigs = {"mono": [(blk[:, 0] + blk[:, 1]) * 0.5], "left": [blk[:, 0]], "right": [blk[:, 1]],
                    "stereo": [blk[:, 0], blk[:, 1]]}[chan]
            for ci, sig in enumerate(sigs):
                gold = so.audio_db(so.audio_compute_power(sig, win, n, fs, psd, precision="gold"), psd)
                got = out[k, ci] if chan == "stereo" else out[k]
                rel, ddb = so.parity_metrics(got, gold)
                worst_rel = max(worst_rel, rel)
                worst = max(worst, ddb / 1e-3)
        classes["real input (audio)"].append(worst)
    print(f"worst relative power error of any case: {worst_rel:.2e} (bound 1e-4)")
    for name, u in classes.items():
        if u:
            u = np.sort(np.array(u))
            print(f"{name:26s} {len(u):5d} cases: dB error / allowance  median {np.median(u):.2f}  95 % {u[int(0.95 * (len(u) - 1))]:.2f}"
                  f"  99 % {u[int(0.99 * (len(u) - 1))]:.2f}  worst {u[-1]:.2f}  (> 1: {(u > 1).sum()})")


if __name__ == "__main__":
    main()
