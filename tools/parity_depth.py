#!/usr/bin/env python3
"""Developer tool: |dB| error of the GPU path and of the reference-precision (float32) oracle against the
float64 gold, binned by depth below the frame maximum (1e-3 dB over a 100 dB window)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import spectrum_oracle as so  # noqa: E402
from topdogspectrumanalyser_amd import SpectrumEngine  # noqa: E402


def main():
    for nfft, hop, nf in ((16384, 8192, 64), (4096, 4096, 64), (1024, 1024, 64)):
        iq = so.synth_iq_int8(hop * (nf - 1) + nfft, nfft, seed=3)
        gold, _, _ = so.hackrf_batch(iq, nfft, hop, 20e6, precision="gold")
        ref, _, _ = so.hackrf_batch(iq, nfft, hop, 20e6, precision="ref")
        with SpectrumEngine(nfft, max_frames=nf) as e:
            e.set_window(so.hackrf_window(nfft))
            e.configure(db_mode="mag", log_floor=so.LOG_FLOOR, dc_alpha=1.0)
            out = e.process(iq, hop=hop)
        depth = gold.max(axis=1, keepdims=True) - gold
        eg, er = np.abs(out - gold), np.abs(np.asarray(ref, dtype=np.float64) - gold)
        print(f"N={nfft}: {nf} frames; max depth {depth.max():.1f} dB")
        for lo, hi in ((0, 40), (40, 60), (60, 70), (70, 80), (80, 90), (90, 100), (100, 400)):
            m = (depth >= lo) & (depth < hi)
            if m.any():
                print(f"  depth {lo:3d}-{hi:3d} dB: bins {m.sum():8d}  gpu max {eg[m].max():.2e} rms {np.sqrt((eg[m]**2).mean()):.2e}"
                      f"   ref32 max {er[m].max():.2e} rms {np.sqrt((er[m]**2).mean()):.2e}")




def top_errors():
    """where the largest |dB| errors sit: bin index, index mod 512, depth (frame 0 of the C3 shape)"""
    nfft, hop, nf = 16384, 8192, 4
    iq = so.synth_iq_int8(hop * (nf - 1) + nfft, nfft, seed=3)
    gold, _, _ = so.hackrf_batch(iq, nfft, hop, 20e6, precision="gold")
    ref, _, _ = so.hackrf_batch(iq, nfft, hop, 20e6, precision="ref")
    with SpectrumEngine(nfft, max_frames=nf) as e:
        e.set_window(so.hackrf_window(nfft))
        e.configure(db_mode="mag", log_floor=so.LOG_FLOOR, dc_alpha=1.0)
        out = e.process(iq, hop=hop)
    a_gold = 10.0 ** (gold[0] / 20.0)
    a_gpu = 10.0 ** (out[0].astype(np.float64) / 20.0)
    a_ref = 10.0 ** (np.asarray(ref[0], dtype=np.float64) / 20.0)
    amax = a_gold.max()
    eg, er = np.abs(a_gpu - a_gold) / amax, np.abs(a_ref - a_gold) / amax
    print(f"amplitude error / A_max, frame 0: gpu rms {np.sqrt((eg**2).mean()):.2e} max {eg.max():.2e}; "
          f"ref32 rms {np.sqrt((er**2).mean()):.2e} max {er.max():.2e}")
    order = np.argsort(eg)[::-1][:24]
    for k in order:
        kk = (int(k) + nfft // 2) % nfft        # natural (unshifted) bin
        print(f"  shifted bin {int(k):6d} natural {kk:6d} mod512 {kk % 512:4d} mod1024 {kk % 1024:5d} depth "
              f"{gold[0].max() - gold[0][k]:6.1f} dB  gpu err {eg[k]:.2e}  ref32 err {er[k]:.2e}")
    # by residue class of the natural bin index
    nat = (np.arange(nfft) + nfft // 2) % nfft
    for m in (16, 32, 512, 1024):
        cls = np.array([np.sqrt((eg[nat % m == r] ** 2).mean()) for r in range(m)])
        print(f"  rms error by (natural bin mod {m}): min {cls.min():.2e} median {np.median(cls):.2e} max {cls.max():.2e} "
              f"argmax {int(cls.argmax())}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "top":
        top_errors()
    else:
        main()
