#!/usr/bin/env python3
"""Developer tool: one GUI tick at the SOURCE level - HackrfSamplesDataSource.get_power_levels() of this package (samples
already in the reservoir, as the reference's tests drive it) next to the numpy restatement of the reference's own arithmetic
for the same frame (oracle/, test infrastructure: the CPU path a maintainer would be replacing), per call on this box."""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import topdogspectrumanalyser_amd as pkg  # noqa: E402
from oracle import spectrum_oracle as so  # noqa: E402


def main():
    for n in [int(a) for a in sys.argv[1:]] or (512, 1024, 2048, 4096, 16384):
        iq = so.synth_iq_int8(65536, n, seed=5)
        x = ((iq[0::2].astype(np.float32) + 1j * iq[1::2].astype(np.float32)) / np.float32(128)).astype(np.complex64)
        src = pkg.HackrfSamplesDataSource(sample_rate=20_000_000, centre_freq=2_450_000_000)
        src.num_samples = n
        src.running = True
        src._allocate_fft_resources()
        ora = so.HackrfBranchOracle(n, 20e6, precision="ref")
        reps = 400

        def tick():
            src._reservoir = x                      # a fresh chunk every tick (the reader thread's job)
            return src.get_power_levels()

        for _ in range(50):
            tick()
        t0 = time.perf_counter()
        for _ in range(reps):
            tick()
        ours = (time.perf_counter() - t0) / reps * 1e6
        fr = x[-n:].copy()
        for _ in range(20):
            ora.power_levels(fr.copy())
        t0 = time.perf_counter()
        for _ in range(reps):
            ora.power_levels(fr.copy())
        ref = (time.perf_counter() - t0) / reps * 1e6
        print(f"N={n:6d}  get_power_levels(): {ours:7.1f} us   numpy restatement of the reference's frame: {ref:7.1f} us")
        src.running = False
        src.stop()


if __name__ == "__main__":
    main()
