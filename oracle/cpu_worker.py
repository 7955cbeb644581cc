"""One worker of bench.py's all-cores CPU baseline leg (SURVEY.md 8(d) "CPU baseline (ii)").
TEST / MEASUREMENT INFRASTRUCTURE ONLY: runs the numpy restatement of get_power_levels on its own slice
of synthetic IQ for a fixed wall-clock budget and prints `frames seconds` on stdout.

    python -m oracle.cpu_worker <branch> <nfft> <hop> <fs> <seconds> <seed>
"""
import sys
import time

import numpy as np

from oracle import spectrum_oracle as so


def main() -> None:
    branch, nfft, hop, fs, seconds, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), \
        float(sys.argv[5]), int(sys.argv[6])
    frames = 64
    iq = so.synth_iq_int8(hop * (frames - 1) + nfft, nfft, seed=seed)
    br = (so.HackrfBranchOracle if branch == "hackrf" else so.RtlBranchOracle)(nfft, fs, precision="ref")
    br.power_levels(so.unpack_iq_int8(iq[: 2 * nfft]))                      # warm caches / plans
    print("ready", flush=True)
    sys.stdin.readline()                                                      # start gun
    done = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        k = done % frames
        br.power_levels(so.unpack_iq_int8(iq[2 * k * hop: 2 * (k * hop + nfft)]))
        done += 1
    print(done, time.perf_counter() - t0, flush=True)


if __name__ == "__main__":
    main()
