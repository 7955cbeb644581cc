#!/usr/bin/env python3
"""Developer tool: many more of the random call sequences of tests/test_gpu_parity.py::test_random_call_sequences
(process in pieces, averaging changes, resets, calibration offset, tare baseline on one plan, checked call by call
against the float64 oracle driven through the same sequence).

python tools/state_soak.py [--trials 300]
"""
import argparse
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import topdogspectrumanalyser_amd as pkg  # noqa: E402
import test_gpu_parity as T  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=300)
    a = ap.parse_args()
    worst_units, worst_rel, n_calls = 0.0, 0.0, 0
    for trial in range(a.trials):
        u, r, c = T.call_sequence_trial(pkg, trial, report=print)
        worst_units, worst_rel, n_calls = max(worst_units, u), max(worst_rel, r), n_calls + c
    print(f"{a.trials} trials, {n_calls} process calls: worst dB error / allowance {worst_units:.2f} rounding units "
          f"(bound 2), worst relative power error {worst_rel:.1e} (bound 1e-4)")


if __name__ == "__main__":
    main()
