"""Synthetic SDR-shaped IQ: three complex tones at bins
{N/8, -N/5 + 0.3, 3N/7 + 0.5} with amplitudes 40 / 12 / 3 LSB, a (2 + 1j) LSB DC offset and complex
Gaussian noise (sigma = 4 LSB), rounded and clipped to interleaved int8 [I0, Q0, I1, Q1, ...].
Used by bench.py and by the replay data sources when no SDR hardware is attached."""
import numpy as np


def synth_iq_int8(n_samples: int, nfft: int, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    n = np.arange(n_samples, dtype=np.float64)
    bins = (nfft / 8 + 0.0, -nfft / 5 + 0.3, 3 * nfft / 7 + 0.5)
    amps = (40.0, 12.0, 3.0)
    sig = np.zeros(n_samples, dtype=np.complex128)
    for b, a in zip(bins, amps):
        sig += a * np.exp(2j * np.pi * b * n / nfft)
    sig += (2.0 + 1.0j)
    sig += 4.0 * (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples)) / np.sqrt(2.0)
    out = np.empty(2 * n_samples, dtype=np.int8)
    out[0::2] = np.clip(np.rint(sig.real), -128, 127).astype(np.int8)
    out[1::2] = np.clip(np.rint(sig.imag), -128, 127).astype(np.int8)
    return out
