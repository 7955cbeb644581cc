#!/usr/bin/env python3
"""Context number, not part of the product: the same C3 step written with stock PyTorch-ROCm ops (int8 -> complex64,
mean removal, window, torch.fft.fft = hipFFT/rocFFT, |X|, 20 log10, fftshift, running max) on the same GPU, inputs
resident in HBM.  Shows what the fused frame kernel is worth against the vendor FFT library plus elementwise
kernels.  The product itself never calls rocFFT (BASELINE.json north_star)."""
import time

import numpy as np
import torch


def main():
    nfft, hop, ns = 16384, 8192, 20_000_000
    nf = (ns - nfft) // hop + 1
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    iq = torch.from_numpy(rng.integers(-100, 100, size=2 * ns, dtype=np.int8)).to(dev)
    win = torch.from_numpy(np.hanning(nfft).astype(np.float32)).to(dev)
    hold = torch.full((nfft,), -float("inf"), device=dev)

    def step():
        nonlocal hold
        x = iq.view(-1, 2).to(torch.float32) * (1.0 / 128.0)
        z = torch.view_as_complex(x)                                   # [ns]
        frames = z.unfold(0, nfft, hop)                                # [nf, nfft] view, 50 % overlap
        frames = (frames - frames.mean(dim=1, keepdim=True)) * win
        spec = torch.fft.fftshift(torch.fft.fft(frames, dim=1), dim=1)
        db = 20.0 * torch.log10(spec.abs() + 1e-12)
        hold = torch.maximum(hold, db.max(dim=0).values)
        return db

    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t_end = time.perf_counter() + 1.0                                   # clocks settled
    while time.perf_counter() < t_end:
        step()
    torch.cuda.synchronize()
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / reps
    print(f"stock torch ops (hipFFT + elementwise), C3 shape, {nf} frames per step: {per * 1e6:.1f} us per step "
          f"-> {nf / per / 1e6:.2f} M frames/s")
    # the library FFT alone, on frames already unpacked, windowed and laid out contiguously
    x = iq.view(-1, 2).to(torch.float32)
    frames = torch.view_as_complex(x).unfold(0, nfft, hop).contiguous()
    for _ in range(50):
        torch.fft.fft(frames, dim=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        torch.fft.fft(frames, dim=1)
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / reps
    print(f"torch.fft.fft alone on {nf} x {nfft} complex64 (contiguous, out of place): {per * 1e6:.1f} us "
          f"-> {nf / per / 1e6:.2f} M frames/s, {2 * nf * nfft * 8 / per / 1e12:.2f} TB/s of its own 640 MB")


if __name__ == "__main__":
    main()
