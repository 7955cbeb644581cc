"""Frame sharding across the GPUs of one node.

Frames are independent, so rank r of W gets the contiguous frame range [r*F/W, (r+1)*F/W) and reads
the matching slice of the IQ stream (extended by nfft - hop samples of halo when frames overlap).
There is NO collective in the data path.  Only the tiny per-GPU trace state is combined afterwards, on
the host: max/min hold with fmax/fmin (associative), Welch / uncapped "lin" averages from per-GPU
float64 means and frame counts.  The "exp" and capped "lin" recurrences are order dependent: they run
on one GPU per trace ("replicas only").
"""
from typing import Sequence, Tuple

import numpy as np


def shard_frames(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """[f0, f1) of `rank`; ranges are contiguous, disjoint, cover [0, n_frames), sizes differ by <= 1."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} of world {world}")
    return (rank * n_frames) // world, ((rank + 1) * n_frames) // world


def shard_samples(f0: int, f1: int, nfft: int, hop: int) -> Tuple[int, int]:
    """Complex-sample range [s0, s1) that frames f0..f1-1 read (empty shard -> (s, s))."""
    if f1 <= f0:
        return f0 * hop, f0 * hop
    return f0 * hop, (f1 - 1) * hop + nfft


def combine_hold(parts: Sequence[np.ndarray], kind: str) -> np.ndarray:
    """Per-GPU hold traces -> one trace (np.fmax / np.fmin: NaN ignored, as the reference's hold)."""
    parts = [p for p in parts if p is not None]
    if not parts:
        raise ValueError("no hold traces to combine")
    op = {"max": np.fmax, "min": np.fmin}[kind]
    return op.reduce(np.stack(parts), axis=0)


def combine_welch(means: Sequence[np.ndarray], counts: Sequence[int]) -> Tuple[np.ndarray, int]:
    """Per-GPU running means of linear power (float64) + frame counts -> overall mean, total count."""
    total = int(sum(counts))
    if total == 0:
        raise ValueError("no frames averaged")
    acc = np.zeros_like(np.asarray(means[0], dtype=np.float64))
    for m, c in zip(means, counts):
        if c:
            acc += np.asarray(m, dtype=np.float64) * c
    return acc / total, total
