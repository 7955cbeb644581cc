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


def process_sharded(iq: np.ndarray, nfft: int, hop: int, devices: Sequence[int], window: np.ndarray,
                    hold: str = "", **configure):
    """One capture over several GPUs from ONE process: a thread and a SpectrumEngine per entry of `devices`
    (the C-ABI calls release the GIL), each taking its contiguous frame range plus halo; returns
    (dB rows [frames, nfft] in capture order, combined max-hold trace or None, combined min-hold or None).

    Only order-independent modes can be sharded: no "exp" / capped "lin" averaging and no tracked DC remover
    (dc_alpha in [0, 1) carries dc_state across frames) - those raise ValueError: replicas only.  A
    device may appear more than once - two plans then share that GPU - which is also how this is tested
    on a single-GPU box.  `configure` goes to SpectrumEngine.configure; `hold` is "", "max", "min" or "maxmin".
    """
    import threading
    from .engine import SpectrumEngine

    if configure.get("avg", ("off", 1))[0] != "off":
        raise ValueError("order-dependent averaging cannot be sharded (SURVEY.md 8(e): replicas only)")
    if 0.0 <= float(configure.get("dc_alpha", 1.0)) < 1.0:
        raise ValueError("the tracked DC remover (0 <= dc_alpha < 1) carries state from frame to frame and cannot "
                         "be sharded: use dc_alpha >= 1 (per-frame mean) or < 0 (off)")
    iq = np.ascontiguousarray(iq)
    per_sample = 1 if np.iscomplexobj(iq) else 2           # interleaved bytes: two array elements per sample
    n_samples = iq.size // per_sample
    n_frames = 0 if n_samples < nfft else (n_samples - nfft) // hop + 1
    world = len(devices)
    rows = np.empty((n_frames, nfft), dtype=np.float32)
    holds = [None] * world
    errors = [None] * world

    def work(rank: int) -> None:
        try:
            f0, f1 = shard_frames(n_frames, rank, world)
            if f1 <= f0:
                return
            s0, s1 = shard_samples(f0, f1, nfft, hop)
            with SpectrumEngine(nfft, max_frames=f1 - f0, device=devices[rank]) as eng:
                eng.set_window(window)
                eng.configure(hold_max="max" in hold, hold_min="min" in hold, **configure)
                rows[f0:f1] = eng.process(iq[per_sample * s0: per_sample * s1], hop=hop, n_frames=f1 - f0)
                holds[rank] = eng.hold()
        except Exception as exc:                          # surfaced by the caller's thread
            errors[rank] = exc

    threads = [threading.Thread(target=work, args=(r,), name=f"tdsa-shard-{r}") for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for exc in errors:
        if exc is not None:
            raise exc
    got = [h for h in holds if h is not None]
    mx = combine_hold([h[0] for h in got], "max") if "max" in hold and got else None
    mn = combine_hold([h[1] for h in got], "min") if "min" in hold and got else None
    return rows, mx, mn
