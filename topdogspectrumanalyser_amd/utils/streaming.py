"""Batch streaming helper over the pinned host pipeline (tdsa_pipe_*).

`stream_spectra` is what a recorder / offline analyser uses instead of calling get_power_levels() once
per 20 ms tick (core/ui_setup.py:60-61 of the reference): every chunk of interleaved int8 IQ goes
straight into a pinned slot, the copies of neighbouring chunks overlap the frame kernel, and the dB rows
come back in submission order.  Framing is the deterministic batch framing of SURVEY.md 8(a) a2
(frame k = iq[k*hop : k*hop + N]) applied per chunk.
"""
from typing import Iterable, Iterator, Optional, Tuple

import numpy as np

from ..engine import SpectrumEngine


def stream_spectra(engine: SpectrumEngine, chunks: Iterable[np.ndarray], hop: Optional[int] = None,
                   n_slots: int = 3, rows=True, copy: bool = True,
                   levels: Optional[Tuple[float, float]] = None) -> Iterator[Optional[np.ndarray]]:
    """Yield the dB rows [n_frames, N] of every chunk (None per chunk when rows=False: hold / averager
    state only; rows="u8": uint8 rows under `levels` = (min_db, max_db), what ImageItem.setImage(rows, levels=...)
    makes of them - a quarter of the read-back, which bounds the pipe).  `chunks` are 1-D int8 arrays of interleaved I,Q; each must hold at least N samples and at
    most the first chunk's length.  With copy=False the yielded array is a view of pinned memory that is
    valid until `n_slots - 1` further chunks have been submitted."""
    it = iter(chunks)
    try:
        first = np.ascontiguousarray(next(it), dtype=np.int8)
    except StopIteration:
        return
    n = engine.nfft
    hop = int(hop or n)
    slot_samples = first.size // 2
    if slot_samples < n:
        raise ValueError(f"chunk of {slot_samples} samples is shorter than one {n}-point frame")
    max_frames = (slot_samples - n) // hop + 1
    if max_frames > engine.max_frames:
        raise ValueError(f"chunk holds {max_frames} frames, engine was created for {engine.max_frames}")

    def emit(q):
        r = q.collect_u8() if rows == "u8" else q.collect()
        return None if r is None else (r.copy() if copy else r)

    with engine.pipe(slot_samples, n_slots=n_slots, rows=rows, levels=levels) as q:
        chunk = first
        while chunk is not None:
            ns = chunk.size // 2
            if ns < n or ns > slot_samples:
                raise ValueError(f"chunk of {ns} samples outside [{n}, {slot_samples}]")
            if q.pending == n_slots:
                yield emit(q)
            q.acquire()[: chunk.size] = chunk
            q.submit(ns, hop, (ns - n) // hop + 1)
            nxt = next(it, None)
            chunk = None if nxt is None else np.ascontiguousarray(nxt, dtype=np.int8)
        while q.pending:
            yield emit(q)
