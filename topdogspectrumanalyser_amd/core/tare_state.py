"""Bookkeeping record of a tare (baseline normalisation) run.

A display manager flips `collecting` on when the user asks for a new baseline and DataProcessor counts the
frames that went into it; on this build the running sum itself lives on the GPU (TraceState), so
`buffer` only exists because code written against the reference's record of the same name
(core/tare_state.py) reads and clears it.
"""
from dataclasses import dataclass, field
from typing import Any, Optional


@dataclass
class TareState:
    count: int = 0                                   # frames folded into the baseline so far
    collecting: bool = False                         # a baseline run is in progress
    buffer: Optional[Any] = field(default=None, repr=False)

    def begin(self) -> None:
        """Start a fresh baseline run."""
        self.count, self.collecting, self.buffer = 0, True, None

    def finish(self) -> None:
        """The baseline is complete (or was abandoned): back to idle."""
        self.count, self.collecting, self.buffer = 0, False, None

    @property
    def idle(self) -> bool:
        return not self.collecting
