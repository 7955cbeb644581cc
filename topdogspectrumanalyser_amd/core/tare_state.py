"""TareState record shared between a display manager and DataProcessor (reference core/tare_state.py:9-13)."""
from dataclasses import dataclass
from typing import Optional

import numpy as np


@dataclass
class TareState:
    collecting: bool = False
    buffer: Optional[np.ndarray] = None     # kept for interface parity; the accumulator lives on the GPU
    count: int = 0
