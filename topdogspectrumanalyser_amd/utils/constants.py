"""The numeric constants and enums the sample-method hot path reads.

Mirrors the names of the reference's utils/constants.py (DSPConstants :152-155, FFTSize :20-41,
WindowType :68-73, SourceLimits :103-118, UIConstants.TARE_NUM_SAMPLES :141, DisplayMode :6-17) so code
written against the reference keeps working; the 228 menu-button ids are GUI-only and not carried.
"""
from enum import Enum, IntEnum


class DisplayMode(IntEnum):
    TWO_D = 0
    THREE_D = 1
    WATERFALL = 2
    SURFACE = 3
    LOGO = 4
    CONSTELLATION_2D = 5
    CONSTELLATION_3D = 6
    ZERO_SPAN = 7
    RIBBON = 8
    DENSITY = 9


class FFTSize(IntEnum):
    """FFT sizes the reference enumerates (512..8192); the GPU path takes every size up to 2^19 and every power
    of two up to 2^20 (gpu_fft_size_supported)."""
    SIZE_512 = 512
    SIZE_1024 = 1024
    SIZE_2048 = 2048
    SIZE_4096 = 4096
    SIZE_8192 = 8192

    @classmethod
    def is_valid(cls, size: int) -> bool:
        return size in [s.value for s in cls]

    @classmethod
    def get_min(cls) -> int:
        return min(s.value for s in cls)

    @classmethod
    def get_max(cls) -> int:
        return max(s.value for s in cls)


GPU_MIN_FFT = 64
GPU_MAX_FFT = 1 << 20      # 64 .. 16384 in one LDS-resident pass, 2^15 .. 2^20 as N1 x 16384 (two passes)
GPU_MAX_ANY_FFT = 1 << 20  # any size from 2 up to here, power of two or not: chirp-z on the power-of-two kernels (above 2^19 as four
                           # half-length sub-convolutions of 2^20 points)


def gpu_fft_size_supported(nfft: int) -> bool:
    """Sizes the device library has a plan for: every N in [2, 2^20]."""
    nfft = int(nfft)
    if 2 <= nfft <= GPU_MAX_ANY_FFT:
        return True
    return GPU_MIN_FFT <= nfft <= GPU_MAX_FFT and nfft & (nfft - 1) == 0


def gpu_real_input_size_supported(nfft: int) -> bool:
    """Sizes tdsa_process_real2 (the audio source's path) has a plan for: powers of two up to 16384 and any other size
    up to 2^19 (include/tdsa_hip.h: tdsa_real_input_supported)."""
    nfft = int(nfft)
    if nfft < 2:
        return False
    if nfft & (nfft - 1) == 0:
        return nfft <= 16384
    return nfft <= 1 << 19


class WindowType(str, Enum):
    HAMMING = "hamming"
    HANNING = "hanning"
    BLACKMAN = "blackman"
    RECTANGLE = "rectangle"


class SourceLimits:
    RTL_MIN_FREQ = 24e6
    RTL_MAX_FREQ = 1.766e9
    RTL_MAX_SAMPLE_RATE = 2.4e6
    HACKRF_MIN_FREQ = 1e6
    HACKRF_MAX_FREQ = 6e9
    HACKRF_MAX_SAMPLE_RATE = 20e6
    MICROPHONE_MIN_FREQ = 20
    MICROPHONE_MAX_FREQ = 20e3
    MICROPHONE_SAMPLE_RATE = 44100


class UIConstants:
    DEFAULT_FFT_SIZE = FFTSize.SIZE_1024.value
    SWEEP_RATE_UPDATE_INTERVAL = 50
    TARE_NUM_SAMPLES = 32          # frames averaged to build the tare baseline
    BUTTON_ACTIVE_STYLE = "background-color: #666666; color: white; font-weight: bold;"


class DSPConstants:
    LOG_FLOOR = 1e-12              # floor inside 20*log10(|X| + .) and the PSD 10*log10
    POWER_LOG_FLOOR = 1e-10        # floor inside 10*log10(|X|^2 + .)


class SourceType(str, Enum):
    RTL_SWEEP = "rtl_sweep"
    HACKRF_SWEEP = "hackrf_sweep"
    RTL_SAMPLES = "rtl_samples"
    MICROPHONE_SAMPLES = "microphone_samples"
    HACKRF_SAMPLES = "hackrf_samples"


class FrequencyPresets:
    """Default spans the GUI falls back to (utils/constants.py:90-100 of the reference)."""
    HACKRF_DEFAULT_START = 2400e6
    HACKRF_DEFAULT_STOP = 2500e6
    MICROPHONE_DEFAULT_START = 0
    MICROPHONE_DEFAULT_STOP = 22050


def format_hz(hz: float, precision: int = 4) -> str:
    """'98 MHz', '1.42 GHz', '440.0 Hz': engineering prefix, `precision` significant figures
    (utils/frequency_helpers.py:80-97 of the reference; the peak-list read-out uses it)."""
    for scale, unit in ((1e9, "GHz"), (1e6, "MHz"), (1e3, "kHz")):
        if abs(hz) >= scale:
            return f"{hz / scale:.{precision}g} {unit}"
    return f"{hz:.1f} Hz"
