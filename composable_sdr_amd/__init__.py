"""MI355X-native DSP chain behind ComposableSDR's Pipe API.

Thin ctypes binding over libcsdr_hip.so (C ABI in include/csdr.h; hand-written HIP
kernels for gfx950).  There is no CPU fallback: importing works without a GPU (so the
symbol table can be checked), but creating any object without one raises CsdrError.
"""
from ._lib import CsdrError, lib, lib_path, build_library, DEMOD_NONE, DEMOD_FM, DEMOD_AM, DEMOD_WBFM  # noqa: F401
from .pipes import (  # noqa: F401
    Pipe, compose, Chain, ChainConfig, dcBlocker, mixDown, mixUp, automaticGainControl,
    fmDemodulator, amDemodulator, resampler, iirFilter, firDecimator, wbFMDemodulator, firpfbchChannelizer,
)
from .trans import compact, takeNArr, mix, mux, distribute_, addPipe  # noqa: F401
