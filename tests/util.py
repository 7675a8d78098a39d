import numpy as np


def rel_rms(a, b):
    a = np.asarray(a).astype(np.complex128 if np.iscomplexobj(a) else np.float64)
    b = np.asarray(b).astype(a.dtype)
    den = np.sqrt(np.mean(np.abs(b) ** 2)) + 1e-30
    return float(np.sqrt(np.mean(np.abs(a - b) ** 2)) / den)


def max_abs_err(a, b):
    return float(np.max(np.abs(np.asarray(a).astype(np.complex128) - np.asarray(b)))) if np.size(a) else 0.0


def wrap_pm(x, period):
    """wrap differences of a quantity that is only defined modulo `period`"""
    return (x + period / 2) % period - period / 2


def knob(monkeypatch, name, value):
    """Set one of the library's diagnostics knobs (CSDR_RUN_MIN_TILES, CSDR_AGC_W, ...: DESIGN.md 6.1) for a test that forces a kernel
    onto a small input: the library reads them only next to CSDR_DIAG=1 (csrc/csdr_internal.h diag_env)."""
    monkeypatch.setenv("CSDR_DIAG", "1")
    monkeypatch.setenv(name, value)
