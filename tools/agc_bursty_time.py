import os as _os; _os.environ.setdefault("CSDR_DIAG", "1")   # tools are diagnostics: the library's A/B knobs are live here
import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import composable_sdr_amd as cs
from composable_sdr_amd import _lib
from synth import channel_centre
M, nf = 64, 262144
rng = np.random.default_rng(5)
n = M * nf
t = np.arange(n)
x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.02).astype(np.complex64)
for k in range(1, M, 8):
    w = channel_centre(k, M)
    gate = np.ones(nf, dtype=np.float32); pos, on = 0, bool(k & 1)
    while pos < nf:
        ln = int(rng.integers(2000, 30000)); gate[pos:pos + ln] = 1.0 if on else 0.0; pos += ln; on = not on
    x += (np.repeat(gate, M) * 0.1 * np.exp(1j * (w * t))).astype(np.complex64)
for flags, name in ((_lib.FLAG_QUIET, "tail"), (_lib.FLAG_QUIET | _lib.FLAG_AGC_SEQUENTIAL, "sequential")):
    a = cs.Chain(channels=M, demod="fm", kf=0.3, agc=8.0, max_frames=nf, flags=flags)
    a.process(x)                      # first call (start transient)
    t0 = time.perf_counter(); y = a.process(x); dt = time.perf_counter() - t0
    st = a.agc_stats() if name == "tail" else None
    print(f"{name}: {dt*1e3:.1f} ms per call of {n/1e6:.1f} M samples (host buffers), stats {st}, checksum {float(np.abs(y).sum()):.6e}")
    a.close()
