import sys, time
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import composable_sdr_amd as cs
from synth import synth_cf32
M, nf = 256, 4096
x = synth_cf32(M * nf, M)
for nfc in (4096, 65536):
    xx = np.tile(x, nfc // nf)
    ch = cs.Chain(channels=M, demod="fm", max_frames=nfc)
    ch.process(xx)
    t = time.perf_counter(); n = 10
    for _ in range(n): ch.process(xx)
    dt = time.perf_counter() - t
    print(f"host-buffer path, {nfc} frames/chunk: {n * xx.size / dt / 1e6:.0f} MS/s ({dt / n * 1e3:.2f} ms per call)")
