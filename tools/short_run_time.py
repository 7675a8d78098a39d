#!/usr/bin/env python3
"""The 20-step timed region of bench.py after different amounts of untimed pre-heating (the board's clock / power state takes
tens of milliseconds of load to settle: a 25-launch run sits entirely inside that transient)."""
import os as _os; _os.environ.setdefault("CSDR_DIAG", "1")   # tools are diagnostics: the library's A/B knobs are live here
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import composable_sdr_amd as cs
from composable_sdr_amd import _lib
from synth import synth_cf32_torch
M, nf = 256, 262144
dev = torch.device("cuda", 0)
xs = [synth_cf32_torch(M * nf, M, dev, seed=20260101 + 7919 * i) for i in range(2)]
out = torch.empty(M * nf, dtype=torch.float32, device=dev)
for pre in (0, 50, 200, 800, 0, 800):
    ch = cs.Chain(channels=M, demod="fm", max_frames=nf, flags=_lib.FLAG_QUIET | _lib.FLAG_TIME_KERNELS)
    time.sleep(0.5)                                   # the idle gap a fresh process has in front of its first launch
    for i in range(pre): ch.process_device(xs[i & 1].data_ptr(), M * nf, out.data_ptr(), 0)
    for i in range(5): ch.process_device(xs[i & 1].data_ptr(), M * nf, out.data_ptr(), 0)
    torch.cuda.synchronize(); ch.kernel_time()
    t0 = time.perf_counter()
    for i in range(20): ch.process_device(xs[i & 1].data_ptr(), M * nf, out.data_ptr(), 0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    k = ch.kernel_time()
    print(f"pre-heat {pre:4d} steps: 20 timed steps at {dt * 1e6:6.1f} us per step; per-launch events {k[1] / k[2] * 1e3:6.1f} us", flush=True)
    ch.close()
