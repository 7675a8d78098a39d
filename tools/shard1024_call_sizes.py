#!/usr/bin/env python3
"""k_shard1024 against the route it replaced (CSDR_NO_SHARD1024=1: k_run1024v2<FM, G> / whole band + gather) by call size, rank 0 of 8."""
import os as _os; _os.environ.setdefault("CSDR_DIAG", "1")
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import composable_sdr_amd as cs
from composable_sdr_amd import _lib
from synth import synth_cf32_torch
M, G = 1024, 8
dev = torch.device("cuda", 0)
x = synth_cf32_torch(M * 65536, M, dev, seed=5)
out = torch.empty(M * 65536 // G * 2, dtype=torch.float32, device=dev)
for demod in ("fm", "none"):
    for nf in (1024, 4096, 16384, 65536):
        ch = cs.Chain(channels=M, demod=demod, kf=0.3, max_frames=nf, flags=_lib.FLAG_QUIET, chan_first=0, chan_stride=G)
        for i in range(20): ch.process_device(x.data_ptr(), M * nf, out.data_ptr(), 0)
        torch.cuda.synchronize()
        n = max(50, int(0.3 / (nf * 3e-9 + 2e-5)))
        t0 = time.perf_counter()
        for i in range(n): ch.process_device(x.data_ptr(), M * nf, out.data_ptr(), 0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"NO_SHARD1024={os.environ.get('CSDR_NO_SHARD1024', '0')} {demod:4s} nf={nf:6d}: {dt * 1e6:8.1f} us per call = {M * nf / dt / 1e9:6.1f} GS/s  {ch.kernel_time()[0]}", flush=True)
        ch.close()
