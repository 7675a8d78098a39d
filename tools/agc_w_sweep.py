#!/usr/bin/env python3
"""cfg3 + AGC at the bench size: step time and repaired segments against the warm-up length W (CSDR_AGC_W), the numbers behind DESIGN 4.7 (W = 1024)."""
import os as _os; _os.environ.setdefault("CSDR_DIAG", "1")
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import composable_sdr_amd as cs
from composable_sdr_amd import _lib
from synth import synth_cf32_torch

M = int(os.environ.get("STEP_M", "256")); nf = int(os.environ.get("STEP_NF", "262144")); steps = int(os.environ.get("STEP_STEPS", "150"))
dev = torch.device("cuda", 0)
xs = [synth_cf32_torch(M * nf, M, dev, seed=20260101 + 7919 * i) for i in range(3)]
out = torch.empty(M * nf, dtype=torch.float32, device=dev)
for W in [int(w) for w in os.environ.get("WS", "1024,896,768,640,512").split(",")]:
    os.environ["CSDR_AGC_W"] = str(W)
    ch = cs.Chain(channels=M, demod="fm", agc=10.0, max_frames=nf, flags=_lib.FLAG_QUIET)
    for i in range(4):
        ch.process_device(xs[i % 3].data_ptr(), M * nf, out.data_ptr(), 0)
    torch.cuda.synchronize()
    c0, r0 = ch.agc_stats()
    t0 = time.perf_counter()
    for i in range(steps):
        ch.process_device(xs[(i + 1) % 3].data_ptr(), M * nf, out.data_ptr(), 0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    c1, r1 = ch.agc_stats()
    print(f"W={W:5d}: {dt * 1e6:8.1f} us per step; boundaries checked per step {(c1 - c0) // steps}, repaired per step {(r1 - r0) / steps:.1f} [{ch.path}]", flush=True)
    ch.close()
