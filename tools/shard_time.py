#!/usr/bin/env python3
"""Per-rank step time of the interleaved channel shards of the fused 256-channel chain (rank 0 of G, same stream on every rank)
beside the whole band, FM and FM + AGC: what an N-GPU channel-sharded run does per GPU."""
import os as _os; _os.environ.setdefault("CSDR_DIAG", "1")   # tools are diagnostics: the library's A/B knobs are live here
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import composable_sdr_amd as cs
from composable_sdr_amd import _lib
from synth import synth_cf32_torch

M = int(os.environ.get("STEP_M", "256")); nf = 262144 * 256 // M; steps = int(os.environ.get("STEP_STEPS", "100"))
dev = torch.device("cuda", 0)
xs = [synth_cf32_torch(M * nf, M, dev, seed=20260101 + 7919 * i) for i in range(2)]
out = torch.empty(M * nf * 2, dtype=torch.float32, device=dev)
for agc in (0.0, 10.0):
    for G in (1, 2, 4, 8):
        kw = dict(channels=M, demod="fm", kf=0.3, agc=agc, max_frames=nf, flags=_lib.FLAG_QUIET)
        if G > 1: kw.update(chan_first=0, chan_stride=G)
        ch = cs.Chain(**kw)
        for i in range(4): ch.process_device(xs[i & 1].data_ptr(), M * nf, out.data_ptr(), 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps): ch.process_device(xs[i & 1].data_ptr(), M * nf, out.data_ptr(), 0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(f"agc {agc:4.1f} G={G}: {dt * 1e6:7.1f} us per step per rank = {M * nf / dt / 1e9:6.1f} GS/s of input  [{ch.path}] {ch.kernel_time()[0]}", flush=True)
        ch.close()
