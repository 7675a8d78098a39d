#!/usr/bin/env python3
"""Serial (csdr_chain_process_device on one stream) against pipelined (csdr_chain_submit_device: independent launches on two
alternating streams) throughput of the bench configuration, same inputs, alternating input buffers."""
import os as _os; _os.environ.setdefault("CSDR_DIAG", "1")   # tools are diagnostics: the library's A/B knobs are live here
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import composable_sdr_amd as cs
from composable_sdr_amd import _lib
from synth import synth_cf32_torch

M = int(os.environ.get("STEP_M", "256")); nf = int(os.environ.get("STEP_NF", str(262144 * 256 // M)))
demod = os.environ.get("STEP_DEMOD", "fm"); steps = int(os.environ.get("STEP_STEPS", "60"))
dev = torch.device("cuda", 0)
xs = [synth_cf32_torch(M * nf, M, dev, seed=20260101 + 7919 * i) for i in range(2)]
outs = [torch.empty(M * nf * (1 if demod == "fm" else 2), dtype=torch.float32, device=dev) for _ in range(2)]
for mode in ("serial", "pipelined", "serial", "pipelined"):
    ch = cs.Chain(channels=M, demod=demod, max_frames=nf, flags=_lib.FLAG_QUIET | _lib.FLAG_TIME_KERNELS | _lib.FLAG_TIME_REGION)
    def step(i):
        if mode == "serial": ch.process_device(xs[i & 1].data_ptr(), M * nf, outs[i & 1].data_ptr(), 0)
        else: ch.submit_device(xs[i & 1].data_ptr(), M * nf, outs[i & 1].data_ptr())
    for i in range(4): step(i)
    ch.wait_device(); torch.cuda.synchronize(); ch.kernel_time()
    t0 = time.perf_counter()
    for i in range(steps): step(i)
    ch.wait_device(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    k = ch.kernel_time()
    print(f"{mode:10s}: {dt * 1e6:7.1f} us per step = {M * nf / dt / 1e9:6.1f} GS/s; region timer {k[0]} {k[1] / max(k[2], 1) * 1e3:.1f} us x {k[2]}; independent launches {ch.independent_launches()}", flush=True)
    ch.close()
