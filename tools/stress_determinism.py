#!/usr/bin/env python3
"""Race hunt: two handles of the same configuration process the same chunks, one after the other on the default stream, the second one
while a memory-bound kernel runs on a side stream; every output is compared BIT FOR BIT with its twin.  The run kernels reuse LDS
buffers across barriers, count vmcnt by hand and stage output in registers: a missing wait shows up here as a mismatch that moves
from run to run."""
import os as _os; _os.environ.setdefault("CSDR_DIAG", "1")   # tools are diagnostics: the library's A/B knobs are live here
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import composable_sdr_amd as cs
from composable_sdr_amd import _lib
from synth import synth_cf32_torch

dev = torch.device("cuda", 0)
steps = int(os.environ.get("STEPS", "150"))
side = torch.cuda.Stream()
junk = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
bad = 0
for M, nf, demod, agc in ((1024, 65536, "fm", 0.0), (1024, 65536, "none", 0.0), (1024, 4096, "fm", 0.0), (1024, 16384, "fm", 10.0), (256, 262144, "fm", 0.0),
                          (256, 262144, "fm", 10.0), (256, 4096, "fm", 0.0), (64, 1048576, "none", 0.0), (64, 4096, "none", 0.0),
                          # round 5: the fused 4096-channel route, the channel-packed row-major AGC tail, short AGC calls
                          (4096, 16384, "none", 0.0), (4096, 16384, "fm", 0.0), (4096, 4096, "fm", 10.0), (256, 4096, "fm", 10.0), (256, 1024, "none", 10.0),
                          (1024, 4096, "fm", 10.0)):
    xs = [synth_cf32_torch(M * nf, M, dev, seed=900 + i) for i in range(3)]
    width = 1 if demod == "fm" else 2
    oa = torch.empty(M * nf * width, dtype=torch.float32, device=dev); ob = torch.empty_like(oa)
    kw = dict(channels=M, demod=demod, kf=0.3, agc=agc, max_frames=nf, flags=_lib.FLAG_QUIET)
    a, b = cs.Chain(**kw), cs.Chain(**kw)
    mism = 0
    t0 = time.time()
    for i in range(steps):
        x = xs[i % 3]
        a.process_device(x.data_ptr(), M * nf, oa.data_ptr(), 0)
        with torch.cuda.stream(side):
            junk.add_(1)                                   # a bandwidth hog beside the second handle's kernels
        b.process_device(x.data_ptr(), M * nf, ob.data_ptr(), 0)
        torch.cuda.synchronize()
        if not torch.equal(oa.view(torch.int32), ob.view(torch.int32)):
            mism += 1
    print(f"M={M} nf={nf} {demod} agc={agc}: {steps} steps, {mism} mismatching outputs [{a.path}] ({time.time() - t0:.1f} s)", flush=True)
    bad += mism
    a.close(); b.close()
print("DETERMINISTIC" if bad == 0 else f"MISMATCHES: {bad}")
sys.exit(1 if bad else 0)
