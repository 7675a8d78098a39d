#!/usr/bin/env python3
"""cfg5 shape on one GPU (4096-ch PFB, DeNo --mix): step time of the product path (M * branch-0 FIR behind the DC blocker,
`generic+mix-identity`) and of the full bank + DFT + channel sum (CSDR_FLAG_NO_MIX_IDENTITY).  Usage: python tools/cfg5_time.py"""
import os as _os; _os.environ.setdefault("CSDR_DIAG", "1")   # tools are diagnostics: the library's A/B knobs are live here
import os
import sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import composable_sdr_amd as cs
from composable_sdr_amd import _lib
from synth import synth_cf32_torch
M, nf = 4096, 16384
dev = torch.device("cuda", 0)
xs = [synth_cf32_torch(M * nf, M, dev, seed=5 + i) for i in range(2)]
out = torch.empty(nf * 2, dtype=torch.float32, device=dev)
for extra in (0, _lib.FLAG_NO_MIX_IDENTITY):
    ch = cs.Chain(channels=M, demod="none", mix=True, max_frames=nf, flags=_lib.FLAG_QUIET | extra)
    for i in range(3):
        ch.process_device(xs[i & 1].data_ptr(), M * nf, out.data_ptr(), 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20):
        ch.process_device(xs[i & 1].data_ptr(), M * nf, out.data_ptr(), 0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"cfg5 shape [{ch.path}]: {ms * 1e3:.1f} us/step = {M * nf / ms / 1e6:.1f} GS/s, {M * nf * 8.0 / ms / 1e9 * 1e3 / 8000:.3f} of the 8 TB/s roofline (8 B/sample)")
    ch.close()
