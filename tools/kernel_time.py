#!/usr/bin/env python3
"""Launch time of the dominant kernel of a chain at the bench size, for A/B builds (CSDR_LIB) and env knobs:
    python tools/kernel_time.py [demod=fm|none] [channels] [frames] [agc]"""
import os as _os; _os.environ.setdefault("CSDR_DIAG", "1")   # tools are diagnostics: the library's A/B knobs are live here
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import composable_sdr_amd as cs
from composable_sdr_amd import _lib
from synth import synth_cf32_torch

demod = sys.argv[1] if len(sys.argv) > 1 else "fm"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 256
nf = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
agc = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
mix = len(sys.argv) > 5 and sys.argv[5] == "mix"
dev = torch.device("cuda", 0)
xs = [synth_cf32_torch(M * nf, M, dev, seed=5 + i) for i in range(2)]
if os.environ.get("ZERO_INPUT"):            # DVFS check: same instruction stream, no data toggling
    xs = [torch.zeros_like(x) for x in xs]
out = torch.empty(M * nf * 2, dtype=torch.float32, device=dev)
ch = cs.Chain(channels=M, demod=demod, agc=agc, mix=mix, max_frames=nf, flags=_lib.FLAG_QUIET | _lib.FLAG_TIME_KERNELS)
for i in range(3):
    ch.process_device(xs[i & 1].data_ptr(), M * nf, out.data_ptr(), 0)
torch.cuda.synchronize()
ch.kernel_time()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 20
for i in range(n):
    ch.process_device(xs[i & 1].data_ptr(), M * nf, out.data_ptr(), 0)
e1.record()
torch.cuda.synchronize()
kn, kms, kl = ch.kernel_time()
print(f"{os.environ.get('CSDR_LIB', 'default'):40s} {demod:5s} M={M} nf={nf} agc={agc}: {kn} {kms / max(kl, 1) * 1e3:8.1f} us/launch, step {e0.elapsed_time(e1) / n * 1e3:8.1f} us")
