#!/usr/bin/env python3
"""Phase times of k_run1024v3 (s_memtime stamps of run 1: front wave 0, back wave 4; CSDR_RUN1024_V3_TRACE): per step, the cycles
of phase P's work, the wait at bar Q, phase Q's work and what lies between it and bar P, for both roles.  Needs a library whose
kernels_run1024_v3.hip was built with -DB3_TRACE=1: tools/build_variant.sh b3trace kernels_run1024_v3.hip -DB3_TRACE=1, then
CSDR_LIB=$PWD/composable_sdr_amd/variants/libcsdr_b3trace.so python tools/trace_run1024v3.py"""
import os as _os; _os.environ.setdefault("CSDR_DIAG", "1")   # tools are diagnostics: the library's A/B knobs are live here
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
path = "/tmp/v3_trace.bin"
os.environ["CSDR_RUN1024_V3_TRACE"] = path
import numpy as np, torch
import composable_sdr_amd as cs
from composable_sdr_amd import _lib
from synth import synth_cf32_torch
M, nf = 1024, 65536
dev = torch.device("cuda", 0)
x = synth_cf32_torch(M * nf, M, dev, seed=1)
out = torch.empty(M * nf, dtype=torch.float32, device=dev)
ch = cs.Chain(channels=M, demod="fm", max_frames=nf, flags=_lib.FLAG_QUIET)
for i in range(30): ch.process_device(x.data_ptr(), M * nf, out.data_ptr(), 0)
torch.cuda.synchronize()
raw = np.fromfile(path, dtype=np.uint64).astype(np.int64)
F = raw[:1024].reshape(128, 8); Bk = raw[1024:].reshape(128, 4)
n = 70
print("total cycles (front, steps 0..%d)" % n, F[n, 0] - F[0, 0])
print("step | front: P-work waitQ  Q-work  tail  vmwait issue waitP | back: P-work waitQ Q-work waitP")
rows = []
for s in range(1, n):
    f, b = F[s], Bk[s]
    fr = (f[1] - f[0], f[2] - f[1], f[3] - f[2], F[s + 1, 4] - f[3], F[s + 1, 5] - F[s + 1, 4], F[s + 1, 6] - F[s + 1, 5], F[s + 1, 0] - F[s + 1, 6])
    bk = (b[1] - b[0], b[2] - b[1], b[3] - b[2], Bk[s + 1, 0] - b[3])
    rows.append(fr + bk)
    if s < 20 or s % 8 == 5: print(f"{s:4d} | " + " ".join(f"{v:6d}" for v in fr) + " | " + " ".join(f"{v:6d}" for v in bk))
r = np.array(rows[14:-2])
print("median (steady steps):", np.median(r, axis=0).astype(int), " step:", int(np.median(r[:, :7].sum(axis=1))))
