#!/usr/bin/env python3
"""Phase times of k_shard1024 (s_memtime stamps of run 1: wave 0 = role 0, wave 4 = role 1; CSDR_SHARD1024_TRACE): per step, the
cycles of phase P's work, the wait at bar Q, phase Q's work and the wait at the next bar P, for both roles.  Needs a library whose
kernels_shard1024.hip was built with -DS1_TRACE=1: tools/build_variant.sh s1trace kernels_shard1024.hip -DS1_TRACE=1, then
CSDR_LIB=$PWD/composable_sdr_amd/variants/libcsdr_s1trace.so python tools/trace_shard1024.py [G] [fm|none]"""
import os as _os; _os.environ.setdefault("CSDR_DIAG", "1")   # tools are diagnostics: the library's A/B knobs are live here
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
path = "/tmp/s1_trace.bin"
os.environ["CSDR_SHARD1024_TRACE"] = path
import numpy as np, torch
import composable_sdr_amd as cs
from composable_sdr_amd import _lib
from synth import synth_cf32_torch
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
demod = sys.argv[2] if len(sys.argv) > 2 else "fm"
M, nf = 1024, 65536
dev = torch.device("cuda", 0)
x = synth_cf32_torch(M * nf, M, dev, seed=1)
out = torch.empty(M * nf * 2 // G, dtype=torch.float32, device=dev)
ch = cs.Chain(channels=M, demod=demod, max_frames=nf, flags=_lib.FLAG_QUIET, chan_first=0, chan_stride=G)
for i in range(30): ch.process_device(x.data_ptr(), M * nf, out.data_ptr(), 0)
torch.cuda.synchronize()
raw = np.fromfile(path, dtype=np.uint64).astype(np.int64)
R0 = raw[:384].reshape(96, 4); R1 = raw[384:].reshape(96, 4)
n = 66
print("kernel", ch.kernel_time()[0], "total cycles (role 0, steps 0..%d)" % n, R0[n, 0] - R0[0, 0])
print("step | role 0: P-work waitQ Q-work waitP | role 1: P-work waitQ Q-work waitP")
rows = []
for s in range(1, n):
    a, b = R0[s], R1[s]
    ra = (a[1] - a[0], a[2] - a[1], a[3] - a[2], R0[s + 1, 0] - a[3])
    rb = (b[1] - b[0], b[2] - b[1], b[3] - b[2], R1[s + 1, 0] - b[3])
    rows.append(ra + rb)
    if s < 12 or s % 8 == 5: print(f"{s:4d} | " + " ".join(f"{v:6d}" for v in ra) + " | " + " ".join(f"{v:6d}" for v in rb))
r = np.array(rows[10:-2])
print("median (steady steps):", np.median(r, axis=0).astype(int), " step:", int(np.median(r[:, :4].sum(axis=1))))
