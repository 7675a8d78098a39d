#!/usr/bin/env python3
"""Per-phase timeline of the fused tile kernel (CSDR_TRACE=1): runs a few bench-sized steps and
prints the median cycles between consecutive s_memtime stamps of thread 0, over all tiles."""
import os
import sys
os.environ["CSDR_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import composable_sdr_amd as cs
from composable_sdr_amd import _lib
from synth import synth_cf32_torch

M, nf = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 262144
dev = torch.device("cuda", 0)
x = synth_cf32_torch(M * nf, M, dev)
out = torch.empty(M * nf, dtype=torch.float32, device=dev)
ch = cs.Chain(channels=M, demod="fm", max_frames=nf, flags=_lib.FLAG_QUIET)
for _ in range(3):
    ch.process_device(x.data_ptr(), M * nf, out.data_ptr(), 0)
torch.cuda.synchronize()
nb = nf // 16
buf = np.zeros((nb, 16), dtype=np.uint64)
got = _lib.lib().csdr_chain_debug_trace(ch.h, buf.ctypes.data, nb)
run = "k_run256" in ch.kernel_time()[0] or nf // 16 >= 8192
if run:
    t = buf[:got, :9].astype(np.int64)
    t = t[t[:, 8] > 0]
    names = ["stage+scan (incl. load wait)", "column read + P", "DC finish + premix", "FIR", "pass1", "pass2", "tail (FM + stores)", "end barrier"]
    d = np.diff(t, axis=1)
    print(f"run kernel: tiles traced {len(t)}; median tile time {np.median(t[:, 8] - t[:, 0]):.0f} cycles")
    for i in range(8):
        print(f"  {names[i]:32s} median {np.median(d[:, i]):8.0f}  p90 {np.quantile(d[:, i], 0.9):8.0f}")
else:
    t = buf[:got, :12].astype(np.int64)
    names = ["ticket", "own load+scan", "look-back(w0)", "barrier", "halo stage+scan", "P + DC finish", "FIR", "pass1", "pass2",
             "tail: publish last", "tail: m[1..15] + wait prev", "m[0] + stores"]
    d = np.diff(t, axis=1)
    print(f"tiles {got}; median tile lifetime {np.median(t[:, 11] - t[:, 0]):.0f}")
    for i in range(11):
        print(f"  {names[i + 1]:32s} median {np.median(d[:, i]):8.0f}  p90 {np.quantile(d[:, i], 0.9):8.0f}")
