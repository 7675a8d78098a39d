#!/usr/bin/env python3
"""Throughput of the host-buffer entry points (PCIe-inclusive; never the bench's `value`):
  * csdr_chain_process        blocking, pageable caller buffers (what the reference's unsafe FFI call looks like)
  * csdr_chain_submit/collect up to CSDR_CHAIN_INFLIGHT chunks in flight, page-locked caller buffers (csdr_host_alloc)
at the reference's chunk (4 * 256 * 1024 samples = 4096 frames) and at 65 536 frames.  Usage: python tools/host_path_rate.py"""
import os as _os; _os.environ.setdefault("CSDR_DIAG", "1")   # tools are diagnostics: the library's A/B knobs are live here
import os
import sys
import time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import composable_sdr_amd as cs
from composable_sdr_amd.pipes import host_array
from synth import synth_cf32

M, nf0 = 256, 4096
x0 = synth_cf32(M * nf0, M)
for demod, ob in (("fm", 4), ("none", 8)):
    for nfc in (4096, 65536):
        xx = np.tile(x0, nfc // nf0)
        ch = cs.Chain(channels=M, demod=demod, max_frames=nfc)
        ref = ch.process(xx)
        n = max(4, (1 << 22) // nfc)
        t = time.perf_counter()
        for _ in range(n):
            ch.process(xx)
        dt = time.perf_counter() - t
        line = f"{demod:4s} {nfc:6d} frames/chunk: blocking pageable {n * xx.size / dt / 1e6:7.0f} MS/s ({dt / n * 1e3:6.2f} ms/call)"
        ch.close()
        # asynchronous, page-locked
        ch = cs.Chain(channels=M, demod=demod, max_frames=nfc)
        depth = 3
        ins = [host_array((xx.size,), np.complex64) for _ in range(depth)]
        outs = [host_array(ref.shape, ref.dtype) for _ in range(depth)]
        for b in ins:
            b.a[:] = xx
        ch.submit(ins[0].a, outs[0].a); got = ch.collect().copy()          # same stream position as `ref`
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "async path differs from the blocking one"
        t = time.perf_counter()
        q = 0
        for i in range(n):
            if q == depth:
                ch.collect(); q -= 1
            ch.submit(ins[i % depth].a, outs[i % depth].a); q += 1
        while q:
            ch.collect(); q -= 1
        dt = time.perf_counter() - t
        gbs = n * xx.size * (8 + ob) / dt / 1e9
        print(line + f" | submit/collect page-locked {n * xx.size / dt / 1e6:7.0f} MS/s ({dt / n * 1e3:6.2f} ms/chunk, {gbs:5.1f} GB/s over PCIe both ways; H2D alone {n * xx.size * 8 / dt / 1e9:5.1f} GB/s of 63)")
        ch.close()
