#!/usr/bin/env python3
"""Channelizer and AGC tail of different chunks on two plain streams (no CU masks): with ONE channelizer workgroup per CU
(CSDR_RESIDENT_WGS) half of every CU's registers and LDS stay free for tail workgroups -- do the two kernels share the CUs?"""
import os as _os; _os.environ.setdefault("CSDR_DIAG", "1")   # tools are diagnostics: the library's A/B knobs are live here
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import composable_sdr_amd as cs
from composable_sdr_amd import _lib
from synth import synth_cf32_torch
M, nf, K = 256, 262144, int(os.environ.get("STEPS", "40"))
dev = torch.device("cuda", 0)
x = synth_cf32_torch(M * nf, M, dev, seed=5)
plane = torch.empty(M * nf * 2, dtype=torch.float32, device=dev)
plane2 = torch.empty(M * nf * 2, dtype=torch.float32, device=dev)
fm = torch.empty(M * nf, dtype=torch.float32, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(a, b, sa, sb, only=None):
    for _ in range(3):
        if only != "b": a.process_device(x.data_ptr(), M * nf, plane.data_ptr(), sa)
        if only != "a": b.process_device(plane2.data_ptr(), M * nf, fm.data_ptr(), sb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        if only != "b": a.process_device(x.data_ptr(), M * nf, plane.data_ptr(), sa)
        if only != "a": b.process_device(plane2.data_ptr(), M * nf, fm.data_ptr(), sb)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e6
for wgs in (0, 256, 384):
    for tail_wgs in (0, 2, 1):
        if wgs: os.environ["CSDR_RESIDENT_WGS"] = str(wgs)
        else: os.environ.pop("CSDR_RESIDENT_WGS", None)
        a = cs.Chain(channels=M, max_frames=nf, flags=_lib.FLAG_QUIET)
        os.environ.pop("CSDR_RESIDENT_WGS", None)
        if tail_wgs: os.environ["CSDR_AGC_WGS"] = str(tail_wgs)
        else: os.environ.pop("CSDR_AGC_WGS", None)
        b = cs.Chain(channels=M, demod="fm", kf=0.3, agc=10.0, tail_only=True, max_frames=nf, flags=_lib.FLAG_QUIET)
        os.environ.pop("CSDR_AGC_WGS", None)
        a.process_device(x.data_ptr(), M * nf, plane2.data_ptr(), 0); torch.cuda.synchronize()
        ta, tb = run(a, b, 0, 0, only="a"), run(a, b, 0, 0, only="b")
        ser = run(a, b, 0, 0)
        par = run(a, b, s1.cuda_stream, s2.cuda_stream)
        print(f"channelizer runs {wgs or 512}, tail WG slots/CU {tail_wgs or 4}: alone {ta:.1f} + {tb:.1f} us; one stream {ser:.1f} us; two streams {par:.1f} us per pair", flush=True)
        a.close(); b.close()
