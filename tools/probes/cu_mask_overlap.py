#!/usr/bin/env python3
"""Can the channelizer (power / VALU-bound) and the AGC tail (memory-bound) of the cfg3 + AGC step run side by side on disjoint sets of
compute units?  Two handles -- a DeNo chain (k_run256v2<CF32>) and a tail-only chain (k_agc_spec*) -- on two streams created with
hipExtStreamCreateWithCUMask; each alone on the whole device, each alone behind its mask, both at once."""
import os as _os; _os.environ.setdefault("CSDR_DIAG", "1")   # tools are diagnostics: the library's A/B knobs are live here
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import composable_sdr_amd as cs
from composable_sdr_amd import _lib
from synth import synth_cf32_torch

hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (len(bits) + 31) // 32
    arr = (ctypes.c_uint32 * words)()
    for i, b in enumerate(bits):
        if b: arr[i // 32] |= 1 << (i % 32)
    st = ctypes.c_void_p()
    r = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), arr)
    assert r == 0, f"hipExtStreamCreateWithCUMask -> {r}"
    return st.value

M, nf, K = 256, 262144, int(os.environ.get("STEPS", "40"))
dev = torch.device("cuda", 0)
x = synth_cf32_torch(M * nf, M, dev, seed=5)
plane = torch.empty(M * nf * 2, dtype=torch.float32, device=dev)
fm = torch.empty(M * nf, dtype=torch.float32, device=dev)
plane2 = torch.empty(M * nf * 2, dtype=torch.float32, device=dev)

def make(cus):
    if cus: os.environ["CSDR_CUS"] = str(cus)
    else: os.environ.pop("CSDR_CUS", None)
def chains(ca, cb):
    make(ca); a = cs.Chain(channels=M, max_frames=nf, flags=_lib.FLAG_QUIET)
    make(cb); b = cs.Chain(channels=M, demod="fm", kf=0.3, agc=10.0, tail_only=True, max_frames=nf, flags=_lib.FLAG_QUIET)
    make(0)
    return a, b

def run(a, b, sa, sb, both=True, only=None):
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

# a plausible plane for the tail: the channelizer's own output
a, b = chains(0, 0)
a.process_device(x.data_ptr(), M * nf, plane2.data_ptr(), 0); torch.cuda.synchronize()
print(f"whole device: channelizer alone {run(a, b, 0, 0, only='a'):.1f} us, tail alone {run(a, b, 0, 0, only='b'):.1f} us, one after the other {run(a, b, 0, 0):.1f} us", flush=True)
a.close(); b.close()
NCU = 256
for na in (int(v) for v in os.environ.get("SPLITS", "128,160,192").split(",")):
    for layout in ("block", "xcd"):
        if layout == "block": bits_a = [i < na for i in range(NCU)]
        else: bits_a = [(i // 8) < na // 8 for i in range(NCU)] if os.environ.get("XCD_MINOR") else [(i % 32) < na // 8 for i in range(NCU)]
        bits_b = [not v for v in bits_a]
        sa, sb = masked_stream(bits_a), masked_stream(bits_b)
        a, b = chains(na, NCU - na)
        ta, tb = run(a, b, sa, sb, only="a"), run(a, b, sa, sb, only="b")
        tab = run(a, b, sa, sb)
        print(f"channelizer on {na} CUs ({layout}) {ta:.1f} us | tail on {NCU - na} CUs {tb:.1f} us | both at once {tab:.1f} us per pair", flush=True)
        a.close(); b.close()
