#!/usr/bin/env python3
"""Whole-step wall time and the dominant kernel's launch time of the bench configuration under the three timer modes
(no events / one event pair per launch / one pair per region), and optionally the board power and clocks while the
chain runs back to back (POWER=1: samples rocm-smi from a side thread).  Diagnostics for DESIGN section 6."""
import os as _os; _os.environ.setdefault("CSDR_DIAG", "1")   # tools are diagnostics: the library's A/B knobs are live here
import os, sys, time, json, subprocess, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import composable_sdr_amd as cs
from composable_sdr_amd import _lib
from synth import synth_cf32_torch

M = int(os.environ.get("STEP_M", "256")); nf = int(os.environ.get("STEP_NF", str(262144 * 256 // M)))
demod = os.environ.get("STEP_DEMOD", "fm"); steps = int(os.environ.get("STEP_STEPS", "40")); agc = float(os.environ.get("STEP_AGC", "0"))
dev = torch.device("cuda", 0)
xs = [synth_cf32_torch(M * nf, M, dev, seed=20260101 + 7919 * i) for i in range(2)]
out = torch.empty(M * nf * 2, dtype=torch.float32, device=dev)
for name, fl in (("no timer", 0), ("event pair per launch", _lib.FLAG_TIME_KERNELS), ("region", _lib.FLAG_TIME_KERNELS | _lib.FLAG_TIME_REGION)):
    ch = cs.Chain(channels=M, demod=demod, agc=agc, max_frames=nf, flags=_lib.FLAG_QUIET | fl)
    for i in range(3):
        ch.process_device(xs[i & 1].data_ptr(), M * nf, out.data_ptr(), 0)
    torch.cuda.synchronize()
    if fl: ch.kernel_time()
    t0 = time.perf_counter()
    for i in range(steps):
        ch.process_device(xs[(i + 1) & 1].data_ptr(), M * nf, out.data_ptr(), 0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    k = ch.kernel_time() if fl else ("-", 0.0, 0)
    print(f"{name:24s}: {dt * 1e6:8.1f} us per step ({M * nf / dt / 1e9:.1f} GS/s); kernel {k[0]} {k[1] / max(k[2], 1) * 1e3:.1f} us over {k[2]} launches", flush=True)
    ch.close()

if os.environ.get("POWER") == "1":
    ch = cs.Chain(channels=M, demod=demod, agc=agc, max_frames=nf, flags=_lib.FLAG_QUIET)
    stop = False
    def sample():
        while not stop:
            try:
                o = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True, text=True, timeout=20).stdout
                d = json.loads(o)
                for card, v in d.items():
                    keep = {k: v[k] for k in v if any(s in k.lower() for s in ("power", "sclk", "mclk", "fclk", "junction", "hotspot", "edge"))}
                    print("  smi", card, keep, flush=True)
            except Exception as e:
                print("  smi failed:", e, flush=True)
            time.sleep(0.7)
    th = threading.Thread(target=sample); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < float(os.environ.get("POWER_SECONDS", "8")):
        for i in range(200):
            ch.process_device(xs[i & 1].data_ptr(), M * nf, out.data_ptr(), 0)
        torch.cuda.synchronize(); n += 200
    dt = time.perf_counter() - t0
    stop = True; th.join()
    print(f"sustained: {n} steps in {dt:.1f} s = {dt / n * 1e6:.1f} us per step")
