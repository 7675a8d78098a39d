#!/usr/bin/env python3
"""Per-phase timeline of the fused tile kernel (CSDR_TRACE=1): runs a few bench-sized steps and
prints the median cycles between consecutive s_memtime stamps of thread 0, over all tiles.
Since round 5 the stamps are compiled out of the product kernel (they cost k_run256v2 the SGPRs its no-warm-up start needs): build the
variant first and point CSDR_LIB at it --
    tools/build_variant.sh trace kernels_fused_v2.hip -DV2_TRACE=1
    CSDR_LIB=$PWD/composable_sdr_amd/variants/libcsdr_trace.so python tools/trace_tiles.py"""
import os as _os; _os.environ.setdefault("CSDR_DIAG", "1")   # tools are diagnostics: the library's A/B knobs are live here
import os
import sys
os.environ.setdefault("CSDR_TRACE", "1")
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
# bench.py alternates two input buffers (together beyond the 256 MiB Infinity Cache); TRACE_ALT=0 re-reads one buffer (MALL-assisted)
xs = [x, synth_cf32_torch(M * nf, M, dev, seed=20260101 + 7919)] if os.environ.get("TRACE_ALT", "1") != "0" else [x, x]
out = torch.empty(M * nf * 2, dtype=torch.float32, device=dev)
ch = cs.Chain(channels=M, demod=os.environ.get("TRACE_DEMOD", "fm"), max_frames=nf, flags=_lib.FLAG_QUIET | _lib.FLAG_TIME_KERNELS)
for i in range(3):
    ch.process_device(xs[i & 1].data_ptr(), M * nf, out.data_ptr(), 0)
torch.cuda.synchronize()
ch.kernel_time()
for i in range(6):
    ch.process_device(xs[(i + 1) & 1].data_ptr(), M * nf, out.data_ptr(), 0)
torch.cuda.synchronize()
nb = nf // 16
buf = np.zeros((nb, 16), dtype=np.uint64)
got = _lib.lib().csdr_chain_debug_trace(ch.h, buf.ctypes.data, nb)
kn, kms, kl = ch.kernel_time()
run = "k_run256" in kn or nf // 16 >= 8192
if "v2" in kn and os.environ["CSDR_TRACE"] == "2":
    L = buf[:got, 0:5].astype(np.int64)
    ent = L[(L[:, 0] > 0) & (L[:, 4] > 0)]
    e0 = ent[:, 0].min()
    q = [0, .1, .5, .9, 1]
    print(f"{kn}: light trace of {len(ent)} runs; avg launch {kms / max(kl, 1) * 1e3:.1f} us (us after the first entry, quantiles 0/10/50/90/100)")
    print("  run entry        :", (np.quantile(ent[:, 0] - e0, q) / 100).round(1))
    wu = ent[ent[:, 1] > 0]
    print("  warm-up done     :", (np.quantile(wu[:, 1] - e0, q) / 100).round(1), " duration", (np.quantile(wu[:, 1] - wu[:, 0], q) / 100).round(1))
    print("  halo + constants :", (np.quantile(ent[:, 2] - e0, q) / 100).round(1), " duration", (np.quantile(wu[:, 2] - wu[:, 1], q) / 100).round(1))
    print("  first tile landed:", (np.quantile(ent[:, 3] - e0, q) / 100).round(1), " duration", (np.quantile(ent[:, 3] - ent[:, 2], q) / 100).round(1))
    print("  run end          :", (np.quantile(ent[:, 4] - e0, q) / 100).round(1), " tile loop duration", (np.quantile(ent[:, 4] - ent[:, 3], q) / 100).round(1))
    idx = np.flatnonzero((buf[:got, 0] > 0) & (buf[:got, 4] > 0))       # row = first tile of run w, ascending in w
    wv = np.arange(len(idx))
    loop = (buf[idx, 4].astype(np.int64) - buf[idx, 3].astype(np.int64)) / 100.0
    endw = (buf[idx, 4].astype(np.int64) - e0) / 100.0
    pro = (buf[idx, 3].astype(np.int64) - e0) / 100.0
    nslot = max(1, len(idx) // 256)
    print("  by CU slot (w // 256): loop", [round(float(loop[wv // 256 == k].mean()), 1) for k in range(nslot)],
          " prologue end", [round(float(pro[wv // 256 == k].mean()), 1) for k in range(nslot)],
          " run end", [round(float(endw[wv // 256 == k].mean()), 1) for k in range(nslot)])
    print("  by XCD (w % 8): loop", [round(float(loop[wv % 8 == k].mean()), 1) for k in range(8)], " run end max", [round(float(endw[wv % 8 == k].max()), 1) for k in range(8)])
    cu = wv % 256
    pair = np.array([endw[cu == c].max() - endw[cu == c].min() for c in range(256)]) if len(idx) >= 512 else np.zeros(1)
    print("  spread of run end inside a CU (max - min over its workgroups): median", round(float(np.median(pair)), 1), "max", round(float(pair.max()), 1),
          "; CU-level last end: quantiles", np.quantile(np.array([endw[cu == c].max() for c in range(256)]), q).round(1))
    sys.exit(0)
if "v2" in kn:
    t = buf[:got, :15].astype(np.int64)
    t = t[(t[:, 14] > 0) & (t[:, 0] > 0)]
    names = ["B_a wait (tile image landed)", "DMA issue + run totals + row scan + blocker (y' in place)", "B_c wait", "column read + frame chain + pre-mix",
             "FIR + X in place", "B_d wait", "pass 1 (read X, radix-16, twiddle)", "B_e wait", "Z write", "B_f wait", "pass 2 (read Z, radix-16)",
             "freqdem + stash", "vmcnt(0) (next tile's DMA)", "stores"]
    d = np.diff(t, axis=1)
    print(f"{kn}: tiles traced {len(t)}; avg launch {kms / max(kl, 1) * 1e3:.1f} us; median tile time {np.median(t[:, 14] - t[:, 0]):.0f} cycles (wave 0 of each workgroup)")
    for i in range(14):
        print(f"  {names[i]:60s} median {np.median(d[:, i]):8.0f}  mean {np.mean(d[:, i]):8.0f}  p90 {np.quantile(d[:, i], 0.9):8.0f}")
    # shader clock: consecutive tiles of one run are consecutive rows; s_memtime ticks per 10 ns of s_memrealtime
    full = buf[:got].astype(np.int64)
    ok = (full[1:, 0] > 0) & (full[:-1, 0] > 0) & (full[1:, 15] > full[:-1, 15]) & (full[1:, 15] - full[:-1, 15] < 5000)
    clk = (full[1:, 0] - full[:-1, 0])[ok] / (full[1:, 15] - full[:-1, 15])[ok] * 0.1
    # per XCD: run w = row // tiles-per-run (uniform split), XCD = w % 8
    nruns = int(os.environ.get("CSDR_RESIDENT_WGS", "512"))
    tpr = max(1, got // nruns)
    xcd = (np.arange(got - 1) // tpr) % 8
    per = (full[1:, 0] - full[:-1, 0]); rt = (full[1:, 15] - full[:-1, 15])
    vm = full[:, 13] - full[:, 12]
    print("  by XCD: clock GHz", [round(float(np.median(clk[xcd[ok] == k])), 3) for k in range(8)],
          " tile period us", [round(float(np.mean(rt[ok & (xcd == k)])) / 100, 2) for k in range(8)],
          " mean DMA wait cyc", [int(np.mean(vm[:-1][ok & (xcd == k)])) for k in range(8)])
    print(f"  mean tile period {np.mean((full[1:, 0] - full[:-1, 0])[ok]):.0f} cycles = {np.mean((full[1:, 15] - full[:-1, 15])[ok]) / 100:.2f} us; shader clock median {np.median(clk):.3f} GHz (p10 {np.quantile(clk, .1):.3f}, p90 {np.quantile(clk, .9):.3f})")
    sys.exit(0)
if run and os.environ["CSDR_TRACE"] == "2":
    L = buf[:got, 11:15].astype(np.int64)
    ent = L[L[:, 0] > 0]; e0 = ent[:, 0].min()
    endt = L[L[:, 3] > 0][:, 3]
    print(f"light trace: {len(ent)} runs; kernel {kn} avg {kms / max(kl, 1) * 1e3:.1f} us (hipEvent)")
    q = [0, .1, .5, .9, 1]
    print("  run entry      (us after first entry):", (np.quantile(ent[:, 0] - e0, q) / 100).round(1))
    wu = ent[ent[:, 1] > 0]
    print("  warm-up done   :", (np.quantile(wu[:, 1] - e0, q) / 100).round(1), " duration", (np.quantile(wu[:, 1] - wu[:, 0], q) / 100).round(1))
    print("  prologue done  :", (np.quantile(ent[:, 2] - e0, q) / 100).round(1), " halo duration", (np.quantile(wu[:, 2] - wu[:, 1], q) / 100).round(1))
    print("  run end        :", (np.quantile(endt - e0, q) / 100).round(1))
    idx = np.flatnonzero(buf[:got, 11] > 0)                      # row = first tile of run w, ascending in w
    endw = (buf[idx, 14].astype(np.int64) - e0) / 100.0
    prow = (buf[idx, 13].astype(np.int64) - e0) / 100.0
    wv = np.arange(len(idx))
    print("  mean run end by w%8 (XCD):", [round(float(endw[wv % 8 == k].mean()), 1) for k in range(8)])
    print("  mean prologue end by w%8 :", [round(float(prow[wv % 8 == k].mean()), 1) for k in range(8)])
    print("  mean run end by w//256   :", [round(float(endw[wv // 256 == k].mean()), 1) for k in range(3)])
    print("  mean run end by (w//8)%32 (CU slot in XCD):", [round(float(endw[(wv // 8) % 32 == k].mean()), 0) for k in range(32)])
    ntl = np.diff(np.append(idx, got))
    print("  tiles per run min/max:", ntl.min(), ntl.max(), " corr(end, ntiles) =", round(float(np.corrcoef(endw, ntl)[0, 1]), 3))
    sys.exit(0)
if run:
    rt = buf[:got, 9:11].astype(np.int64)
    ok = (rt[:, 0] > 0) & (rt[:, 1] > 0)
    r0 = rt[ok, 0].min()
    print(f"s_memrealtime (100 MHz): launch span {(rt[ok, 1].max() - r0) / 100:.1f} us")
    ntile = ok.sum()
    starts = np.sort(rt[ok, 0] - r0) / 100.0; ends = np.sort(rt[ok, 1] - r0) / 100.0
    print("  tile start times us (quantiles 0,25,50,75,100):", np.quantile(starts, [0, .25, .5, .75, 1]).round(1))
    print("  tile end   times us (quantiles 0,25,50,75,100):", np.quantile(ends, [0, .25, .5, .75, 1]).round(1))
    dm = (buf[:got, 8].astype(np.int64) - buf[:got, 0].astype(np.int64))[ok]; dr = (rt[ok, 1] - rt[ok, 0])
    print(f"  shader clock from per-tile memtime/realtime: median {np.median(dm / np.maximum(dr, 1)) * 0.1:.3f} GHz")
    t = buf[:got, :9].astype(np.int64)
    t = t[(t[:, 8] > 0) & (t[:, 0] > 0)]
    names = ["stage+scan (incl. load wait)", "column read + P", "DC finish + premix", "FIR", "pass1", "pass2", "tail (FM + stores)", "end barrier"]
    d = np.diff(t, axis=1)
    print(f"run kernel: tiles traced {len(t)}; median tile time {np.median(t[:, 8] - t[:, 0]):.0f} cycles")
    o = np.argsort(t[:, 0]); ts = t[o]
    cuts = np.flatnonzero(np.diff(ts[:, 0]) > 5_000_000)          # per-XCD counters are not aligned
    groups = np.split(ts, cuts + 1)
    spans = [int(g[:, 8].max() - g[:, 0].min()) for g in groups]
    print("clusters:", [(len(g), sp) for g, sp in zip(groups, spans)])
    span = int(np.median(spans))
    print(f"launch span {span} ticks; kernel {kn} avg {kms / max(kl, 1):.4f} ms over {kl} launches -> {span / (kms / max(kl, 1)) / 1e6:.3f} GHz tick rate (if span ~ launch)")
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
