#!/usr/bin/env python3
"""Generates tests/golden/chain_*.npz: seeded inputs + per-stage outputs of the CPU oracle
(oracle/csdr_oracle.c).  These are RESTATEMENT goldens (liquid-dsp is not available here);
the only liquid-derived anchors are the KATs in kat_ex1_5_gif.json.  Re-run from the repo root:
    python tests/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle_lib as O  # noqa: E402
from synth import synth_cf32  # noqa: E402

GOLD = os.path.join(HERE, "golden")


def case(M, nfs, seed):
    x = synth_cf32(M * sum(nfs), M, seed=seed)
    dc = O.DcBlock(0.0005)
    chan = O.Chan(M)
    deno = O.Chain(M)
    fm = O.Chain(M, demod="fm", kf=0.3)
    agc = O.Chain(M, agc_db=10.0, demod="fm", kf=0.3)
    mix = O.Chain(M, demod="fm", kf=0.3, mix=True)
    outs = {k: [] for k in ("dc", "pfb", "deno", "fm", "agcfm", "mixfm")}
    pos = 0
    for nf in nfs:
        c = x[pos:pos + nf * M]
        pos += nf * M
        d = dc.execute(c)
        outs["dc"].append(d)
        outs["pfb"].append(chan.process(d))
        outs["deno"].append(deno.process(c))
        outs["fm"].append(fm.process(c))
        outs["agcfm"].append(agc.process(c))
        outs["mixfm"].append(mix.process(c))
    cat = lambda v, ax: np.concatenate(v, axis=ax)
    return dict(x=x, nfs=np.array(nfs), dc=cat(outs["dc"], 0), pfb=cat(outs["pfb"], 1), deno=cat(outs["deno"], 1),
                fm=cat(outs["fm"], 1), agcfm=cat(outs["agcfm"], 1), mixfm=cat(outs["mixfm"], 0),
                taps=O.Pfb(M).taps, dtheta=np.uint32(chan.dtheta))


def blocks():
    """front-end and tail blocks outside the channelizer: resampler (two rates, two chunks), AM peak detector,
    de-emphasis biquad, audio decimator, and the WBFM / AM chains at M = 8"""
    rng = np.random.default_rng(31)
    n = 6000
    t = np.arange(n)
    x = (0.5 * np.exp(2j * np.pi * 0.013 * t) + 0.25 * np.exp(-2j * np.pi * 0.004 * t)
         + 0.03 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    out = dict(x=x, split=np.array([2500, 3500]))
    for name, r in (("rs_0078125", 200e3 / 2.56e6), ("rs_0625", 0.625)):
        q = O.MsResamp(np.float32(r))
        out[name] = np.concatenate([q.execute(x[:2500]), q.execute(x[2500:])])
    am = O.AmpDem()
    out["am"] = np.concatenate([am.demodulate_block(x[:2500]), am.demodulate_block(x[2500:])])
    xr = (np.sin(2 * np.pi * 0.01 * t) + 0.2 * rng.standard_normal(n)).astype(np.float32)
    out["xr"] = xr
    bq = O.Butter2(0.025)
    out["butter2"] = np.concatenate([bq.execute_block(xr[:2500]), bq.execute_block(xr[2500:])])
    out["butter2_ba"] = np.concatenate(bq.coeffs)
    fd = O.FirDecim(4)
    out["firdecim4"] = np.concatenate([fd.execute_block(xr[:2500 // 4 * 4]), fd.execute_block(xr[2500 // 4 * 4:])])
    out["firdecim4_taps"] = fd.taps
    M, nfs = 8, [64, 192]
    xc = synth_cf32(M * sum(nfs), M, seed=15)
    out["xc"] = xc
    out["nfs"] = np.array(nfs)
    wb, amc = O.Chain(M, demod="wbfm", decim=4, deemph_fc=0.025), O.Chain(M, demod="am")
    wbo, amo, pos = [], [], 0
    for nf in nfs:
        c = xc[pos:pos + nf * M]
        pos += nf * M
        wbo.append(wb.process(c)); amo.append(amc.process(c))
    out["chain_wbfm"] = np.concatenate(wbo, axis=1)
    out["chain_am"] = np.concatenate(amo, axis=1)
    return out


if __name__ == "__main__":
    path = os.path.join(GOLD, "blocks_v1.npz")
    np.savez_compressed(path, **blocks())
    print(path, os.path.getsize(path))
    for M, nfs, seed in ((4, [24, 8], 11), (20, [16, 16], 12), (64, [12, 4], 13), (256, [20, 12], 14)):
        path = os.path.join(GOLD, f"chain_M{M}.npz")
        np.savez_compressed(path, **case(M, nfs, seed))
        print(path, os.path.getsize(path))
