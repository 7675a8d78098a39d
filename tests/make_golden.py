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


if __name__ == "__main__":
    for M, nfs, seed in ((4, [24, 8], 11), (20, [16, 16], 12), (64, [12, 4], 13), (256, [20, 12], 14)):
        path = os.path.join(GOLD, f"chain_M{M}.npz")
        np.savez_compressed(path, **case(M, nfs, seed))
        print(path, os.path.getsize(path))
