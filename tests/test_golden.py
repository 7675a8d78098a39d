"""Golden fixtures (tests/golden/chain_M*.npz, written by tests/make_golden.py from the CPU
oracle): the oracle must reproduce them bit for bit (CPU), the HIP path within tolerance (GPU)."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as O
from util import max_abs_err, rel_rms, wrap_pm

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "chain_M*.npz")))


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_oracle_reproduces_golden_bit_exact(path):
    g = np.load(path)
    M = int(os.path.basename(path)[7:-4])
    x, nfs = g["x"], list(g["nfs"])
    assert np.array_equal(O.Pfb(M).taps, g["taps"])
    assert O.Chan(M).dtheta == int(g["dtheta"])
    fm, agc, mix = O.Chain(M, demod="fm", kf=0.3), O.Chain(M, agc_db=10.0, demod="fm", kf=0.3), O.Chain(M, demod="fm", kf=0.3, mix=True)
    deno = O.Chain(M)
    pos, t = 0, 0
    for nf in nfs:
        c = x[pos:pos + nf * M]
        pos += nf * M
        assert np.array_equal(deno.process(c), g["deno"][:, t:t + nf])
        assert np.array_equal(fm.process(c), g["fm"][:, t:t + nf])
        assert np.array_equal(agc.process(c), g["agcfm"][:, t:t + nf])
        assert np.array_equal(mix.process(c), g["mixfm"][t:t + nf])
        t += nf


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_hip_matches_golden(path):
    import composable_sdr_amd as cs
    g = np.load(path)
    M = int(os.path.basename(path)[7:-4])
    x, nfs = g["x"], list(g["nfs"])
    deno, fm = cs.Chain(channels=M, max_frames=max(nfs)), cs.Chain(channels=M, demod="fm", kf=0.3, max_frames=max(nfs))
    pfb = cs.Chain(channels=M, dc_block=False, max_frames=max(nfs))
    assert np.max(np.abs(deno.taps - g["taps"])) < 1e-9 and deno.nco[1] == int(g["dtheta"])
    pos, t = 0, 0
    for nf in nfs:
        c = x[pos:pos + nf * M]
        a = deno.process(c)
        w = g["deno"][:, t:t + nf]
        assert rel_rms(a, w) < 1e-5 and max_abs_err(a, w) < 1e-4 * np.abs(w).max()
        # the channelizer alone, fed with the golden DC-blocked stream: f32 round-off only
        b = pfb.process(g["dc"][pos:pos + nf * M])
        assert rel_rms(b, g["pfb"][:, t:t + nf]) < 1e-6
        f = fm.process(c)
        d = np.abs(wrap_pm(f.astype(np.float64) - g["fm"][:, t:t + nf], 1.0 / 0.3))
        assert np.median(d) < 2e-5
        pos += nf * M
        t += nf
