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


BLOCKS = os.path.join(os.path.dirname(__file__), "golden", "blocks_v1.npz")


def test_oracle_reproduces_block_goldens_bit_exact():
    """resampler / AM / de-emphasis / decimator / WBFM + AM chains: restatement goldens (unpinned blocks)"""
    g = np.load(BLOCKS)
    x, a = g["x"], int(g["split"][0])
    for name, r in (("rs_0078125", 200e3 / 2.56e6), ("rs_0625", 0.625)):
        q = O.MsResamp(np.float32(r))
        assert np.array_equal(np.concatenate([q.execute(x[:a]), q.execute(x[a:])]), g[name])
    am = O.AmpDem()
    assert np.array_equal(np.concatenate([am.demodulate_block(x[:a]), am.demodulate_block(x[a:])]), g["am"])
    xr = g["xr"]
    bq = O.Butter2(0.025)
    assert np.array_equal(np.concatenate(bq.coeffs), g["butter2_ba"])
    assert np.array_equal(np.concatenate([bq.execute_block(xr[:a]), bq.execute_block(xr[a:])]), g["butter2"])
    fd = O.FirDecim(4)
    assert np.array_equal(fd.taps, g["firdecim4_taps"])
    assert np.array_equal(np.concatenate([fd.execute_block(xr[:a // 4 * 4]), fd.execute_block(xr[a // 4 * 4:])]), g["firdecim4"])
    M, nfs, xc = 8, list(g["nfs"]), g["xc"]
    wb, amc = O.Chain(M, demod="wbfm", decim=4, deemph_fc=0.025), O.Chain(M, demod="am")
    pos, t = 0, 0
    for nf in nfs:
        c = xc[pos:pos + nf * M]
        assert np.array_equal(wb.process(c), g["chain_wbfm"][:, t // 4:(t + nf) // 4])
        assert np.array_equal(amc.process(c), g["chain_am"][:, t:t + nf])
        pos += nf * M
        t += nf


@pytest.mark.gpu
def test_hip_matches_block_goldens(monkeypatch):
    import composable_sdr_amd as cs
    monkeypatch.setenv("CSDR_QUIET", "1")
    g = np.load(BLOCKS)
    x, a = g["x"], int(g["split"][0])

    def run(pipe, chunks):
        r = pipe._start()
        try:
            return np.concatenate([pipe._process(r, c) for c in chunks])
        finally:
            pipe._done(r)
    for name, r in (("rs_0078125", 200e3 / 2.56e6), ("rs_0625", 0.625)):
        y = run(cs.resampler(float(np.float32(r)), 60.0, max_samples=4096), [x[:a], x[a:]])
        assert y.size == g[name].size and max_abs_err(y, g[name]) < 2e-6
    assert max_abs_err(run(cs.amDemodulator(max_samples=4096), [x[:a], x[a:]]), g["am"]) < 2e-6
    xr = g["xr"]
    assert max_abs_err(run(cs.iirFilter(2, 0.025, max_samples=4096), [xr[:a], xr[a:]]), g["butter2"]) < 5e-6
    assert max_abs_err(run(cs.firDecimator(4, max_samples=4096), [xr[:a // 4 * 4], xr[a // 4 * 4:]]), g["firdecim4"]) < 5e-6
    M, nfs, xc = 8, list(g["nfs"]), g["xc"]
    wb = cs.Chain(channels=M, demod="wbfm", decim=4, deemph_fc=0.025, max_frames=max(nfs))
    amc = cs.Chain(channels=M, demod="am", max_frames=max(nfs))
    pos, t = 0, 0
    for nf in nfs:
        c = xc[pos:pos + nf * M]
        dw = np.abs(wb.process(c).astype(np.float64) - g["chain_wbfm"][:, t // 4:(t + nf) // 4])
        assert np.median(dw) < 2e-5 * np.abs(g["chain_wbfm"]).max()
        da = max_abs_err(amc.process(c), g["chain_am"][:, t:t + nf])
        assert da < 2e-4 * max(1.0, np.abs(g["chain_am"]).max())
        pos += nf * M
        t += nf
