"""Pins the CPU oracle (oracle/csdr_oracle.c) to the only liquid-dsp-derived
numbers the reference owns: values printed in /root/reference/images/ex1_5.gif
 and the README Example 3 sizes (KAT4)."""
import json
import os

import numpy as np

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _kat():
    with open(os.path.join(GOLD, "kat_ex1_5_gif.json")) as f:
        return json.load(f)


def test_kat1_prototype_taps_m20():
    kat = _kat()["kat1_taps_M20_m7_As80"]
    pfb = O.Pfb(20, 7, 80.0)
    h = pfb.taps
    assert h.size == 280                      # 2*M*m of the 281 designed taps are used
    for idx, val in kat.items():
        assert abs(float(h[int(idx)]) - val) < 2e-8, (idx, h[int(idx)], val)
    assert abs(h[260]) < 1e-9                 # centre at 140, fc = 0.5/M


def test_kat2_nco_frequency_word():
    kat = _kat()["kat2_nco"]
    off = O.pfb_offset(20)
    assert abs(off - kat["offset_printed"]) < 1e-6
    assert O.nco_constrain(off) == int(kat["freq_word_hex"], 16)
    # power-of-two channel counts of BASELINE.json configs (exact)
    for M, word in kat["predicted_pow2"].items():
        assert O.nco_constrain(O.pfb_offset(int(M))) == int(word, 16)
        assert O.Chan(int(M)).dtheta == int(word, 16)


def test_kat3_dc_blocker_form():
    kat = _kat()["kat3_dcblock"]
    # the GIF was recorded with alpha = 0.001
    q = O.DcBlock(kat["alpha_at_recording"])
    assert abs(q.a1 - kat["a"][1]) < 1e-8
    # structure b=[1,-1], a=[1,a1]: impulse response is 1, (beta-1), (beta-1)beta, ...
    q = O.DcBlock(0.0005)
    assert q.a1 == np.float32(-1.0) + np.float32(0.0005)
    imp = np.zeros(8, dtype=np.complex64)
    imp[0] = 1
    y = q.execute(imp)
    beta = 1 - 0.0005
    want = np.array([1.0] + [(beta - 1) * beta ** (i - 1) for i in range(1, 8)])
    assert np.allclose(y.real, want, atol=1e-7)
    assert np.all(y.imag == 0)


def test_kat4_example3_sizes():
    kat = _kat()["kat4_example3"]
    n, M = kat["numsamples"], kat["channels"]
    assert (n // M) * 8 == kat["bytes_per_file"]
    # one compacted chunk of 4*M*1024 gives exactly 4096 samples per channel
    x = np.zeros(4 * M * 1024, dtype=np.complex64)
    y = O.Chan(M).process(x)
    assert y.shape == (M, 4096)


def test_kat5_per_channel_files_grow_in_32768_byte_quanta():
    """KAT5 (images/ex1_5.gif frames 40 / 105: the per-channel files of the Example 3 run grow in 32 768-byte steps): `compact (4 * nch * 1024)`
    (SoapySDR.hs:215) hands the channelizer 4096 frames at a time, so every `distribute_` write (Trans.hs:106-117) is 4096 CF32 = 32 768
    bytes per channel -- whatever the source's chunk size (1000 here, not a divisor)."""
    from composable_sdr_amd.pipes import Pipe
    from composable_sdr_amd.trans import addPipe, collect, compact, distribute_
    kat = _kat()["kat5_file_quanta"]
    M = kat["channels"]
    assert 4 * M * 1024 == kat["compact_samples"] and kat["frames_per_chunk"] * 8 == kat["bytes_per_write"]
    chan = O.Chan(M)
    pfb = Pipe(lambda: None, lambda r, a: list(chan.process(a)) if a.size else [a], lambda r: None)   # firpfbchChannelizer on the oracle
    sinks = [collect() for _ in range(M)]
    fold = compact(4 * M * 1024, addPipe(pfb, distribute_(sinks)))
    x = np.zeros(3 * kat["compact_samples"], dtype=np.complex64)
    sizes = [[] for _ in range(M)]
    for a in np.array_split(x, x.size // 1000):
        fold.step(a)
        for k, s_ in enumerate(sinks):
            sizes[k].append(sum(i.nbytes for i in s_.items))
    for k in range(M):
        assert all(v % kat["bytes_per_write"] == 0 for v in sizes[k]) and sizes[k][-1] >= 2 * kat["bytes_per_write"]
