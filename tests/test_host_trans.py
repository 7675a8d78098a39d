"""Host-side combinators (composable_sdr_amd/trans.py) against the semantics of
/root/reference/src/ComposableSDR/Trans.hs and Types.hs (cited per test)."""
import numpy as np

from composable_sdr_amd.pipes import Pipe, compose
from composable_sdr_amd.trans import addPipe, collect, compact, distribute_, mix, mux, takeNArr


def _arr(a, b):
    return np.arange(a, b, dtype=np.float32)


def test_takeNArr_trims_the_last_array_and_stops():          # Trans.hs:33-56
    src = [_arr(0, 4), _arr(4, 8), _arr(8, 12)]
    out = list(takeNArr(6, src))
    assert [len(a) for a in out] == [4, 2] and np.array_equal(np.concatenate(out), _arr(0, 6))
    assert [len(a) for a in takeNArr(8, src)] == [4, 4]      # exact boundary: third array not emitted
    assert list(takeNArr(0, src)) == []
    assert [len(a) for a in takeNArr(100, src)] == [4, 4, 4]


def test_compact_emits_exactly_n_and_flushes_remainder():    # Trans.hs:58-84
    sink = collect()
    f = compact(5, sink)
    for a in (_arr(0, 3), _arr(3, 6), _arr(6, 7), _arr(7, 13)):
        f.step(a)
    f.done()
    lens = [len(a) for a in sink.items]
    # 3 (buffer) -> 6: emit 5 keep 1 -> 7: keep 2 -> 13: emit 5 keep 3 ; done: flush 3
    assert lens == [5, 5, 3]
    assert np.array_equal(np.concatenate(sink.items), _arr(0, 13))
    # an input larger than 2n emits n and keeps the (>= n) rest until the next step / done
    sink = collect()
    f = compact(4, sink)
    f.step(_arr(0, 11))
    f.done()
    assert [len(a) for a in sink.items] == [4, 7]
    # empty remainder is still pushed downstream at end of stream (Trans.hs:66-68)
    sink = collect()
    f = compact(4, sink)
    f.step(_arr(0, 4))
    f.done()
    assert [len(a) for a in sink.items] == [4, 0]


def test_mix_is_left_fold_and_mux_keeps_per_channel_state():  # Trans.hs:119-129
    chans = [np.float32([1e8, 1.0]), np.float32([-1e8, 1.0]), np.float32([1.0, 1.0])]
    out = mix._process(None, chans)
    assert np.array_equal(out, np.float32([1.0, 3.0]))        # ((1e8 + -1e8) + 1), not 1e8 + (-1e8 + 1)
    starts = []

    def counter():
        st = {"n": 0}
        starts.append(st)
        return st

    def proc(st, a):
        st["n"] += len(a)
        return a + st["n"]
    p = Pipe(counter, proc, lambda st: st.update(done=True))
    m = mux([p, p, p])                                        # replicate nch demod: same Pipe VALUE
    r = m._start()
    assert len(starts) == 3                                   # but one state per channel (Trans.hs:127)
    o1 = m._process(r, [_arr(0, 2), _arr(0, 3), _arr(0, 1)])
    o2 = m._process(r, [_arr(0, 2), _arr(0, 3), _arr(0, 1)])
    assert [st["n"] for st in starts] == [4, 6, 2]
    assert np.array_equal(o2[1], _arr(0, 3) + 6) and np.array_equal(o1[2], _arr(0, 1) + 1)
    m._done(r)
    assert all(st.get("done") for st in starts)


def test_compose_order_and_addPipe_lifecycle():                # Types.hs:93-131
    log = []
    a = Pipe(lambda: log.append("startA") or "A", lambda r, x: x + 1, lambda r: log.append("doneA"))
    b = Pipe(lambda: log.append("startB") or "B", lambda r, x: x * 2, lambda r: log.append("doneB"))
    c = compose(a, b)                                         # a . b : b first, then a
    r = c._start()
    assert c._process(r, np.float32([1, 2])).tolist() == [3.0, 5.0]
    c._done(r)
    assert log == ["startA", "startB", "doneB", "doneA"]
    assert (a @ b)._process((None, None), np.float32([1])).tolist() == [3.0]
    sink = collect()
    f = addPipe(b, sink)
    f.step(np.float32([1, 2])).step(np.float32([3]))
    f.done()
    assert [x.tolist() for x in sink.items] == [[2.0, 4.0], [6.0]]


def test_unPipe_creates_now_and_returns_stream_map_and_cleanup():   # Types.hs:109-115, SoapySDR.hs:206,282
    from composable_sdr_amd.pipes import idPipe, unPipe
    log = []
    a = Pipe(lambda: log.append("start") or {"n": 0}, lambda r, x: x + 1, lambda r: log.append("done"))
    process, cleanup = unPipe(compose(a, idPipe))             # resampler . offset with offset = id
    assert log == ["start"]                                   # r <- creat happens inside unPipe
    out = list(process(iter([np.float32([1]), np.float32([2, 3])])))
    assert [o.tolist() for o in out] == [[2.0], [3.0, 4.0]] and log == ["start"]
    cleanup()
    assert log == ["start", "done"]
    assert idPipe._process(idPipe._start(), 7) == 7


def test_time_stripes_reject_tails_with_unwarmed_memory():     # am / wbfm / agc in time mode
    from composable_sdr_amd.pipes import ChainConfig
    from composable_sdr_amd.sharded import ShardedChain
    import pytest
    for kw in (dict(demod="am"), dict(demod="wbfm"), dict(demod="fm", agc=10.0)):
        with pytest.raises(ValueError):
            ShardedChain(ChainConfig(channels=16, **kw), mode="time", rank=1, world=2, chain_factory=lambda c: None)
    ShardedChain(ChainConfig(channels=16, demod="fm"), mode="time", rank=1, world=2, chain_factory=lambda c: None)


def test_distribute_routes_channel_k_to_sink_k():              # Trans.hs:106-117, SoapySDR.hs:209-212
    sinks = [collect() for _ in range(3)]
    d = distribute_(sinks)
    d.step([_arr(0, 2), _arr(10, 12), _arr(20, 22)])
    d.step([_arr(2, 4)])                                      # a single (empty-chunk) element reaches sink 1 only
    d.done()
    assert np.array_equal(sinks[0].concat(), _arr(0, 4))
    assert np.array_equal(sinks[1].concat(), _arr(10, 12)) and np.array_equal(sinks[2].concat(), _arr(20, 22))


def test_cpp_host_combinators_build_and_pass():
    """The C++ mirror (composable_sdr_amd/host/csdr_host.hpp) runs its own semantic checks."""
    import os
    import subprocess
    host = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "composable_sdr_amd", "host")
    subprocess.check_call(["make", "-C", host, "-s", "test_host"])
    out = subprocess.run([os.path.join(host, "test_host")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "host combinators ok" in out.stdout, out.stderr


def test_audio_file_sink_layouts(tmp_path):
    """audioFileSink (Sink.hs:41-74): libsndfile float / big-endian AU and WAV.  AU = 24-byte header + BE floats;
    WAV = RIFX + fmt(tag 3) + fact + data (libsndfile adds a time-stamped PEAK chunk: not reproducible)."""
    import struct
    from composable_sdr_amd.app import audioFileSink
    x = np.linspace(-1, 1, 1000, dtype=np.float32)
    s = audioFileSink("AU", 48000, 1000, 1, str(tmp_path / "a"))
    s.step(x[:300]); s.step(x[300:]); s.done()
    b = open(s.path, "rb").read()
    assert s.path.endswith(".au") and struct.unpack(">4sIIIII", b[:24]) == (b".snd", 24, 4000, 6, 48000, 1)
    assert np.array_equal(np.frombuffer(b[24:], dtype=">f4").astype(np.float32), x)
    s = audioFileSink("WAV", 44100, 1000, 1, str(tmp_path / "w"))
    s.step(x); s.done()
    b = open(s.path, "rb").read()
    assert s.path.endswith(".wav") and struct.unpack(">4sI4s", b[:12]) == (b"RIFX", len(b) - 8, b"WAVE")
    assert struct.unpack(">4sIHHIIHH", b[12:36]) == (b"fmt ", 16, 3, 1, 44100, 176400, 4, 32)
    assert struct.unpack(">4sII", b[36:48]) == (b"fact", 4, 1000) and struct.unpack(">4sI", b[48:56]) == (b"data", 4000)
    assert np.array_equal(np.frombuffer(b[56:], dtype=">f4").astype(np.float32), x)


def test_wav_sink_is_read_back_by_an_independent_parser(tmp_path):
    """audioFileSink's RIFX / IEEE-float WAV through scipy.io.wavfile (a parser that is not ours): rate, dtype, frame count and every
    sample, mono and interleaved stereo, without a warning about the chunk structure."""
    import warnings
    from scipy.io import wavfile
    from composable_sdr_amd.app import audioFileSink
    rng = np.random.default_rng(5)
    for nch, n in ((1, 1000), (2, 777)):
        x = rng.standard_normal(n * nch).astype(np.float32)
        s = audioFileSink("WAV", 48000 // nch, n, nch, str(tmp_path / f"w{nch}"))
        s.step(x[: 100 * nch]); s.step(x[100 * nch:]); s.done()
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            sr, data = wavfile.read(s.path)
        assert sr == 48000 // nch and data.dtype.kind == "f" and data.dtype.itemsize == 4
        assert data.shape == ((n,) if nch == 1 else (n, nch))
        assert np.array_equal(np.asarray(data, dtype=np.float32).reshape(-1), x)
