"""Replay of `sdrProcess` / `assembleFold` (apps/SoapySDR.hs:181-283) for the file-input path:

    readFromFile chunksize fp        Source.chs:259-271   headerless LE float32 I/Q, <= chunksize samples per array
      -> takeNArr n                  Trans.hs:33-56
      -> dcBlocker -> compact (4*nch*1024) -> PFB -> per-channel demod -> sinks   SoapySDR.hs:208-226

with the DSP behind `compact` done by the fused C-ABI chain.  Sinks are the reference's raw
`fileSink`s (Sink.hs:29-34): `<out>.cf32` / `<out>_ch<k>.cf32` for DeNo (SoapySDR.hs:240), and raw
`.f32` for FM (the reference wraps FM audio in WAV/AU through libsndfile, which is out of scope;
the sample values and their order are the same).  `--offset` is the reference's mixDown/mixUp in front; the msresamp resampler (`-b`) is not built (f2)."""
import numpy as np

from .pipes import Chain, ChainConfig, mixDown, mixUp, resampler
from .trans import Fold, compact, takeNArr


def readFromFile(n, fp):
    """Source.chs:259-271: stream of arrays of at most n CF32 samples."""
    with open(fp, "rb") as f:
        while True:
            b = f.read(8 * n)
            if not b:
                return
            yield np.frombuffer(b[: len(b) // 8 * 8], dtype=np.complex64)


class fileSink(Fold):
    """Sink.hs:29-34: FS.writeChunks"""

    def __init__(self, path):
        self.f = open(path, "wb")

    def step(self, a):
        self.f.write(np.ascontiguousarray(a).tobytes())
        return self

    def done(self):
        self.f.close()


class _FusedFold(Fold):
    """addPipe fused (distribute_ sinks | sink): one chain call per compacted chunk, then the
    channel-major buffer is sliced exactly like Liquid.chs:850-862 and handed to sink k+1."""

    def __init__(self, chain, sinks, mixed):
        self.chain, self.sinks, self.mixed = chain, sinks, mixed

    def step(self, a):
        M = self.chain.M
        if len(a) == 0:
            self.sinks[0].step(a[:0])                 # nx = 0 -> [empty]: only sink 1 sees it
            return self
        usable = len(a) // M * M                      # a ragged stream tail cannot be channelized
        y = self.chain.process(a[:usable])
        if self.mixed or M == 1:
            self.sinks[0].step(y.reshape(-1))
        else:
            for k, s in enumerate(self.sinks):
                s.step(y[k])
        return self

    def done(self):
        self.chain.close()
        for s in self.sinks:
            s.done()


def sdr_process(filename, channels=1, demod="none", kf=0.3, agc=0.0, mix=False, numsamples=1024,
                outname="output", chunksize=1024, m=4, offset=0.0, samplerate=2.56e6, bandwidth=0.0, decim=4):
    """soapy-sdr --filename F -s samplerate -b bandwidth --offset f -c channels --demod ... -a agc [-m]
    -n numsamples -o outname.  Returns the list of files written."""
    nch = channels
    mixed = bool(mix) and nch > 1
    ext = ".cf32" if demod == "none" else ".f32"
    names = [outname + ext] if (mixed or nch == 1) else [f"{outname}_ch{k}{ext}" for k in range(1, nch + 1)]
    # DeWBFM decim: de-emphasis corner 5000 / outBW, outBW = bandwidth or the sample rate (SoapySDR.hs:227-231, Liquid.chs:655)
    out_bw = bandwidth if bandwidth != 0 else samplerate
    chain = Chain(ChainConfig(channels=nch, demod=demod, kf=kf, agc=agc, mix=mixed, max_frames=m * 1024, decim=decim,
                              deemph_fc=float(np.float32(5000.0 / out_bw))))
    fold = compact(m * nch * 1024, _FusedFold(chain, [fileSink(n) for n in names], mixed))
    # prep = takeNArr ns . (resampler . offset)   (SoapySDR.hs:206-207): per source chunk, the --offset mixer first
    # (f = 2*pi*offset/fs; mixDown f if f > 0, mixUp (-f) if f < 0, :200-205), then the resampler
    # (rate = bandwidth / samplerate, 60 dB, identity when -b 0, :190-194)
    f = np.float32(2 * np.pi * offset / samplerate)
    stages = []
    if f != 0:
        stages.append(mixDown(float(f), max_samples=chunksize) if f > 0 else mixUp(float(-f), max_samples=chunksize))
    if bandwidth != 0:
        stages.append(resampler(float(np.float32(bandwidth / samplerate)), 60.0, max_samples=chunksize))
    states = [p._start() for p in stages]
    src = readFromFile(chunksize, filename)

    def front(gen):
        for a in gen:
            for p, st in zip(stages, states):
                a = p._process(st, a)
            yield a
    for a in takeNArr(numsamples, front(src) if stages else src):
        fold.step(a)
    fold.done()
    for p, st in zip(stages, states):
        p._done(st)
    return names
