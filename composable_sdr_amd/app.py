"""Replay of `sdrProcess` / `assembleFold` (apps/SoapySDR.hs:181-283) for the file-input path:

    readFromFile chunksize fp        Source.chs:259-271   headerless LE float32 I/Q, <= chunksize samples per array
      -> takeNArr n                  Trans.hs:33-56
      -> dcBlocker -> compact (4*nch*1024) -> PFB -> per-channel demod -> sinks   SoapySDR.hs:208-226

with the DSP behind `compact` done by the fused C-ABI chain.  Sinks are the reference's raw
`fileSink`s (Sink.hs:29-34): `<out>.cf32` / `<out>_ch<k>.cf32` for DeNo (SoapySDR.hs:240); demodulated audio goes to
raw `.f32` by default or, with `audio="AU" | "WAV"`, through `audioFileSink` (Sink.hs:41-74: libsndfile float,
big-endian; written here by hand, see the class).  `--offset` and `-b` are the reference's mixDown/mixUp and
resampler in front of `takeNArr`."""
import struct
import numpy as np

from .pipes import Chain, ChainConfig, compose, idPipe, mixDown, mixUp, resampler, unPipe
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
        d = self.chain.decim                          # DeWBFM: firDecimator's n = length `div` m drops the leftover (Liquid.chs:495-497)
        if d > 1:
            usable = usable // (M * d) * (M * d)
            if usable == 0:
                for s in (self.sinks[:1] if (self.mixed or M == 1) else self.sinks):
                    s.step(np.empty(0, dtype=np.float32))
                return self
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


class audioFileSink(Fold):
    """audioFileSink fmt sr sn nch fp (Sink.hs:41-74): libsndfile, SampleFormatFloat, EndianBig, file fp + ".au" / ".wav".

    AU  : 24-byte header (".snd", data offset 24, data bytes, encoding 6 = 32-bit IEEE float, rate, channels), all
          big-endian, then big-endian floats -- what libsndfile's au_write_header leaves after close.
    WAV : libsndfile writes RIFX (big-endian WAV) for EndianBig with fmt (tag 3), fact and a PEAK chunk that holds a
          time stamp, so its bytes are not reproducible anyway; this writer emits RIFX + fmt + fact + data (no PEAK).
    libsndfile is not in the image: the layout is from the format definitions, unverified against the library."""

    def __init__(self, fmt, sr, sn, nch, fp):
        self.fmt, self.sr, self.nch = fmt.upper(), int(sr), int(nch)
        assert self.fmt in ("AU", "WAV")
        self.path = fp + (".au" if self.fmt == "AU" else ".wav")
        self.f = open(self.path, "wb")
        self.nbytes = 0
        self._header(0xffffffff if self.fmt == "AU" else 0)

    def _header(self, nbytes):
        self.f.seek(0)
        if self.fmt == "AU":
            self.f.write(struct.pack(">4sIIIII", b".snd", 24, nbytes, 6, self.sr, self.nch))
        else:
            frames = nbytes // (4 * self.nch)
            self.f.write(struct.pack(">4sI4s", b"RIFX", 4 + 24 + 12 + 8 + nbytes, b"WAVE"))
            self.f.write(struct.pack(">4sIHHIIHH", b"fmt ", 16, 3, self.nch, self.sr, self.sr * 4 * self.nch, 4 * self.nch, 32))
            self.f.write(struct.pack(">4sII", b"fact", 4, frames))
            self.f.write(struct.pack(">4sI", b"data", nbytes))

    def step(self, a):
        if len(a) == 0:
            return self
        b = np.ascontiguousarray(a, dtype=np.float32).astype(">f4").tobytes()
        self.f.write(b)
        self.nbytes += len(b)
        return self

    def done(self):
        self._header(self.nbytes)
        self.f.close()


def sdr_process(filename, channels=1, demod="none", kf=0.3, agc=0.0, mix=False, numsamples=1024,
                outname="output", chunksize=1024, m=4, offset=0.0, samplerate=2.56e6, bandwidth=0.0, decim=4, audio=None):
    """soapy-sdr --filename F -s samplerate -b bandwidth --offset f -c channels --demod ... -a agc [-m]
    -n numsamples -o outname.  Returns the list of files written."""
    nch = channels
    mixed = bool(mix) and nch > 1
    ext = ".cf32" if demod == "none" else ".f32"
    stems = [outname] if (mixed or nch == 1) else [f"{outname}_ch{k}" for k in range(1, nch + 1)]
    out_bw = bandwidth if bandwidth != 0 else samplerate
    if audio and demod != "none":
        # getAudioSink decim fmt chn (SoapySDR.hs:232-234): rate = round outBW `div` decim `div` nch, mono
        dec = decim if demod == "wbfm" else 1
        sinks = [audioFileSink(audio, int(round(out_bw)) // dec // nch, numsamples, 1, st) for st in stems]
        names = [sk.path for sk in sinks]
    else:
        names = [st + ext for st in stems]
        sinks = [fileSink(n) for n in names]
    # DeWBFM decim: de-emphasis corner 5000 / outBW, outBW = bandwidth or the sample rate (SoapySDR.hs:227-231, Liquid.chs:655)
    chain = Chain(ChainConfig(channels=nch, demod=demod, kf=kf, agc=agc, mix=mixed, max_frames=m * 1024, decim=decim,
                              deemph_fc=float(np.float32(5000.0 / out_bw))))
    fold = compact(m * nch * 1024, _FusedFold(chain, sinks, mixed))
    # prep = takeNArr ns . (resampler . offset)   (SoapySDR.hs:206-207): per source chunk, the --offset mixer first
    # (f = 2*pi*offset/fs; mixDown f if f > 0, mixUp (-f) if f < 0, :200-205), then the resampler
    # (rate = bandwidth / samplerate, 60 dB, identity when -b 0, :190-194)
    f = np.float32(2 * np.pi * offset / samplerate)
    offset_p = idPipe
    if f != 0:
        offset_p = mixDown(float(f), max_samples=chunksize) if f > 0 else mixUp(float(-f), max_samples=chunksize)
    resamp_p = idPipe
    if bandwidth != 0:
        resamp_p = resampler(float(np.float32(bandwidth / samplerate)), 60.0, max_samples=chunksize)
    process, cleanup = unPipe(compose(resamp_p, offset_p))       # (process, cleanup) <- unPipe (resampler . offset)
    try:
        for a in takeNArr(numsamples, process(readFromFile(chunksize, filename))):
            fold.step(a)
    finally:
        fold.done()
        cleanup()
    return names
