"""Host-side mirror of the reference's DSP blocks (src/ComposableSDR/Liquid.chs) and of
the Pipe protocol (src/ComposableSDR/Types.hs:51-55, 93-131) over the C ABI.

A Pipe is (start, process, done); `start` creates the native object, `process` maps one
array to one array, `done` destroys it -- exactly the life-cycle addPipe/unPipe drive.
Names follow the reference: dcBlocker, mixDown, mixUp, automaticGainControl,
fmDemodulator, firpfbchChannelizer.  `Chain` is the fused replacement for
`mix . mux (replicate nch demod) . firpfbchChannelizer nc` (apps/SoapySDR.hs:218-225).
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import CsdrError, check, lib, DEMOD_AM, DEMOD_FM, DEMOD_NONE, DEMOD_WBFM


def _c64(x):
    return np.ascontiguousarray(x, dtype=np.complex64)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Pipe:
    """Pipe {_start, _process, _done} (Types.hs:51-55)."""

    def __init__(self, start, process, done):
        self._start, self._process, self._done = start, process, done

    # Category instance: (.) = compose (Types.hs:101-103); `self . other` runs other first
    def __matmul__(self, other):
        return compose(self, other)


def compose(p1, p2):
    """compose (Types.hs:93-99): process = process2 >=> process1; done2 before done1."""
    def start():
        return (p1._start(), p2._start())

    def process(r, a):
        return p1._process(r[0], p2._process(r[1], a))

    def done(r):
        p2._done(r[1])
        p1._done(r[0])
    return Pipe(start, process, done)


# Category instance: id (Types.hs:101-102)
idPipe = Pipe(lambda: None, lambda r, a: a, lambda r: None)


def unPipe(pipe):
    """unPipe (Types.hs:109-115): create the pipe's resource NOW and hand back
    (stream transformer = S.mapM (process r), cleanup = dest r).  The caller runs
    `cleanup` after the stream has been folded (SoapySDR.hs:206, :282)."""
    r = pipe._start()

    def process(stream):
        for a in stream:
            yield pipe._process(r, a)

    return process, (lambda: pipe._done(r))


class _Handle:
    """Owns one native handle; destroy is idempotent."""

    def __init__(self, h, destroy):
        self.h, self._destroy = h, destroy

    def close(self):
        if self.h:
            check(self._destroy(self.h))
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def dcBlocker(alpha=0.0005, max_samples=1 << 20):
    """dcBlocker (Liquid.chs:575-589): iirfilt_crcf_create_dc_blocker(0.0005)."""
    def start():
        h = C.c_void_p()
        check(lib().csdr_dcblock_create(alpha, max_samples, C.byref(h)))
        return _Handle(h, lib().csdr_dcblock_destroy)

    def process(r, a):
        x = _c64(a)
        y = np.empty_like(x)
        check(lib().csdr_dcblock_process(r.h, _ptr(x), x.size, _ptr(y)))
        return y
    return Pipe(start, process, lambda r: r.close())


def _mixer(f, up, max_samples):
    def start():
        h = C.c_void_p()
        check(lib().csdr_nco_create(f, max_samples, C.byref(h)))
        return _Handle(h, lib().csdr_nco_destroy)

    def process(r, a):
        x = _c64(a)
        y = np.empty_like(x)
        fn = lib().csdr_nco_mix_up if up else lib().csdr_nco_mix_down
        check(fn(r.h, _ptr(x), x.size, _ptr(y)))
        return y
    return Pipe(start, process, lambda r: r.close())


def mixDown(f, max_samples=1 << 20):
    """mixDown f (Liquid.chs:798-799): y = x * conj(nco)."""
    return _mixer(f, False, max_samples)


def mixUp(f, max_samples=1 << 20):
    """mixUp f (Liquid.chs:808-809): y = x * nco."""
    return _mixer(f, True, max_samples)


def automaticGainControl(tres, nchan=1, max_samples=4096):
    """automaticGainControl tres (Liquid.chs:727-728); nchan independent instances, input
    and output channel-major [nchan][n] (1-D arrays are treated as one channel)."""
    def start():
        h = C.c_void_p()
        check(lib().csdr_agc_create(tres, nchan, max_samples, C.byref(h)))
        return _Handle(h, lib().csdr_agc_destroy)

    def process(r, a):
        x = _c64(a)
        n = x.size // nchan
        y = np.empty_like(x)
        check(lib().csdr_agc_process(r.h, _ptr(x), n, _ptr(y)))
        return y
    return Pipe(start, process, lambda r: r.close())


def fmDemodulator(kf, nchan=1, max_samples=4096):
    """fmDemodulator kf (Liquid.chs:333-334)."""
    def start():
        h = C.c_void_p()
        check(lib().csdr_freqdem_create(kf, nchan, max_samples, C.byref(h)))
        return _Handle(h, lib().csdr_freqdem_destroy)

    def process(r, a):
        x = _c64(a)
        n = x.size // nchan
        m = np.empty(x.shape, dtype=np.float32)
        check(lib().csdr_freqdem_process(r.h, _ptr(x), n, _ptr(m)))
        return m
    return Pipe(start, process, lambda r: r.close())


def iirFilter(n, fc, f0=0.0, ap=10.0, as_db=10.0, nchan=1, max_samples=4096):
    """iirFilter n fc f0 ap as (Liquid.chs:629-638): real-valued Butterworth low-pass (order 2 is what the
    reference instantiates, as the WBFM de-emphasis)."""
    def start():
        h = C.c_void_p()
        check(lib().csdr_iirfilt_create(n, fc, f0, ap, as_db, nchan, max_samples, C.byref(h)))
        return _Handle(h, lib().csdr_iirfilt_destroy)

    def process(r, a):
        x = np.ascontiguousarray(a, dtype=np.float32)
        y = np.empty_like(x)
        check(lib().csdr_iirfilt_process(r.h, _ptr(x), x.size // nchan, _ptr(y)))
        return y
    return Pipe(start, process, lambda r: r.close())


def firDecimator(m, nchan=1, max_samples=4096):
    """firDecimator m (Liquid.chs:500-501): Kaiser decimator (semi-length 10, 60 dB), n div m samples out."""
    def start():
        h = C.c_void_p()
        check(lib().csdr_firdecim_create(m, nchan, max_samples, C.byref(h)))
        return _Handle(h, lib().csdr_firdecim_destroy)

    def process(r, a):
        x = np.ascontiguousarray(a, dtype=np.float32)
        n = x.size // nchan
        y = np.empty(x.shape[:-1] + (n // m,) if x.ndim > 1 else (n // m,), dtype=np.float32)
        check(lib().csdr_firdecim_process(r.h, _ptr(x), n, _ptr(y)))
        return y
    return Pipe(start, process, lambda r: r.close())


def wbFMDemodulator(quadRate, decim, max_samples=4096):
    """wbFMDemodulator quadRate decim = firDecimator decim . iirDeemph . fmDemodulator 0.6 (Liquid.chs:653-656)"""
    return compose(firDecimator(decim, max_samples=max_samples),
                   compose(iirFilter(2, 5000.0 / quadRate, 0.0, 10.0, 10.0, max_samples=max_samples),
                           fmDemodulator(0.6, max_samples=max_samples)))


def resampler(r, as_db=60.0, max_samples=1 << 20):
    """resampler r as (Liquid.chs:115-117): Pipe IO (Array CF32) (Array CF32) with a variable-length output
    (`shrinkToFit` to the count msresamp_crcf_execute reports, :79-98).  r == 0 is the identity."""
    def start():
        h = C.c_void_p()
        check(lib().csdr_resamp_create(r, as_db, max_samples, C.byref(h)))
        return _Handle(h, lib().csdr_resamp_destroy)

    def process(rh, a):
        x = _c64(a)
        y = np.empty(int(lib().csdr_resamp_max_out(rh.h, x.size)), dtype=np.complex64)
        n = C.c_uint32()
        check(lib().csdr_resamp_process(rh.h, _ptr(x), x.size, _ptr(y), C.byref(n)))
        return y[:n.value].copy()
    return Pipe(start, process, lambda rh: rh.close())


def amDemodulator(nchan=1, max_samples=4096, mod_index=0.8):
    """amDemodulator (Liquid.chs:468-469): ampmodem_create 0.8 DSB, carrier present."""
    def start():
        h = C.c_void_p()
        check(lib().csdr_ampdem_create(mod_index, nchan, max_samples, C.byref(h)))
        return _Handle(h, lib().csdr_ampdem_destroy)

    def process(r, a):
        x = _c64(a)
        n = x.size // nchan
        m = np.empty(x.shape, dtype=np.float32)
        check(lib().csdr_ampdem_process(r.h, _ptr(x), n, _ptr(m)))
        return m
    return Pipe(start, process, lambda r: r.close())


@dataclass
class ChainConfig:
    channels: int = 1
    dc_block: bool = True
    dc_alpha: float = 0.0005
    agc: float = 0.0            # -a; 0 = off
    demod: str = "none"         # "none" (DeNo) | "fm" (DeNBFM kf) | "am" (DeAM) | "wbfm" (DeWBFM decim)
    decim: int = 4              # DeWBFM decim
    deemph_fc: float = 0.025    # 5000 / quadRate (Liquid.chs:655)
    kf: float = 0.3
    mix: bool = False
    chan_first: int = 0
    chan_count: int = 0
    chan_stride: int = 0        # G > 1: interleaved ownership, channels chan_first + G*m (pruned DFT)
    device: int = -1
    max_frames: int = 4096
    flags: int = _lib.FLAG_QUIET
    pfb_m: int = 7
    pfb_as: float = 80.0
    tail_only: bool = False     # CSDR_FLAG_TAIL_ONLY: the per-channel tail alone on a channel-major CF32 plane [channels][nf]
    dft_backward: bool = False  # CSDR_FLAG_DFT_BACKWARD: the analyzer's transform as e^{+j} (row k = forward row (M - k) mod M); include/csdr.h


class Chain:
    """The fused chain object behind `csdr_chain_*`."""

    def __init__(self, cfg: ChainConfig = None, **kw):
        cfg = cfg or ChainConfig(**kw)
        self.cfg = cfg
        c = _lib.ChainCfg()
        lib().csdr_chain_cfg_default(C.byref(c), cfg.channels)
        c.channels = cfg.channels
        c.dc_block, c.dc_alpha = int(cfg.dc_block), cfg.dc_alpha
        c.agc_threshold_db = cfg.agc
        c.demod = {"none": DEMOD_NONE, "fm": DEMOD_FM, "am": DEMOD_AM, "wbfm": DEMOD_WBFM}[cfg.demod]
        c.wbfm_decim, c.deemph_fc = cfg.decim, cfg.deemph_fc
        c.kf, c.mix = cfg.kf, int(cfg.mix)
        c.chan_first, c.chan_count = cfg.chan_first, cfg.chan_count
        c.chan_stride = cfg.chan_stride
        c.device, c.max_frames, c.flags = cfg.device, cfg.max_frames, cfg.flags | (_lib.FLAG_TAIL_ONLY if cfg.tail_only else 0) | (_lib.FLAG_DFT_BACKWARD if cfg.dft_backward else 0)
        c.pfb_m, c.pfb_as = cfg.pfb_m, cfg.pfb_as
        h = C.c_void_p()
        check(lib().csdr_chain_create(C.byref(c), C.byref(h)))
        self._h = _Handle(h, lib().csdr_chain_destroy)
        self.M = cfg.channels
        self.C = cfg.channels // cfg.chan_stride if cfg.chan_stride > 1 else (cfg.chan_count or (cfg.channels - cfg.chan_first))
        self.mixed = bool(cfg.mix) and self.M > 1
        self.out_dtype = np.float32 if cfg.demod in ("fm", "am", "wbfm") else np.complex64
        self.decim = cfg.decim if cfg.demod == "wbfm" else 1

    @property
    def h(self):
        if not self._h.h:
            raise CsdrError(_lib.ERR_INVALID, "chain already destroyed")
        return self._h.h

    @property
    def path(self):
        return lib().csdr_chain_path(self.h).decode()

    @property
    def taps(self):
        n = self.M * 2 * self.cfg.pfb_m
        t = np.zeros(n, dtype=np.float32)
        got = lib().csdr_chain_get_taps(self.h, _ptr(t), n)
        return t[:max(got, 0)]

    @property
    def nco(self):
        th, d = C.c_uint32(), C.c_uint32()
        check(lib().csdr_chain_get_nco(self.h, C.byref(th), C.byref(d)))
        return th.value, d.value

    def out_shape(self, n_in):
        nf = n_in // self.M // self.decim
        return (nf,) if self.mixed else (self.C, nf)

    def process(self, x):
        """One compacted chunk (host arrays) -> channel-major [C][nf] (or [nf] when mixing)."""
        x = _c64(x)
        out = np.empty(self.out_shape(x.size), dtype=self.out_dtype)
        n_out = C.c_uint32()
        check(lib().csdr_chain_process(self.h, _ptr(x), x.size, _ptr(out), C.byref(n_out)))
        if x.size == 0:
            return out
        assert n_out.value == out.size, (n_out.value, out.size)
        return out

    def process_device(self, d_in_ptr, n_in, d_out_ptr, stream=0):
        """Device-resident variant: raw device pointers (ints), enqueues on `stream`."""
        n_out = C.c_uint32()
        check(lib().csdr_chain_process_device(self.h, C.c_void_p(d_in_ptr), n_in, C.c_void_p(d_out_ptr),
                                              C.byref(n_out), C.c_void_p(stream)))
        return n_out.value

    # ---- pipelined device entry point (csdr_chain_submit_device / csdr_chain_wait_device)
    def submit_device(self, d_in_ptr, n_in, d_out_ptr, ready_event=0):
        """Queue one HBM-resident chunk on the handle's own streams; consecutive chunks' launches may overlap (include/csdr.h).
        `d_in_ptr` must stay untouched and `d_out_ptr` unread until wait_device()."""
        n_out = C.c_uint32()
        check(lib().csdr_chain_submit_device(self.h, C.c_void_p(d_in_ptr), n_in, C.c_void_p(d_out_ptr), C.byref(n_out), C.c_void_p(ready_event)))
        return n_out.value

    def independent_launches(self):
        return lib().csdr_chain_debug_independent_launches(self.h)

    def wait_device(self, stream=None):
        """Host (stream=None) or `stream` waits for every chunk queued with submit_device()."""
        check(lib().csdr_chain_wait_device(self.h, C.c_void_p(stream) if stream is not None else None))

    # ---- asynchronous host-buffer entry point (csdr_chain_submit / csdr_chain_collect)
    def submit(self, x, out=None):
        """Queue one chunk; `x` / `out` should come from host_array() (page-locked) for full PCIe speed.  Returns `out`,
        which is valid after the matching collect()."""
        x = _c64(x)
        if out is None:
            out = np.empty(self.out_shape(x.size), dtype=self.out_dtype)
        check(lib().csdr_chain_submit(self.h, _ptr(x), x.size, _ptr(out)))
        self._pending = getattr(self, "_pending", [])
        self._pending.append((x, out))                   # keep the buffers alive until collect
        return out

    def collect(self):
        """Wait for the oldest submitted chunk; returns its output array."""
        n_out = C.c_uint32()
        check(lib().csdr_chain_collect(self.h, C.byref(n_out)))
        x, out = self._pending.pop(0)
        assert n_out.value == out.size or x.size == 0, (n_out.value, out.size)
        return out

    def status(self):
        """raises CsdrError if a device-side inter-workgroup wait timed out since the last check"""
        check(lib().csdr_chain_status(self.h))

    def kernel_time(self):
        ms, n = C.c_double(), C.c_uint32()
        name = lib().csdr_chain_kernel_time(self.h, C.byref(ms), C.byref(n))
        return name.decode(), ms.value, n.value

    def agc_stats(self):
        """(segments checked, segments recomputed) of the time-parallel AGC tail since create"""
        a, b = C.c_uint32(), C.c_uint32()
        check(lib().csdr_chain_debug_agc(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def agc_tile_major_calls(self):
        """calls since create whose AGC tail ran on a tile-major plane (k_agc_spec_tm)"""
        return lib().csdr_chain_debug_agc_tile_major_calls(self.h)

    def reset(self):
        check(lib().csdr_chain_reset(self.h))
        self._pending = []                              # the native side abandons chunks still in flight

    def seek_frames(self, frames):
        """reset, then continue as if `frames` frames of the stream had already gone by"""
        check(lib().csdr_chain_seek_frames(self.h, int(frames)))
        self._pending = []

    def close(self):
        self._pending = []
        self._h.close()


class host_array:
    """numpy view of page-locked host memory from csdr_host_alloc (freed with the object)."""

    def __init__(self, shape, dtype):
        self.dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * self.dtype.itemsize
        self.p = lib().csdr_host_alloc(n)
        if not self.p:
            raise CsdrError(_lib.ERR_NOMEM, "csdr_host_alloc failed")
        buf = (C.c_char * n).from_address(self.p)
        buf._csdr_owner = _PinnedBlock(self.p)          # the block lives as long as any numpy view of it (views keep `buf` as their base)
        self.a = np.frombuffer(buf, dtype=self.dtype).reshape(shape)


class _PinnedBlock:
    def __init__(self, p):
        self.p = p

    def __del__(self):
        try:
            if self.p:
                lib().csdr_host_free(self.p)
                self.p = None
        except Exception:
            pass


def firpfbchChannelizer(n, **kw):
    """firpfbchChannelizer n (Liquid.chs:864-866): Pipe IO (Array CF32) [Array CF32].
    Output is the list of n per-channel arrays the reference produces by slicing one
    channel-major buffer (Liquid.chs:850-862); an empty input yields [empty]."""
    def start():
        return Chain(ChainConfig(channels=n, dc_block=False, **kw))

    def process(r, a):
        x = _c64(a)
        if x.size == 0:
            return [np.empty(0, dtype=np.complex64)]
        y = r.process(x)
        return [y[k] for k in range(y.shape[0])]
    return Pipe(start, process, lambda r: r.close())
