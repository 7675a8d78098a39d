"""ctypes loader for the CPU oracle (oracle/csdr_oracle.c).

Test infrastructure: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(_ORACLE_DIR, "libcsdr_oracle.so")


def build(force=False):
    src = os.path.join(_ORACLE_DIR, "csdr_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        vp, u32, f32, i32 = C.c_void_p, C.c_uint32, C.c_float, C.c_int
        sigs = {
            "orc_kaiser_prototype": (None, [u32, u32, f32, vp]),
            "orc_nco_constrain": (u32, [f32]),
            "orc_pfb_offset": (f32, [u32]),
            "orc_nco_create": (vp, [f32]),
            "orc_nco_destroy": (None, [vp]),
            "orc_nco_get_theta": (u32, [vp]),
            "orc_nco_get_dtheta": (u32, [vp]),
            "orc_nco_set_theta": (None, [vp, u32]),
            "orc_nco_phasor": (None, [u32, vp]),
            "orc_nco_mix_down": (None, [vp, vp, vp, u32]),
            "orc_nco_mix_up": (None, [vp, vp, vp, u32]),
            "orc_dcblock_create": (vp, [f32]),
            "orc_dcblock_destroy": (None, [vp]),
            "orc_dcblock_a1": (f32, [vp]),
            "orc_dcblock_get_state": (None, [vp, vp]),
            "orc_dcblock_execute": (None, [vp, vp, u32, vp]),
            "orc_pfb_create": (vp, [u32, u32, f32]),
            "orc_pfb_destroy": (None, [vp]),
            "orc_pfb_taps": (vp, [vp]),
            "orc_pfb_analyzer_execute": (None, [vp, vp, vp]),
            "orc_pfb_set_dft_backward": (None, [vp, i32]),
            "orc_chan_set_dft_backward": (None, [vp, i32]),
            "orc_chain_set_dft_backward": (None, [vp, i32]),
            "orc_chan_create": (vp, [u32]),
            "orc_chan_destroy": (None, [vp]),
            "orc_chan_dtheta": (u32, [vp]),
            "orc_chan_theta": (u32, [vp]),
            "orc_chan_process": (None, [vp, vp, u32, vp]),
            "orc_agc_create_ref": (vp, [f32]),
            "orc_agc_destroy": (None, [vp]),
            "orc_agc_get_state": (None, [vp, vp, vp, vp, vp]),
            "orc_agc_execute_block_ref": (None, [vp, vp, u32, vp]),
            "orc_freqdem_create": (vp, [f32]),
            "orc_freqdem_destroy": (None, [vp]),
            "orc_freqdem_ref": (f32, [vp]),
            "orc_freqdem_demodulate_block": (None, [vp, vp, u32, vp]),
            "orc_msresamp_create": (vp, [f32, f32]),
            "orc_msresamp_destroy": (None, [vp]),
            "orc_msresamp_get_rate": (f32, [vp]),
            "orc_msresamp_num_halfband": (u32, [vp]),
            "orc_msresamp_halfband_len": (u32, [vp, u32]),
            "orc_msresamp_max_out": (u32, [vp, u32]),
            "orc_msresamp_execute": (u32, [vp, vp, u32, vp]),
            "orc_msresamp_get_pfb": (None, [vp, vp]),
            "orc_butter2_lowpass_create": (vp, [f32]),
            "orc_biquad_destroy": (None, [vp]),
            "orc_biquad_coeffs": (None, [vp, vp]),
            "orc_biquad_execute_block": (None, [vp, vp, u32, vp]),
            "orc_firdecim_create_kaiser": (vp, [u32, u32, f32]),
            "orc_firdecim_destroy": (None, [vp]),
            "orc_firdecim_len": (u32, [vp]),
            "orc_firdecim_taps": (None, [vp, vp]),
            "orc_firdecim_execute_block": (None, [vp, vp, u32, vp]),
            "orc_chain_create_wbfm": (vp, [u32, i32, i32, f32, f32, u32, i32]),
            "orc_ampdem_create": (vp, [f32]),
            "orc_ampdem_destroy": (None, [vp]),
            "orc_ampdem_demodulate_block": (None, [vp, vp, u32, vp]),
            "orc_mix_f32": (None, [vp, u32, u32, vp]),
            "orc_chain_create": (vp, [u32, i32, i32, f32, i32, f32, i32]),
            "orc_chain_destroy": (None, [vp]),
            "orc_chain_process": (None, [vp, vp, u32, vp]),
        }
        for name, (res, args) in sigs.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c64(x):
    x = np.ascontiguousarray(x, dtype=np.complex64)
    return x


def kaiser_prototype(M, m=7, As=80.0):
    h = np.zeros(2 * M * m + 1, dtype=np.float32)
    lib().orc_kaiser_prototype(M, m, As, _p(h))
    return h


def nco_constrain(f):
    return int(lib().orc_nco_constrain(np.float32(f)))


def pfb_offset(M):
    return float(lib().orc_pfb_offset(M))


def nco_phasor(theta):
    sc = np.zeros(2, dtype=np.float32)
    lib().orc_nco_phasor(int(theta) & 0xFFFFFFFF, _p(sc))
    return complex(float(sc[1]), float(sc[0]))  # cos + j sin


class _Obj:
    _destroy = None

    def __del__(self):
        try:
            if getattr(self, "h", None):
                getattr(lib(), self._destroy)(self.h)
                self.h = None
        except Exception:
            pass


class Nco(_Obj):
    _destroy = "orc_nco_destroy"

    def __init__(self, freq):
        self.h = lib().orc_nco_create(np.float32(freq))

    @property
    def theta(self):
        return int(lib().orc_nco_get_theta(self.h))

    @property
    def dtheta(self):
        return int(lib().orc_nco_get_dtheta(self.h))

    def mix_down(self, x):
        x = _c64(x)
        y = np.empty_like(x)
        lib().orc_nco_mix_down(self.h, _p(x), _p(y), x.size)
        return y

    def mix_up(self, x):
        x = _c64(x)
        y = np.empty_like(x)
        lib().orc_nco_mix_up(self.h, _p(x), _p(y), x.size)
        return y


class DcBlock(_Obj):
    _destroy = "orc_dcblock_destroy"

    def __init__(self, alpha=0.0005):
        self.h = lib().orc_dcblock_create(np.float32(alpha))

    @property
    def a1(self):
        return float(lib().orc_dcblock_a1(self.h))

    @property
    def state(self):
        v = np.zeros(2, dtype=np.float32)
        lib().orc_dcblock_get_state(self.h, _p(v))
        return complex(v[0], v[1])

    def execute(self, x):
        x = _c64(x)
        y = np.empty_like(x)
        lib().orc_dcblock_execute(self.h, _p(x), x.size, _p(y))
        return y


class Pfb(_Obj):
    _destroy = "orc_pfb_destroy"

    def __init__(self, M, m=7, As=80.0):
        self.M, self.p = M, 2 * m
        self.h = lib().orc_pfb_create(M, m, As)

    @property
    def taps(self):
        ptr = lib().orc_pfb_taps(self.h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(self.M * self.p,)).copy()

    def analyzer_execute(self, x):
        x = _c64(x)
        assert x.size == self.M
        y = np.empty_like(x)
        lib().orc_pfb_analyzer_execute(self.h, _p(x), _p(y))
        return y


class Chan(_Obj):
    """firpfbchChan: premix + analyzer per frame + transpose -> [M][nf]."""
    _destroy = "orc_chan_destroy"

    def __init__(self, M):
        self.M = M
        self.h = lib().orc_chan_create(M)

    @property
    def dtheta(self):
        return int(lib().orc_chan_dtheta(self.h))

    @property
    def theta(self):
        return int(lib().orc_chan_theta(self.h))

    def process(self, x):
        x = _c64(x)
        assert x.size % self.M == 0
        nf = x.size // self.M
        y = np.empty((self.M, nf), dtype=np.complex64)
        lib().orc_chan_process(self.h, _p(x), x.size, _p(y))
        return y


class Agc(_Obj):
    _destroy = "orc_agc_destroy"

    def __init__(self, threshold_db):
        self.h = lib().orc_agc_create_ref(np.float32(threshold_db))

    @property
    def state(self):
        g, y2 = C.c_float(), C.c_float()
        mode, timer = C.c_int(), C.c_uint()
        lib().orc_agc_get_state(self.h, C.byref(g), C.byref(y2), C.byref(mode), C.byref(timer))
        return dict(g=g.value, y2=y2.value, mode=mode.value, timer=timer.value)

    def execute_block(self, x):
        x = _c64(x)
        y = np.empty_like(x)
        lib().orc_agc_execute_block_ref(self.h, _p(x), x.size, _p(y))
        return y


class FreqDem(_Obj):
    _destroy = "orc_freqdem_destroy"

    def __init__(self, kf):
        self.h = lib().orc_freqdem_create(np.float32(kf))

    @property
    def ref(self):
        return float(lib().orc_freqdem_ref(self.h))

    def demodulate_block(self, r):
        r = _c64(r)
        m = np.empty(r.size, dtype=np.float32)
        lib().orc_freqdem_demodulate_block(self.h, _p(r), r.size, _p(m))
        return m


class MsResamp(_Obj):
    """resampler r as (Liquid.chs:115-117): msresamp_crcf structure, parameters fixed by this repo (unpinned)"""
    _destroy = "orc_msresamp_destroy"

    def __init__(self, rate, As=60.0):
        self.h = lib().orc_msresamp_create(np.float32(rate), np.float32(As))
        if not self.h:
            raise ValueError("rate out of range")

    @property
    def num_halfband(self):
        return int(lib().orc_msresamp_num_halfband(self.h))

    def halfband_len(self, s):
        return int(lib().orc_msresamp_halfband_len(self.h, s))

    def execute(self, x):
        x = _c64(x)
        y = np.empty(int(lib().orc_msresamp_max_out(self.h, x.size)), dtype=np.complex64)
        n = int(lib().orc_msresamp_execute(self.h, _p(x), x.size, _p(y)))
        return y[:n].copy()


class Butter2(_Obj):
    """iirFilter 2 fc 0 10 10 (Liquid.chs:636-638 -> iirfilt_rrrf_create_prototype BUTTER LOWPASS SOS) -- recalled"""
    _destroy = "orc_biquad_destroy"

    def __init__(self, fc):
        self.h = lib().orc_butter2_lowpass_create(np.float32(fc))

    @property
    def coeffs(self):
        ba = np.zeros(6, dtype=np.float32)
        lib().orc_biquad_coeffs(self.h, _p(ba))
        return ba[:3].copy(), ba[3:].copy()

    def execute_block(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.empty_like(x)
        lib().orc_biquad_execute_block(self.h, _p(x), x.size, _p(y))
        return y


class FirDecim(_Obj):
    """firDecimator m (Liquid.chs:485-501): firdecim_rrrf_create_kaiser(m, 10, 60) -- recalled"""
    _destroy = "orc_firdecim_destroy"

    def __init__(self, M, m=10, As=60.0):
        self.M = M
        self.h = lib().orc_firdecim_create_kaiser(M, m, np.float32(As))

    @property
    def taps(self):
        h = np.zeros(int(lib().orc_firdecim_len(self.h)), dtype=np.float32)
        lib().orc_firdecim_taps(self.h, _p(h))
        return h

    def execute_block(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.size % self.M == 0
        y = np.empty(x.size // self.M, dtype=np.float32)
        lib().orc_firdecim_execute_block(self.h, _p(x), x.size, _p(y))
        return y


class AmpDem(_Obj):
    """ampmodem DSB, carrier present (amdemodCreate, Liquid.chs:452-457) -- recalled, unpinned"""
    _destroy = "orc_ampdem_destroy"

    def __init__(self, mod_index=0.8):
        self.h = lib().orc_ampdem_create(np.float32(mod_index))

    def demodulate_block(self, y):
        y = _c64(y)
        x = np.empty(y.size, dtype=np.float32)
        lib().orc_ampdem_demodulate_block(self.h, _p(y), y.size, _p(x))
        return x


def mix_f32(chans):
    chans = np.ascontiguousarray(chans, dtype=np.float32)
    M, n = chans.shape
    out = np.empty(n, dtype=np.float32)
    lib().orc_mix_f32(_p(chans), M, n, _p(out))
    return out


class Chain(_Obj):
    """assembleFold's DSP (SoapySDR.hs:208-226) on compacted chunks."""
    _destroy = "orc_chain_destroy"

    def __init__(self, M, dc_block=True, agc_db=0.0, demod="none", kf=0.3, mix=False, decim=4, deemph_fc=0.025, dft_backward=False):
        self.M = M
        self.demod = {"none": 0, "fm": 1, "am": 2, "wbfm": 3}[demod]
        self.mix = bool(mix) and M > 1
        self.decim = decim if self.demod == 3 else 1
        if self.demod == 3:
            self.h = lib().orc_chain_create_wbfm(M, int(dc_block), int(agc_db != 0.0), np.float32(agc_db),
                                                 np.float32(deemph_fc), decim, int(self.mix))
        else:
            self.h = lib().orc_chain_create(M, int(dc_block), int(agc_db != 0.0), np.float32(agc_db),
                                            self.demod, np.float32(kf), int(self.mix))
        if dft_backward:                      # the other possible convention of the analyzer's transform (unpinned: SURVEY 7.1)
            lib().orc_chain_set_dft_backward(self.h, 1)

    def process(self, x):
        x = _c64(x)
        assert x.size % self.M == 0
        nf = x.size // self.M
        dt = np.float32 if self.demod in (1, 2, 3) else np.complex64
        no = nf // self.decim
        shape = (no,) if self.mix else (self.M, no)
        out = np.empty(shape, dtype=dt)
        lib().orc_chain_process(self.h, _p(x), x.size, _p(out))
        return out
