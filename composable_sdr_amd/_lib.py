"""Loads libcsdr_hip.so and declares the C ABI of include/csdr.h for ctypes."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("CSDR_LIB") or os.path.join(_HERE, "libcsdr_hip.so")   # CSDR_LIB: A/B builds

DEMOD_NONE, DEMOD_FM, DEMOD_AM, DEMOD_WBFM = 0, 1, 2, 3
FLAG_TIME_KERNELS, FLAG_FORCE_GENERIC, FLAG_QUIET, FLAG_AGC_SEQUENTIAL, FLAG_NO_MIX_IDENTITY = 1, 2, 4, 8, 16
FLAG_TIME_REGION = 32
FLAG_TAIL_ONLY = 64
FLAG_DFT_BACKWARD = 128

ERR_INVALID, ERR_HIP, ERR_NODEV, ERR_SIZE, ERR_NOMEM = -1, -2, -3, -4, -5


class CsdrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"csdr error {code}: {msg}")
        self.code = code


class ChainCfg(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("channels", C.c_uint32), ("dc_block", C.c_uint32),
        ("dc_alpha", C.c_float), ("agc_threshold_db", C.c_float), ("demod", C.c_uint32),
        ("kf", C.c_float), ("mix", C.c_uint32), ("chan_first", C.c_uint32), ("chan_count", C.c_uint32),
        ("device", C.c_int32), ("max_frames", C.c_uint32), ("flags", C.c_uint32),
        ("pfb_m", C.c_uint32), ("pfb_as", C.c_float), ("wbfm_decim", C.c_uint32), ("deemph_fc", C.c_float),
        ("chan_stride", C.c_uint32),
    ]


# name -> (restype, argtypes); kept in one table so tests can check that every symbol
# include/csdr.h declares is exported.
_vp, _u32, _f32, _i32 = C.c_void_p, C.c_uint32, C.c_float, C.c_int
_pp = C.POINTER(C.c_void_p)
_pu32 = C.POINTER(C.c_uint32)
SIGNATURES = {
    "csdr_last_error": (C.c_char_p, []),
    "csdr_device_count": (_i32, []),
    "csdr_version": (C.c_char_p, []),
    "csdr_dcblock_create": (_i32, [_f32, _u32, _pp]),
    "csdr_dcblock_process": (_i32, [_vp, _vp, _u32, _vp]),
    "csdr_dcblock_process_device": (_i32, [_vp, _vp, _u32, _vp, _vp]),
    "csdr_dcblock_destroy": (_i32, [_vp]),
    "csdr_nco_create": (_i32, [_f32, _u32, _pp]),
    "csdr_nco_mix_down": (_i32, [_vp, _vp, _u32, _vp]),
    "csdr_nco_mix_up": (_i32, [_vp, _vp, _u32, _vp]),
    "csdr_nco_get_words": (_i32, [_vp, _pu32, _pu32]),
    "csdr_nco_destroy": (_i32, [_vp]),
    "csdr_agc_create": (_i32, [_f32, _u32, _u32, _pp]),
    "csdr_agc_process": (_i32, [_vp, _vp, _u32, _vp]),
    "csdr_agc_destroy": (_i32, [_vp]),
    "csdr_freqdem_create": (_i32, [_f32, _u32, _u32, _pp]),
    "csdr_freqdem_process": (_i32, [_vp, _vp, _u32, _vp]),
    "csdr_freqdem_destroy": (_i32, [_vp]),
    "csdr_iirfilt_create": (_i32, [_u32, _f32, _f32, _f32, _f32, _u32, _u32, _pp]),
    "csdr_iirfilt_process": (_i32, [_vp, _vp, _u32, _vp]),
    "csdr_iirfilt_destroy": (_i32, [_vp]),
    "csdr_firdecim_create": (_i32, [_u32, _u32, _u32, _pp]),
    "csdr_firdecim_process": (_i32, [_vp, _vp, _u32, _vp]),
    "csdr_firdecim_destroy": (_i32, [_vp]),
    "csdr_resamp_create": (_i32, [_f32, _f32, _u32, _pp]),
    "csdr_resamp_get_rate": (_f32, [_vp]),
    "csdr_resamp_max_out": (_u32, [_vp, _u32]),
    "csdr_resamp_process": (_i32, [_vp, _vp, _u32, _vp, _vp]),
    "csdr_resamp_process_device": (_i32, [_vp, _vp, _u32, _vp, _vp, _vp]),
    "csdr_resamp_destroy": (_i32, [_vp]),
    "csdr_ampdem_create": (_i32, [_f32, _u32, _u32, _pp]),
    "csdr_ampdem_process": (_i32, [_vp, _vp, _u32, _vp]),
    "csdr_ampdem_destroy": (_i32, [_vp]),
    "csdr_chain_cfg_default": (None, [C.POINTER(ChainCfg), _u32]),
    "csdr_chain_create": (_i32, [C.POINTER(ChainCfg), _pp]),
    "csdr_chain_process": (_i32, [_vp, _vp, _u32, _vp, _pu32]),
    "csdr_chain_process_device": (_i32, [_vp, _vp, _u32, _vp, _pu32, _vp]),
    "csdr_chain_submit_device": (_i32, [_vp, _vp, _u32, _vp, _pu32, _vp]),
    "csdr_chain_wait_device": (_i32, [_vp, _vp]),
    "csdr_chain_debug_independent_launches": (_u32, [_vp]),
    "csdr_chain_submit": (_i32, [_vp, _vp, _u32, _vp]),
    "csdr_chain_collect": (_i32, [_vp, _pu32]),
    "csdr_chain_status": (_i32, [_vp]),
    "csdr_host_alloc": (_vp, [C.c_size_t]),
    "csdr_host_free": (None, [_vp]),
    "csdr_chain_reset": (_i32, [_vp]),
    "csdr_chain_seek_frames": (_i32, [_vp, C.c_uint64]),
    "csdr_chain_destroy": (_i32, [_vp]),
    "csdr_chain_out_elem_size": (_u32, [_vp]),
    "csdr_chain_get_taps": (_i32, [_vp, _vp, _u32]),
    "csdr_chain_get_nco": (_i32, [_vp, _pu32, _pu32]),
    "csdr_chain_path": (C.c_char_p, [_vp]),
    "csdr_route_table": (C.c_char_p, []),
    "csdr_chain_debug_trace": (_i32, [_vp, _vp, _u32]),
    "csdr_chain_debug_agc": (_i32, [_vp, _vp, _vp]),
    "csdr_chain_debug_agc_tile_major_calls": (_u32, [_vp]),
    "csdr_chain_kernel_time": (C.c_char_p, [_vp, C.POINTER(C.c_double), _pu32]),
    "csdr_chain_get_cfg": (_i32, [_vp, C.POINTER(ChainCfg)]),
    # collectives (RCCL over xGMI, csrc/comm.cpp)
    "csdr_comm_unique_id": (_i32, [_vp]),
    "csdr_comm_create": (_i32, [_i32, _i32, _vp, _i32, _pp]),
    "csdr_comm_rank": (_i32, [_vp]),
    "csdr_comm_world": (_i32, [_vp]),
    "csdr_comm_destroy": (_i32, [_vp]),
    "csdr_comm_broadcast": (_i32, [_vp, _vp, C.c_size_t, _i32, _vp]),
    "csdr_comm_allreduce_f32": (_i32, [_vp, _vp, C.c_size_t, _vp]),
    "csdr_chain_process_device_mix": (_i32, [_vp, _vp, _vp, _u32, _vp, _pu32, _vp]),
    "csdr_chain_process_mix": (_i32, [_vp, _vp, _vp, _u32, _vp, _pu32]),
    "csdr_hybrid_exchange": (_i32, [_vp, _vp, _vp, _u32, _pu32, _u32, _vp]),
}
COMM_ID_BYTES = 128


def lib_path():
    return _SO


def build_library(force=False):
    """Compile libcsdr_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", src, "clean"])
    subprocess.check_call(["make", "-C", src, "-j4"])
    return _SO


_lib = None


def lib():
    """The loaded library.  Fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise CsdrError(ERR_NODEV, f"{_SO} is missing: build it with "
                            "`python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no CPU fallback)")
        L = C.CDLL(_SO)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code):
    if code != 0:
        raise CsdrError(code, lib().csdr_last_error().decode("utf-8", "replace"))
