"""The C-ABI library loads without a GPU and exports every symbol include/csdr.h declares
(no compute calls here); creating an object without a GPU must fail loudly, not fall back."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "csdr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(csdr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import composable_sdr_amd as cs
    from composable_sdr_amd import _lib
    if not os.path.exists(cs.lib_path()):
        cs.build_library()
    lib = C.CDLL(cs.lib_path())
    names = _declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/csdr.h but not exported"
    # and the ctypes table binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == names


def test_library_exports_nothing_but_the_header():
    """What a Haskell `foreign import ccall` can see (Liquid.chs:730-780 binds liquid's C symbols the same way) is the header:
    the dynamic symbol table holds the declared csdr_* functions and no C++ internal, kernel handle or template instance
    (-fvisibility=hidden + csrc/exports.map)."""
    import subprocess
    import composable_sdr_amd as cs
    if not os.path.exists(cs.lib_path()):
        cs.build_library()
    out = subprocess.check_output(["nm", "-D", "--defined-only", cs.lib_path()], text=True)
    defined = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert defined == _declared_symbols(), sorted(set(defined) ^ set(_declared_symbols()))
    assert not [n for n in defined if n.startswith("_Z")]


def test_cfg_struct_matches_header_layout():
    from composable_sdr_amd import _lib
    cfg = _lib.ChainCfg()
    _lib.lib().csdr_chain_cfg_default(C.byref(cfg), 256)
    assert cfg.struct_size == C.sizeof(_lib.ChainCfg) == 72
    assert (cfg.channels, cfg.dc_block, cfg.max_frames, cfg.pfb_m) == (256, 1, 4096, 7)
    assert abs(cfg.dc_alpha - 0.0005) < 1e-9 and abs(cfg.pfb_as - 80.0) < 1e-6 and cfg.device == -1


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import composable_sdr_amd as cs
    with pytest.raises(cs.CsdrError) as e:
        cs.Chain(channels=256)
    assert e.value.code == -3 and "no CPU fallback" in str(e.value)
    for pipe in (cs.dcBlocker(), cs.mixDown(0.1), cs.fmDemodulator(0.3), cs.automaticGainControl(-10.0)):
        with pytest.raises(cs.CsdrError):
            pipe._start()


def test_product_package_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "composable_sdr_amd")
    for dp, _, files in os.walk(pkg):
        if "build" in dp.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_lib" not in txt and "csdr_oracle" not in txt and "libcsdr_oracle" not in txt, (dp, f)


def test_route_table_names_every_product_kernel_family():
    """csdr_route_table() is what csdr_chain_create selects from: it must name the plans and the kernels DESIGN.md quotes."""
    from composable_sdr_amd import _lib
    t = _lib.lib().csdr_route_table().decode()
    for name in ("fused-256", "k_run256v2", "k_tile256", "fused-k_run64", "k_run64v2", "fused-k_run1024", "k_run1024v2", "k_shard1024", "generic",
                 "k_dc_fold", "k_dc_fold8", "k_agc_spec_tm", "k_agc_fix", "tail-only"):
        assert name in t, name
