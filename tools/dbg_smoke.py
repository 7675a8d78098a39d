import sys, os, ctypes as C
sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/tests")
import numpy as np
import composable_sdr_amd as cs, oracle_lib as O
from composable_sdr_amd import _lib, pipes
from synth import synth_cf32
if os.environ.get("OLD_CFG"):
    class OldCfg(C.Structure):
        _fields_ = _lib.ChainCfg._fields_[:-2]
    real = pipes.Chain.__init__
    def init(self, cfg=None, **kw):
        cfg = cfg or pipes.ChainConfig(**kw)
        self.cfg = cfg
        c = OldCfg()
        _lib.lib().csdr_chain_cfg_default(C.byref(c), cfg.channels)
        c.channels = cfg.channels; c.demod = {"none":0,"fm":1}[cfg.demod]; c.kf = cfg.kf; c.max_frames = cfg.max_frames; c.flags = cfg.flags
        h = C.c_void_p(); _lib.check(_lib.lib().csdr_chain_create(C.byref(c), C.byref(h)))
        self._h = pipes._Handle(h, _lib.lib().csdr_chain_destroy); self.M = cfg.channels; self.C = cfg.channels; self.mixed = False
        self.out_dtype = np.float32 if cfg.demod == "fm" else np.complex64; self.decim = 1
    pipes.Chain.__init__ = init
M,nf,kf=256,512,0.3
x=synth_cf32(M*nf,M,seed=1)
want=O.Chain(M,demod="fm",kf=kf).process(x)
ch=cs.Chain(channels=M,demod="fm",kf=kf,max_frames=nf); got=ch.process(x); 
d=(got.astype(np.float64)-want+0.5/kf)%(1.0/kf)-0.5/kf
print(os.environ.get("CSDR_LIB"), "p99.9", float(np.quantile(np.abs(d),0.999)), "max", float(np.abs(d).max()), "median", float(np.median(np.abs(d))))
